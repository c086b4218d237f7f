"""round 6: does a process that has imported / initialised torch get blit-kernel copies (rocprofv3 kernel trace: __amd_rocclr_copyBuffer) where a plain HIP
program gets SDMA copies?  python3 sdma_torch.py [none|import|init|tensor]   (run under rocprofv3 --kernel-trace, count copyBuffer rows with grid 131072)"""
import ctypes as C, sys, os
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode != "none":
    import torch
    if mode in ("init", "tensor"):
        torch.cuda.init(); torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
hip = C.CDLL("libamdhip64.so")
B = 64 << 20
dev = C.c_void_p(); pin = C.c_void_p(); s = C.c_void_p()
assert hip.hipMalloc(C.byref(dev), C.c_size_t(B)) == 0
assert hip.hipHostMalloc(C.byref(pin), C.c_size_t(B), 0) == 0
assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
for _ in range(3):
    assert hip.hipMemcpyAsync(pin, dev, C.c_size_t(B), 2, s) == 0    # D2H
    assert hip.hipStreamSynchronize(s) == 0
    assert hip.hipMemcpyAsync(dev, pin, C.c_size_t(B), 1, s) == 0    # H2D
    assert hip.hipStreamSynchronize(s) == 0
if mode == "tensor":
    t = torch.empty(B // 4, dtype=torch.int32, device="cuda:0"); h = torch.empty(B // 4, dtype=torch.int32).pin_memory()
    h.copy_(t, non_blocking=True); torch.cuda.synchronize()
print(mode, "done", {k: v for k, v in os.environ.items() if "SDMA" in k or k.startswith("HSA_") or k.startswith("HIP_") or k.startswith("GPU_") or k.startswith("AMD_")})
