#!/bin/bash
# Builds the library's host code with ThreadSanitizer (the recipe of `make asan` with the sanitizer swapped) and runs the host-side tests that go through the
# worker pool, the condition-variable hand-overs and the host stages' thread fans (round 6).  CPU only; no torch in the process.
set -e
cd "$(dirname "$0")/.."
make -s -C draco-oxide_amd/csrc
mkdir -p /tmp/dmi_tsan
( cd draco-oxide_amd/csrc && make -n asan ASAN_DIR=/tmp/dmi_tsan 2>/dev/null | sed 's/-fsanitize=address,undefined/-fsanitize=thread/g; s/-shared-libasan/-shared-libsan/g; s/libdraco_mi_asan/libdraco_mi_tsan/g' > /tmp/dmi_tsan/build.sh && bash -e /tmp/dmi_tsan/build.sh 2>&1 | grep -v "Woption-ignored\|warning: ignoring" || true )
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)
export DMI_LIBRARY=/tmp/dmi_tsan/libdraco_mi_tsan.so LD_PRELOAD=$RT TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4 exitcode=66" DMI_NO_TORCH_PREIMPORT=1
python -m pytest tests/test_host_connectivity.py tests/test_decode_connectivity.py tests/test_gltf.py tests/test_host_chains.py tests/test_native_gltf_json.py -x -q -m "not gpu" -k "not fork" 2>&1 | tee /tmp/dmi_tsan/run.log | tail -3
if grep -q "WARNING: ThreadSanitizer" /tmp/dmi_tsan/run.log; then echo "host tsan: REPORTS (see /tmp/dmi_tsan/run.log)"; exit 1; fi
echo "host tsan: clean"
