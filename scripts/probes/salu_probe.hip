// Single-wave issue-cost probe for the scalar chain (dmi_chains.hip): how many core clocks does one wave spend
// per instruction for the instruction kinds the rANS recurrence uses?  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP 512
#define STR2(x) #x
#define STR(x) STR2(x)

#define PROBE(NAME, BODY)                                                                              \
  __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {               \
    uint32_t x = a, y = b, z = c, w = d, v = 0;                                                        \
    uint64_t t0 = clock64();                                                                           \
    asm volatile(".rept " STR(REP) "\n" BODY "\n.endr" : "+s"(x), "+s"(y), "+s"(z), "+s"(w), "+v"(v) : : "s40", "s41", "scc"); \
    uint64_t t1 = clock64();                                                                           \
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = x + y + z + w + v; }                            \
  }

PROBE(p_empty, "")
PROBE(p_add_dep, "s_add_u32 %0, %0, %1")
PROBE(p_add_indep, "s_add_u32 %0, %0, %1\n s_add_u32 %2, %2, %1\n s_add_u32 %3, %3, %1")
PROBE(p_mul_dep, "s_mul_i32 %0, %0, %1")
PROBE(p_mulhi_dep, "s_mul_hi_u32 %0, %0, %1")
PROBE(p_lshr_dep, "s_lshr_b32 %0, %0, %1")
PROBE(p_flbit_dep, "s_flbit_i32_b32 %0, %0")
PROBE(p_and_dep, "s_and_b32 %0, %0, %1")
PROBE(p_add_wl, "s_add_u32 %0, %0, %1\n v_writelane_b32 %4, %0, 5")
PROBE(p_add_wl_other, "s_add_u32 %0, %0, %1\n v_writelane_b32 %4, %2, 5")
PROBE(p_wl_only, "v_writelane_b32 %4, %2, 5")
PROBE(p_vadd_dep, "v_add_u32 %4, %4, %4")
PROBE(p_vmulhi_dep, "v_mul_hi_u32 %4, %4, %4")
PROBE(p_nop, "s_nop 0")
// the current chain step: x=%0, m=%1, d=%2, c=%3 (b folded as a literal 3)
PROBE(p_step, "s_mul_hi_u32 s40, %0, %1\n s_lshr_b32 s40, s40, 3\n s_flbit_i32_b32 s41, s40\n s_sub_i32 s41, 29, s41\n s_and_b32 s41, s41, 24\n"
              "v_writelane_b32 %4, %0, 7\n s_lshr_b32 %0, %0, s41\n s_lshr_b32 s40, s40, s41\n s_mul_i32 s40, s40, %2\n s_add_i32 %0, %0, %3\n s_add_i32 %0, %0, s40")
PROBE(p_step_nowl, "s_mul_hi_u32 s40, %0, %1\n s_lshr_b32 s40, s40, 3\n s_flbit_i32_b32 s41, s40\n s_sub_i32 s41, 29, s41\n s_and_b32 s41, s41, 24\n"
              "s_lshr_b32 %0, %0, s41\n s_lshr_b32 s40, s40, s41\n s_mul_i32 s40, s40, %2\n s_add_i32 %0, %0, %3\n s_add_i32 %0, %0, s40")
// variant: writelane placed right after the state is produced by the previous step (i.e. first in the step)
PROBE(p_step_wlfirst, "v_writelane_b32 %4, %0, 7\n s_mul_hi_u32 s40, %0, %1\n s_lshr_b32 s40, s40, 3\n s_flbit_i32_b32 s41, s40\n s_sub_i32 s41, 29, s41\n s_and_b32 s41, s41, 24\n"
              "s_lshr_b32 %0, %0, s41\n s_lshr_b32 s40, s40, s41\n s_mul_i32 s40, s40, %2\n s_add_i32 %0, %0, %3\n s_add_i32 %0, %0, s40")
// variant: 64-bit shift of {q0:x} + mask instead of two 32-bit shifts; and s_andn2-based byte count
PROBE(p_step_alt, "s_mul_hi_u32 s40, %0, %1\n s_lshr_b32 s40, s40, 3\n s_flbit_i32_b32 s41, s40\n s_sub_i32 s41, 29, s41\n s_and_b32 s41, s41, 24\n"
              "v_writelane_b32 %4, %0, 7\n s_lshr_b32 %0, %0, s41\n s_lshr_b32 s40, s40, s41\n s_mul_i32 s40, s40, %2\n s_add_i32 %0, %0, s40\n s_add_i32 %0, %0, %3")

// calibration: s_memtime (clock64) against the 100 MHz wall_clock64 over a long run of chain steps
__global__ void p_calib(uint64_t* out, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  uint32_t x = a, y = b, z = c, w = d, v = 0;
  uint64_t t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < 16000; ++it)
    asm volatile(".rept 64\n s_mul_hi_u32 s40, %0, %1\n s_lshr_b32 s40, s40, 3\n s_flbit_i32_b32 s41, s40\n s_sub_i32 s41, 29, s41\n s_and_b32 s41, s41, 24\n"
              "v_writelane_b32 %4, %0, 7\n s_lshr_b32 %0, %0, s41\n s_lshr_b32 s40, s40, s41\n s_mul_i32 s40, s40, %2\n s_add_i32 %0, %0, %3\n s_add_i32 %0, %0, s40\n.endr"
                 : "+s"(x), "+s"(y), "+s"(z), "+s"(w), "+v"(v) : : "s40", "s41", "scc");
  uint64_t t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = x + v; }
}

// chain steps with the record traffic of the real loop: one s_load_dwordx16 per 4 steps (into scratch SGPRs, from `out`)
__global__ void p_step_sload(uint64_t* out, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  uint32_t x = a, y = b, z = c, w = d, v = 0;
  uint64_t t0 = clock64();
  asm volatile(".rept 128\n"
               ".rept 4\n s_mul_hi_u32 s40, %0, %1\n s_lshr_b32 s40, s40, 3\n s_flbit_i32_b32 s41, s40\n s_sub_i32 s41, 29, s41\n s_and_b32 s41, s41, 24\n"
               "v_writelane_b32 %4, %0, 7\n s_lshr_b32 %0, %0, s41\n s_lshr_b32 s40, s40, s41\n s_mul_i32 s40, s40, %2\n s_add_i32 %0, %0, %3\n s_add_i32 %0, %0, s40\n.endr\n"
               "s_load_dwordx16 s[44:59], %5, 0x0\n.endr\n s_waitcnt lgkmcnt(0)"
               : "+s"(x), "+s"(y), "+s"(z), "+s"(w), "+v"(v) : "s"(out) : "s40", "s41", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "scc");
  uint64_t t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = x + y + z + w + v; }
}

typedef void (*kern_t)(uint64_t*, uint32_t, uint32_t, uint32_t, uint32_t);
struct Entry { const char* name; kern_t k; int instrs; };

int main() {
  uint64_t* d; hipMalloc(&d, 4096);
  Entry es[] = {{"empty", p_empty, 0}, {"s_add dep", p_add_dep, 1}, {"s_add x3 indep", p_add_indep, 3}, {"s_mul_i32 dep", p_mul_dep, 1},
                {"s_mul_hi_u32 dep", p_mulhi_dep, 1}, {"s_lshr dep", p_lshr_dep, 1}, {"s_flbit dep", p_flbit_dep, 1}, {"s_and dep", p_and_dep, 1},
                {"s_add + writelane(dep)", p_add_wl, 2}, {"s_add + writelane(indep)", p_add_wl_other, 2}, {"writelane only", p_wl_only, 1},
                {"v_add dep", p_vadd_dep, 1}, {"v_mul_hi dep", p_vmulhi_dep, 1}, {"s_nop", p_nop, 1},
                {"chain step", p_step, 11}, {"chain step no writelane", p_step_nowl, 10}, {"chain step writelane first", p_step_wlfirst, 11},
                {"chain step add order", p_step_alt, 11}, {"4 steps + s_load_dwordx16 (x128)", p_step_sload, 45}};
  for (auto& e : es) {
    uint64_t best = ~0ull;
    for (int r = 0; r < 5; ++r) {
      hipLaunchKernelGGL(e.k, 1, 64, 0, 0, d, 0x12345678u, 0x9E3779B1u, 1000u, 77u);
      uint64_t h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      if (h[0] < best) best = h[0];
    }
    printf("%-32s %8llu clk / %d reps = %7.2f clk/rep  (%5.2f per instr)\n", e.name, (unsigned long long)best, REP, (double)best / REP,
           e.instrs ? (double)best / REP / e.instrs : 0.0);
  }
  for (int r = 0; r < 3; ++r) {
    hipLaunchKernelGGL(p_calib, 1, 64, 0, 0, d, 0x12345678u, 0x9E3779B1u, 1000u, 77u);
    uint64_t h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("calib: %llu memtime clk, %llu wall ticks (100 MHz) => memtime %.1f MHz; %.2f memtime clk/step, %.2f ns/step\n", (unsigned long long)h[0],
           (unsigned long long)h[1], (double)h[0] / ((double)h[1] / 100.0), (double)h[0] / (16000.0 * 64), (double)h[1] * 10.0 / (16000.0 * 64));
  }
  return 0;
}
