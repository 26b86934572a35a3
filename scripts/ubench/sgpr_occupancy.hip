// micro-benchmark (round 6): how many waves per SIMD are RESIDENT on gfx950 as a function of a kernel's SGPR count.
//
// LLVM's occupancy table for gfx9 assumes 800 SGPRs per SIMD in granules of 16 (7 waves of 112, 8 of 96: "; Occupancy: 7" in the
// assembly of a 103-SGPR kernel).  Round 6 found the per-tile sort's main kernel at 103 SGPRs running six workgroups per CU
// where LDS allows seven (profiles/r6/19b).  This measures it directly: one-wave workgroups, no LDS, ~16 VGPRs, each wave spins
// for a fixed time on the 100 MHz clock and logs {start, end}; `want` waves per SIMD are launched on every SIMD (256 CUs x 4 x
// want) and the number resident at once is the number whose start lies before the first wave's end.
//
//   hipcc --offload-arch=gfx950 -O3 -o sgpr_occupancy.bin sgpr_occupancy.hip && ./sgpr_occupancy.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int TOP>
__global__ __launch_bounds__(64) void spin_kernel(unsigned long long* log, unsigned ticks) {
  // touching s[TOP] makes the kernel's SGPR count TOP + 1 (+ VCC / flat scratch / XNACK as the assembler adds them)
  if constexpr (TOP == 31) asm volatile("s_mov_b32 s31, 0" ::: "s31");
  if constexpr (TOP == 47) asm volatile("s_mov_b32 s47, 0" ::: "s47");
  if constexpr (TOP == 63) asm volatile("s_mov_b32 s63, 0" ::: "s63");
  if constexpr (TOP == 71) asm volatile("s_mov_b32 s71, 0" ::: "s71");
  if constexpr (TOP == 79) asm volatile("s_mov_b32 s79, 0" ::: "s79");
  if constexpr (TOP == 87) asm volatile("s_mov_b32 s87, 0" ::: "s87");
  if constexpr (TOP == 89) asm volatile("s_mov_b32 s89, 0" ::: "s89");
  if constexpr (TOP == 95) asm volatile("s_mov_b32 s95, 0" ::: "s95");
  if constexpr (TOP == 101) asm volatile("s_mov_b32 s101, 0" ::: "s101");
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0) { log[2 * blockIdx.x] = t0; log[2 * blockIdx.x + 1] = wall_clock64(); }
}

template <int TOP>
void run(unsigned long long* d_log, int want) {
  const int waves = 256 * 4 * want;
  std::vector<unsigned long long> h(2 * waves);
  hipFuncAttributes at;
  (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(spin_kernel<TOP>));
  int best = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(spin_kernel<TOP>, dim3(waves), dim3(64), 0, 0, d_log, 5000u);      // 50 us
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d_log, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long first_end = ~0ull;
    for (int i = 0; i < waves; ++i) first_end = std::min(first_end, h[2 * i + 1]);
    int resident = 0;
    for (int i = 0; i < waves; ++i) resident += h[2 * i] < first_end ? 1 : 0;
    best = std::max(best, resident);
  }
  printf("s%-3d touched, %2d waves per SIMD launched (%5d waves): %5d resident at once = %.2f per SIMD\n", TOP, want, waves, best, best / 1024.0);
}

int main() {
  unsigned long long* d_log;
  (void)hipMalloc(&d_log, 2 * 256 * 4 * 10 * sizeof(unsigned long long));
  for (int want : {8, 10}) {
    run<31>(d_log, want); run<47>(d_log, want); run<63>(d_log, want); run<71>(d_log, want); run<79>(d_log, want);
    run<87>(d_log, want); run<89>(d_log, want); run<95>(d_log, want); run<101>(d_log, want);
  }
  return 0;
}
