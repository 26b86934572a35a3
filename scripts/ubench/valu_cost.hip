// micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU forms the
// raster blend uses.  One workgroup of 256 threads (1 wave per SIMD) or 1024 (4 per SIMD) per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0001f, c = 0.5f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a1, a2}, p2 = {a2, a3}, p3 = {a3, a0}, pb = {b, b}, pc = {c, c};
  unsigned long long m = iters & 1 ? 0xffffffff00000000ull : 0x00000000ffffffffull;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if (KIND == 1) { REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
    if (KIND == 2) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (KIND == 3) { asm volatile("s_mov_b64 s[20:21], %0" :: "s"(m) : "s20", "s21"); REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s20", "s21");) }
    if (KIND == 4) { REP16(asm volatile("v_cmp_le_f32 vcc, %0, %4\n v_cmp_le_f32 vcc, %1, %4\n v_cmp_le_f32 vcc, %2, %4\n v_cmp_le_f32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
    if (KIND == 5) { REP16(asm volatile("v_cmp_le_f32 s[20:21], %0, %4\n v_cmp_le_f32 s[22:23], %1, %4\n v_cmp_le_f32 s[24:25], %2, %4\n v_cmp_le_f32 s[26:27], %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s20","s21","s22","s23","s24","s25","s26","s27");) }
    if (KIND == 6) { REP16(asm volatile("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
    if (KIND == 7) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (KIND == 8) { REP16(asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (KIND == 9) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n s_and_b64 s[20:21], s[20:21], s[22:23]\n v_fma_f32 %1, %1, %4, %5\n s_and_b64 s[20:21], s[20:21], s[22:23]\n v_fma_f32 %2, %2, %4, %5\n s_and_b64 s[20:21], s[20:21], s[22:23]\n v_fma_f32 %3, %3, %4, %5\n s_and_b64 s[20:21], s[20:21], s[22:23]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "s20","s21","scc");) }
    if (KIND == 10) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));) }
    if (KIND == 11) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));) }
    if (KIND == 12) { REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));) }
    if (KIND == 13) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5 op_sel_hi:[1,0,0]\n v_pk_fma_f32 %1, %1, %4, %5 op_sel_hi:[1,0,0]\n v_pk_fma_f32 %2, %2, %4, %5 op_sel_hi:[1,0,0]\n v_pk_fma_f32 %3, %3, %4, %5 op_sel_hi:[1,0,0]" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));) }
    if (KIND == 14) { REP16(asm volatile("v_max_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_min_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
    if (KIND == 15) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_min_f32 %1, %1, %4\n v_fma_f32 %2, %2, %4, %5\n v_min_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if (KIND == 16) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_exp_f32 %1, %1\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p1.y + p2.x + p3.y;
}
template <int KIND> void run(const char* name, float* d, int threads) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, blocks = 256;
  k<KIND><<<blocks, threads>>>(d, 10, 1.f);
  hipDeviceSynchronize();
  float best = 1e9;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0); k<KIND><<<blocks, threads>>>(d, iters, 1.f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
  }
  double insts_per_wave = (double)iters * 64;            // 16 x 4 instructions per iteration (KIND 9: +64 SALU)
  double waves_per_simd = threads / 256.0;
  double cyc = best * 1e-3 * 2.4e9 / (insts_per_wave * waves_per_simd);
  printf("%-34s %4d thr/CU: %.2f cycles per VALU wave-instruction per SIMD (at 2.4 GHz nominal)\n", name, threads, cyc);
}
int main() {
  float* d; hipMalloc(&d, 256 * 1024 * 4);
  for (int threads : {256, 1024}) {
    run<0>("v_fma_f32", d, threads); run<1>("v_mul_f32", d, threads); run<6>("v_min_f32", d, threads);
    run<2>("v_exp_f32", d, threads); run<7>("v_rcp_f32", d, threads);
    run<3>("v_cndmask_b32 (sgpr mask)", d, threads); run<4>("v_cmp_le_f32 -> vcc", d, threads);
    run<5>("v_cmp_le_f32 -> sgpr pair", d, threads); run<8>("v_add_f32 dpp quad_perm", d, threads);
    run<9>("v_fma_f32 + s_and_b64 interleaved", d, threads);
    run<10>("v_pk_fma_f32", d, threads); run<11>("v_pk_mul_f32", d, threads); run<12>("v_pk_add_f32", d, threads);
    run<13>("v_pk_fma_f32 op_sel_hi broadcast", d, threads);
    run<14>("v_max/v_min alternating", d, threads); run<15>("v_fma/v_min alternating", d, threads);
    run<16>("3 v_fma + 1 v_exp", d, threads);
  }
  return 0;
}
