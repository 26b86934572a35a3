// micro-benchmark (round 2): VALU issue cost on gfx950 in REAL shader cycles.
//
// Round 1's valu_cost.hip converted hipEvent time at a nominal 2.4 GHz and reported 3.0 "cycles" for
// v_fma_f32 against the guide's 2 (MI355X_MICROARCH.md per-instruction table).  This version reads the
// shader clock itself (s_memtime) and the constant 100 MHz clock (s_memrealtime) in every wave, uses
// 8 independent accumulators, VGPR-only operands, and sweeps 1 / 2 / 4 / 8 waves per SIMD, so that
//   cycles per wave64 instruction per SIMD = wave duration in shader cycles / (instructions x waves/SIMD)
// and the effective clock = d(s_memtime) / d(s_memrealtime) x 100 MHz are both measured, not assumed.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue.bin valu_issue.hip && ./valu_issue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define REP8(X) X X X X X X X X

// eight independent chains: a0..a7 are %0..%7; %8 = b (vgpr), %9 = c (vgpr), %10 = sb (sgpr)
#define OPERANDS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(sb)
#define CHAIN8(PRE, POST)                                                                       \
  asm volatile(PRE " %0, %0" POST "\n" PRE " %1, %1" POST "\n" PRE " %2, %2" POST "\n" PRE      \
               " %3, %3" POST "\n" PRE " %4, %4" POST "\n" PRE " %5, %5" POST "\n" PRE          \
               " %6, %6" POST "\n" PRE " %7, %7" POST OPERANDS)
// compares: no VGPR destination
#define CMP8(PRE, DST0, DST1, DST2, DST3)                                                       \
  asm volatile(PRE " " DST0 ", %0, %8\n" PRE " " DST1 ", %1, %8\n" PRE " " DST2 ", %2, %8\n" PRE \
               " " DST3 ", %3, %8\n" PRE " " DST0 ", %4, %8\n" PRE " " DST1 ", %5, %8\n" PRE     \
               " " DST2 ", %6, %8\n" PRE " " DST3 ", %7, %8" OPERANDS                          \
               : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")

enum Kind {
  K_FMA, K_MUL, K_ADD, K_SUB, K_FMAC, K_MUL_CLAMP, K_FMA_CLAMP, K_FMA_ABSNEG, K_FMA_SGPR, K_FMA_LIT,
  K_MIN, K_MAX, K_MED3, K_CMP_VCC, K_CMP_SGPR, K_CNDMASK_VCC, K_CNDMASK_SGPR, K_AND, K_OR, K_XOR, K_MOV,
  K_ADD_U32, K_LSHL, K_CVT, K_EXP, K_RCP, K_LOG, K_DPP_MOV, K_DPP_ADD, K_PK_FMA, K_PK_MUL, K_PK_ADD,
  K_MAX3, K_MIN3, K_LDEXP, K_FRACT, K_BFE, K_MAD_U24, K_MUL_LEGACY, K_SUBREV, K_MAC_MIX_MIN,
  K_MIX_FMA_CMP, K_MIX_FMA_CND, K_MIX_FMA_EXP, K_BLEND_NOW, K_BLEND_CLAMP, K_BLEND_R3, K_BLEND_MASK, K_COUNT
};
static const char* kNames[] = {
  "v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fmac_f32 (VOP2)", "v_mul_f32 clamp (VOP3)",
  "v_fma_f32 clamp", "v_fma_f32 |abs| -neg", "v_fma_f32 sgpr operand", "v_fmaak_f32 (VOP2, literal K)",
  "v_min_f32", "v_max_f32", "v_med3_f32", "v_cmp_le_f32 -> vcc", "v_cmp_le_f32 -> sgpr pair",
  "v_cndmask_b32 vcc", "v_cndmask_b32 sgpr mask", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32",
  "v_add_u32", "v_lshlrev_b32", "v_cvt_f32_i32", "v_exp_f32", "v_rcp_f32", "v_log_f32", "v_mov_b32 dpp",
  "v_add_f32 dpp", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_max3_f32", "v_min3_f32",
  "v_ldexp_f32", "v_fract_f32", "v_bfe_u32", "v_mad_u32_u24", "v_mul_legacy_f32", "v_subrev_f32",
  "mix: 4 fma + 4 min", "mix: 6 fma + 2 cmp", "mix: 6 fma + 2 cndmask", "mix: 7 fma + 1 exp",
  "blend body as shipped (13 fma-class, 7 cmp/sel/min, 1 exp)", "blend body, clamp/step form (18 fma-class, 1 exp, 2 other)",
  "round-3 blend, 4 ch, sign of T as the flag (11 fma-class, 1 cmpx, 1 cmp, 2 cndmask, 1 exp)",
  "round-3 blend, 4 ch, lane masks (12 fma-class, 2 cmpx, 1 exp; 4 salu)"
};

__device__ __forceinline__ unsigned long long shader_clock() { return __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ unsigned long long real_clock() { return __builtin_amdgcn_s_memrealtime(); }

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* times, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,
        a7 = a0 + 7, b = 1.0001f, c = 0.5f;
  float sb = seed * 0.999f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a1, a2}, p2 = {a2, a3}, p3 = {a3, a0}, pb = {b, b}, pc = {c, c};
  asm volatile("s_mov_b64 s[20:21], 0x0f0f0f0f\n s_mov_b64 s[22:23], 0x33333333" ::: "s20", "s21", "s22", "s23");
  asm volatile("s_mov_b64 vcc, 0x55555555" ::: "vcc");
  const unsigned long long t0 = shader_clock(), r0 = real_clock();
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == K_FMA) { REP8(CHAIN8("v_fma_f32", ", %8, %9");) }
    if constexpr (KIND == K_MUL) { REP8(CHAIN8("v_mul_f32", ", %8");) }
    if constexpr (KIND == K_ADD) { REP8(CHAIN8("v_add_f32", ", %8");) }
    if constexpr (KIND == K_SUB) { REP8(CHAIN8("v_sub_f32", ", %8");) }
    if constexpr (KIND == K_SUBREV) { REP8(CHAIN8("v_subrev_f32", ", %8");) }
    if constexpr (KIND == K_FMAC) { REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9" OPERANDS);) }
    if constexpr (KIND == K_MUL_CLAMP) { REP8(CHAIN8("v_mul_f32_e64", ", %8 clamp");) }
    if constexpr (KIND == K_FMA_CLAMP) { REP8(CHAIN8("v_fma_f32", ", %8, %9 clamp");) }
    if constexpr (KIND == K_FMA_ABSNEG) { REP8(asm volatile("v_fma_f32 %0, |%0|, -%8, %9\n v_fma_f32 %1, |%1|, -%8, %9\n v_fma_f32 %2, |%2|, -%8, %9\n v_fma_f32 %3, |%3|, -%8, %9\n v_fma_f32 %4, |%4|, -%8, %9\n v_fma_f32 %5, |%5|, -%8, %9\n v_fma_f32 %6, |%6|, -%8, %9\n v_fma_f32 %7, |%7|, -%8, %9" OPERANDS);) }
    if constexpr (KIND == K_FMA_SGPR) { REP8(CHAIN8("v_fma_f32", ", %10, %9");) }
    if constexpr (KIND == K_FMA_LIT) { REP8(CHAIN8("v_fmaak_f32", ", %8, 0x3f7fbe77");) }
    if constexpr (KIND == K_MIN) { REP8(CHAIN8("v_min_f32", ", %8");) }
    if constexpr (KIND == K_MAX) { REP8(CHAIN8("v_max_f32", ", %8");) }
    if constexpr (KIND == K_MED3) { REP8(CHAIN8("v_med3_f32", ", %8, %9");) }
    if constexpr (KIND == K_MAX3) { REP8(CHAIN8("v_max3_f32", ", %8, %9");) }
    if constexpr (KIND == K_MIN3) { REP8(CHAIN8("v_min3_f32", ", %8, %9");) }
    if constexpr (KIND == K_CMP_VCC) { REP8(CMP8("v_cmp_le_f32", "vcc", "vcc", "vcc", "vcc");) }
    if constexpr (KIND == K_CMP_SGPR) { REP8(CMP8("v_cmp_le_f32", "s[20:21]", "s[22:23]", "s[24:25]", "s[26:27]");) }
    if constexpr (KIND == K_CNDMASK_VCC) { REP8(CHAIN8("v_cndmask_b32", ", %8, vcc");) }
    if constexpr (KIND == K_CNDMASK_SGPR) { REP8(CHAIN8("v_cndmask_b32_e64", ", %8, s[20:21]");) }
    if constexpr (KIND == K_AND) { REP8(CHAIN8("v_and_b32", ", %8");) }
    if constexpr (KIND == K_OR) { REP8(CHAIN8("v_or_b32", ", %8");) }
    if constexpr (KIND == K_XOR) { REP8(CHAIN8("v_xor_b32", ", %8");) }
    if constexpr (KIND == K_MOV) { REP8(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" OPERANDS);) }
    if constexpr (KIND == K_ADD_U32) { REP8(CHAIN8("v_add_u32", ", %8");) }
    if constexpr (KIND == K_LSHL) { REP8(asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7" OPERANDS);) }
    if constexpr (KIND == K_CVT) { REP8(asm volatile("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n v_cvt_f32_i32 %4, %4\n v_cvt_f32_i32 %5, %5\n v_cvt_f32_i32 %6, %6\n v_cvt_f32_i32 %7, %7" OPERANDS);) }
    if constexpr (KIND == K_EXP) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" OPERANDS);) }
    if constexpr (KIND == K_RCP) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" OPERANDS);) }
    if constexpr (KIND == K_LOG) { REP8(asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7" OPERANDS);) }
    if constexpr (KIND == K_FRACT) { REP8(asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3\n v_fract_f32 %4, %4\n v_fract_f32 %5, %5\n v_fract_f32 %6, %6\n v_fract_f32 %7, %7" OPERANDS);) }
    if constexpr (KIND == K_DPP_MOV) { REP8(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" OPERANDS);) }
    if constexpr (KIND == K_DPP_ADD) { REP8(asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" OPERANDS);) }
    if constexpr (KIND == K_PK_FMA) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));) }
    if constexpr (KIND == K_PK_MUL) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));) }
    if constexpr (KIND == K_PK_ADD) { REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));) }
    if constexpr (KIND == K_LDEXP) { REP8(asm volatile("v_ldexp_f32 %0, %0, 1\n v_ldexp_f32 %1, %1, 1\n v_ldexp_f32 %2, %2, 1\n v_ldexp_f32 %3, %3, 1\n v_ldexp_f32 %4, %4, 1\n v_ldexp_f32 %5, %5, 1\n v_ldexp_f32 %6, %6, 1\n v_ldexp_f32 %7, %7, 1" OPERANDS);) }
    if constexpr (KIND == K_BFE) { REP8(asm volatile("v_bfe_u32 %0, %0, 1, 31\n v_bfe_u32 %1, %1, 1, 31\n v_bfe_u32 %2, %2, 1, 31\n v_bfe_u32 %3, %3, 1, 31\n v_bfe_u32 %4, %4, 1, 31\n v_bfe_u32 %5, %5, 1, 31\n v_bfe_u32 %6, %6, 1, 31\n v_bfe_u32 %7, %7, 1, 31" OPERANDS);) }
    if constexpr (KIND == K_MAD_U24) { REP8(CHAIN8("v_mad_u32_u24", ", %8, %9");) }
    if constexpr (KIND == K_MUL_LEGACY) { REP8(CHAIN8("v_mul_legacy_f32", ", %8");) }
    if constexpr (KIND == K_MAC_MIX_MIN) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_min_f32 %1, %1, %8\n v_fma_f32 %2, %2, %8, %9\n v_min_f32 %3, %3, %8\n v_fma_f32 %4, %4, %8, %9\n v_min_f32 %5, %5, %8\n v_fma_f32 %6, %6, %8, %9\n v_min_f32 %7, %7, %8" OPERANDS);) }
    if constexpr (KIND == K_MIX_FMA_CMP) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_cmp_le_f32 vcc, %3, %8\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_cmp_le_f32 vcc, %7, %8" OPERANDS : "vcc");) }
    if constexpr (KIND == K_MIX_FMA_CND) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, vcc\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, vcc" OPERANDS);) }
    if constexpr (KIND == K_MIX_FMA_EXP) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_exp_f32 %3, %3\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" OPERANDS);) }
    if constexpr (KIND == K_BLEND_NOW) {
      // the shipped blend_pixel, one quadrant, 3 channels, as the compiler emits it (dependency structure kept;
      // a0 = T, a1..a3 = C, a4 = px, a5 = py; b, c stand in for the broadcast Gaussian parameters)
      REP8(asm volatile(
          "v_sub_f32 %6, %8, %4\n v_sub_f32 %7, %9, %5\n"
          "v_mul_f32 v20, %8, %6\n v_fma_f32 v20, %9, %7, v20\n v_mul_f32 v21, %8, %7\n v_mul_f32 v21, v21, %7\n"
          "v_fma_f32 v20, %6, v20, v21\n v_exp_f32 v21, v20\n v_mul_f32 v21, %9, v21\n v_min_f32 v21, 0x3f7fbe77, v21\n"
          "v_cmp_ge_f32 s[20:21], 0, v20\n v_cmp_le_f32 vcc, 0x3b808081, v21\n s_and_b64 vcc, vcc, s[20:21]\n"
          "v_cndmask_b32 v21, 0, v21, vcc\n v_fma_f32 v22, -v21, %0, %0\n v_cmp_lt_f32 vcc, 0x38d1b717, v22\n"
          "v_mul_f32 v21, v21, %0\n v_cndmask_b32 v21, 0, v21, vcc\n"
          "v_fma_f32 %1, v21, %8, %1\n v_fma_f32 %2, v21, %9, %2\n v_fma_f32 %3, v21, %8, %3\n"
          "v_cndmask_b32_e64 %0, -|%0|, v22, vcc" OPERANDS : "vcc", "scc", "s20", "s21", "v20", "v21", "v22");)
    }
    if constexpr (KIND == K_BLEND_CLAMP) {
      // candidate: min(0.999,.) via VOP3 clamp on pre-scaled opacity, alpha >= 1/255 via a clamped step FMA,
      // accumulate test folded the same way (all FMA-class except the exp, one cmp and one cndmask for T)
      REP8(asm volatile(
          "v_sub_f32 %6, %8, %4\n v_sub_f32 %7, %9, %5\n"
          "v_mul_f32 v20, %8, %6\n v_fma_f32 v20, %9, %7, v20\n v_mul_f32 v21, %8, %7\n v_mul_f32 v21, v21, %7\n"
          "v_fma_f32 v20, %6, v20, v21\n v_exp_f32_e64 v21, v20 clamp\n v_mul_f32_e64 v21, %9, v21 clamp\n"
          "v_fma_f32 v23, v21, %8, %9 clamp\n v_mul_f32 v21, v21, v23\n"
          "v_mul_f32 v23, v21, %0\n v_fma_f32 v22, %8, v23, %0\n v_cmp_lt_f32 vcc, 0x38d1b717, v22\n"
          "v_fma_f32 v24, v22, %8, %9 clamp\n v_mul_f32 v23, v23, v24\n"
          "v_fma_f32 %1, v23, %8, %1\n v_fma_f32 %2, v23, %9, %2\n v_fma_f32 %3, v23, %8, %3\n"
          "v_cndmask_b32_e64 %0, -|%0|, v22, vcc" OPERANDS : "vcc", "v20", "v21", "v22", "v23", "v24");)
    }
    if constexpr (KIND == K_BLEND_R3) {
      // raster_fwd.hip blend_pixel_safe_asm before the lane masks: a0 = T, a1..a4 = C, a5 / a6 the pixel offsets;
      // the compares are unsigned "0 <= x" (always true: EXEC stays full, the worst case for the issue rate)
      REP8(asm volatile(
          "v_fma_f32 v20, %8, %5, %9\n v_fmac_f32 v20, %8, %6\n v_fmac_f32 v20, %9, %5\n v_fmac_f32 v20, %9, %6\n"
          "v_fmac_f32 v20, %8, %5\n v_exp_f32 v20, v20\n s_nop 0\n v_cmpx_le_u32 vcc, 0, v20\n"
          "v_fma_f32 v21, -v20, %0, %0\n v_mul_f32 v22, v20, %0\n v_cmp_le_u32 vcc, 0, v21\n s_nop 1\n"
          "v_cndmask_b32 v22, 0, v22, vcc\n v_cndmask_b32_e64 %0, -|%0|, v21, vcc\n"
          "v_fmac_f32 %1, v22, %8\n v_fmac_f32 %2, v22, %9\n v_fmac_f32 %3, v22, %8\n v_fmac_f32 %4, v22, %9\n"
          "s_mov_b64 exec, -1" OPERANDS : "vcc", "v20", "v21", "v22");)
    }
    if constexpr (KIND == K_BLEND_MASK) {
      // ... with the finished pixels in a lane mask (s[20:21], all ones here): EXEC narrowed three times, no select
      asm volatile("s_mov_b64 s[20:21], -1" ::: "s20", "s21");
      REP8(asm volatile(
          "s_mov_b64 exec, s[20:21]\n"
          "v_fma_f32 v20, %8, %5, %9\n v_fmac_f32 v20, %8, %6\n v_fmac_f32 v20, %9, %5\n v_fmac_f32 v20, %9, %6\n"
          "v_fmac_f32 v20, %8, %5\n v_exp_f32 v20, v20\n s_nop 0\n v_cmpx_le_u32 vcc, 0, v20\n"
          "v_fma_f32 v21, -v20, %0, %0\n v_mul_f32 v22, v20, %0\n v_cmpx_le_u32_e64 s[22:23], 0, v21\n"
          "v_mov_b32 %0, v21\n"
          "v_fmac_f32 %1, v22, %8\n v_fmac_f32 %2, v22, %9\n v_fmac_f32 %3, v22, %8\n v_fmac_f32 %4, v22, %9\n"
          "s_xor_b64 vcc, vcc, s[22:23]\n s_andn2_b64 s[20:21], s[20:21], vcc\n"
          "s_cselect_b32 vcc_lo, -1, -2\n s_and_b32 s24, s24, vcc_lo\n"
          "s_mov_b64 exec, -1" OPERANDS : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "v20", "v21", "v22");)
    }
  }
  const unsigned long long t1 = shader_clock(), r1 = real_clock();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y;
  if ((threadIdx.x & 63) == 0) {
    const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    times[2 * w] = t1 - t0;
    times[2 * w + 1] = r1 - r0;
  }
}

static int insts_per_iter(int kind) {
  if (kind == K_BLEND_NOW) return 8 * 21;       // VALU only (the s_and is extra)
  if (kind == K_BLEND_CLAMP) return 8 * 20;
  if (kind == K_BLEND_R3) return 8 * 16;
  if (kind == K_BLEND_MASK) return 8 * 15;
  return 64;
}

template <int KIND>
void run(float* d, unsigned long long* dt, int waves_per_simd) {
  // waves/SIMD w: 256 CUs x 4 SIMDs x w waves; blocks of 256*min(w,4) threads, (w+3)/4 blocks per CU
  const int threads = 256 * (waves_per_simd < 4 ? waves_per_simd : 4);
  const int blocks = 256 * ((waves_per_simd + 3) / 4);
  const int iters = 1500;
  k<KIND><<<blocks, threads>>>(d, dt, 10, 1.f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double best_cyc = 1e30, best_ms = 1e30, mhz = 0;
  const size_t nw = (size_t)blocks * threads / 64;
  std::vector<unsigned long long> h(2 * nw);
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0); k<KIND><<<blocks, threads>>>(d, dt, iters, 1.f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), dt, 2 * nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum_c = 0, sum_r = 0;
    for (size_t w = 0; w < nw; ++w) { sum_c += (double)h[2 * w]; sum_r += (double)h[2 * w + 1]; }
    const double cyc = sum_c / nw / ((double)iters * insts_per_iter(KIND) * waves_per_simd);
    if (cyc < best_cyc) { best_cyc = cyc; best_ms = ms; mhz = sum_c / sum_r * 100.0; }
  }
  const double nominal = best_ms * 1e-3 * 2.4e9 / ((double)iters * insts_per_iter(KIND) * waves_per_simd);
  printf("%-62s %d w/SIMD: %5.2f shader cycles/inst/SIMD  (clock %4.0f MHz; hipEvent@2.4GHz: %5.2f)\n", kNames[KIND],
         waves_per_simd, best_cyc, mhz, nominal);
  fflush(stdout);
}

template <int KIND>
void sweep(float* d, unsigned long long* dt, bool all) {
  if (all) { run<KIND>(d, dt, 1); run<KIND>(d, dt, 2); }
  run<KIND>(d, dt, 4);
  run<KIND>(d, dt, 8);
}

template <int K0>
void all_kinds(float* d, unsigned long long* dt) {
  if constexpr (K0 < K_COUNT) {
    sweep<K0>(d, dt, K0 == K_FMA || K0 == K_MIN || K0 == K_CMP_VCC || K0 == K_EXP || K0 == K_BLEND_NOW || K0 == K_BLEND_CLAMP || K0 == K_BLEND_R3 || K0 == K_BLEND_MASK);
    all_kinds<K0 + 1>(d, dt);
  }
}

int main(int argc, char** argv) {
  float* d; hipMalloc(&d, 512 * 1024 * 4);
  unsigned long long* dt; hipMalloc(&dt, 2 * 512 * 16 * 8);
  if (argc > 1 && !strcmp(argv[1], "blend")) {      // only the two blend bodies
    sweep<K_BLEND_NOW>(d, dt, true);
    sweep<K_BLEND_CLAMP>(d, dt, true);
    sweep<K_BLEND_R3>(d, dt, true);
    sweep<K_BLEND_MASK>(d, dt, true);
    return 0;
  }
  all_kinds<0>(d, dt);
  return 0;
}
