// micro-benchmark (round 3): can the matrix pipe take the FMAs of the blend off the vector ALU?
//
// The forward blend costs 21 VALU instructions per (Gaussian, 64 pixels) = 65 SIMD-cycles (valu_issue.hip).
// 7 of them evaluate the conic's quadratic form and 4 accumulate the colour channels.  Both are rank-1
// updates  D[pixel][i] += A[i] * B[pixel]  -- exactly what v_mfma_f32_4x4x1_16b_f32 does for 16 blocks of
// 4 pixels (one pixel per lane) and i = 0..3, in exact f32 (one rounding per product, bitwise an fmaf
// chain).  With CBSZ = 4 the A operand of all 16 blocks is taken from the four lanes 4*ABID .. 4*ABID+3:
//   power:  6 MFMAs give  q0 + q1 x + q2 y + q3 x^2 + q4 xy + q5 y^2  for 4 Gaussians x 64 pixels, the q_k of
//           Gaussian g living in lane g of a register (lane = Gaussian, as the cull leaves them);
//   colour: 1 MFMA gives C[pixel][0..3] += w[pixel] * feat[g][0..3], feat of Gaussian g in lanes 4g' .. 4g'+3.
// What remains on the vector ALU per pair: exp2, two compares, three selects, one fma, one mul = 8.
//
// This file (1) checks the operand layout and bitwise equality with the fmaf chain, (2) times
//   0  the shipped blend body (21 VALU)                      3  VALU part alone (8 per pair)
//   1  8 VALU + 2.5 MFMA per pair (power + colour on MFMA)   4  MFMA part alone (2.5 per pair)
//   2  12 VALU + 1.5 MFMA per pair (power on MFMA)
// in real shader cycles per pair per SIMD at 2 / 4 / 8 waves per SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_blend.bin mfma_blend.hip && ./mfma_blend.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

#define MFMA(a, b, c, abid) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 4, (abid), 0)

// ---- (1) layout + bit check -----------------------------------------------------------------------
// lane = pixel (x = lane & 7, y = lane >> 3, centred), q[k][g] for 64 Gaussians; out[g][lane] = power
__global__ void layout_kernel(const float* __restrict__ q, float* __restrict__ out_mfma, float* __restrict__ out_fma,
                              const float* __restrict__ feat, const float* __restrict__ w, float* __restrict__ col_mfma,
                              float* __restrict__ col_fma) {
  const unsigned lane = threadIdx.x;
  const float x = (float)(lane & 7) - 3.5f, y = (float)(lane >> 3) - 3.5f;
  const float one = 1.0f, xx = x * x, xy = x * y, yy = y * y;
  float qk[6];
  for (int k = 0; k < 6; ++k) qk[k] = q[k * 64 + lane];                 // lane = Gaussian
#define GROUP(n)                                                                                   \
  {                                                                                                \
    f4 p = {0.f, 0.f, 0.f, 0.f};                                                                   \
    p = MFMA(qk[0], one, p, n); p = MFMA(qk[1], x, p, n); p = MFMA(qk[2], y, p, n);                \
    p = MFMA(qk[3], xx, p, n); p = MFMA(qk[4], xy, p, n); p = MFMA(qk[5], yy, p, n);               \
    for (int i = 0; i < 4; ++i) out_mfma[(4 * n + i) * 64 + lane] = p[i];                          \
  }
  GROUP(0) GROUP(1) GROUP(2) GROUP(3) GROUP(4) GROUP(5) GROUP(6) GROUP(7)
  GROUP(8) GROUP(9) GROUP(10) GROUP(11) GROUP(12) GROUP(13) GROUP(14) GROUP(15)
#undef GROUP
  for (int g = 0; g < 64; ++g) {
    float p = fmaf(q[0 * 64 + g], one, 0.f);
    p = fmaf(q[1 * 64 + g], x, p); p = fmaf(q[2 * 64 + g], y, p);
    p = fmaf(q[3 * 64 + g], xx, p); p = fmaf(q[4 * 64 + g], xy, p); p = fmaf(q[5 * 64 + g], yy, p);
    out_fma[g * 64 + lane] = p;
  }
  // colour: feat[g][c] flat = 4 g + c; register v holds floats 64 v + lane (Gaussian 16 v + lane / 4, channel lane % 4)
  float fr[4];
  for (int v = 0; v < 4; ++v) fr[v] = feat[64 * v + lane];
  f4 c = {0.f, 0.f, 0.f, 0.f};
#define COL(v, n) c = MFMA(fr[v], w[(16 * v + n) * 64 + lane], c, n);
#define COL16(v) COL(v, 0) COL(v, 1) COL(v, 2) COL(v, 3) COL(v, 4) COL(v, 5) COL(v, 6) COL(v, 7) \
                 COL(v, 8) COL(v, 9) COL(v, 10) COL(v, 11) COL(v, 12) COL(v, 13) COL(v, 14) COL(v, 15)
  COL16(0) COL16(1) COL16(2) COL16(3)
#undef COL16
#undef COL
  float cf[4] = {0.f, 0.f, 0.f, 0.f};
  for (int g = 0; g < 64; ++g)
    for (int i = 0; i < 4; ++i) cf[i] = fmaf(feat[4 * g + i], w[g * 64 + lane], cf[i]);
  for (int i = 0; i < 4; ++i) { col_mfma[i * 64 + lane] = c[i]; col_fma[i * 64 + lane] = cf[i]; }
}

// ---- (2) timing -----------------------------------------------------------------------------------
struct Px { float T, C0, C1, C2, C3; };

// the 8 VALU instructions that stay: exp2, cmp, select, fma, cmp, mul, select, select
__device__ __forceinline__ float blend_w(float& T, float power) {
  const float alpha = __builtin_amdgcn_exp2f(power);
  const float a_eff = alpha >= (1.0f / 255.0f) ? alpha : 0.f;
  const float next_T = fmaf(-a_eff, T, T);
  const bool acc = next_T > 1e-4f;
  float w = a_eff * T;
  w = acc ? w : 0.f;
  T = acc ? next_T : -fabsf(T);
  return w;
}

template <int KIND, int WPS>
__global__ __launch_bounds__(256, WPS) void time_kernel(const float* __restrict__ q, const float* __restrict__ feat,
                                                    float* __restrict__ out, unsigned long long* __restrict__ times,
                                                    int iters) {
  const unsigned lane = threadIdx.x & 63u;
  const float x = (float)(lane & 7) - 3.5f, y = (float)(lane >> 3) - 3.5f;
  float one = 1.0f, xx = x * x, xy = x * y, yy = y * y;
  asm volatile("" : "+v"(one));                      // keep it in a VGPR
  float qk[6], fr[4];
  for (int k = 0; k < 6; ++k) qk[k] = q[k * 64 + lane];
  for (int v = 0; v < 4; ++v) fr[v] = feat[64 * v + lane];
  float T = 1.f;
  f4 C = {0.f, 0.f, 0.f, 0.f};
  float mx = q[lane & 7], my = q[8 + (lane & 7)];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    // keep the compiler from hoisting anything out of the loop
    asm volatile("" : "+v"(qk[0]), "+v"(qk[1]), "+v"(qk[2]), "+v"(qk[3]), "+v"(qk[4]), "+v"(qk[5]));
    asm volatile("" : "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]));
    if constexpr (KIND == 0) {
      // shipped body: 21 VALU per pair, Gaussian parameters as uniform values (stand-ins: readlane'd SGPRs
      // would be cheaper than the real broadcast LDS reads; this is the VALU cost only)
#pragma unroll
      for (int g = 0; g < 64; ++g) {
        const float A = qk[3], B = qk[4], Cc = qk[5], op = qk[0];
        const float dx = mx - x, dy = my - y;
        const float power = fmaf(dx, fmaf(B, dy, A * dx), (Cc * dy) * dy);
        const float ov = op * __builtin_amdgcn_exp2f(power);
        const float alpha = ov;
        const bool valid = alpha >= (1.0f / 255.0f);
        const float a_eff = valid ? alpha : 0.f;
        const float next_T = fmaf(-a_eff, T, T);
        const bool acc = next_T > 1e-4f;
        float w = a_eff * T;
        w = acc ? w : 0.f;
        C[0] = fmaf(w, fr[0], C[0]); C[1] = fmaf(w, fr[1], C[1]); C[2] = fmaf(w, fr[2], C[2]); C[3] = fmaf(w, fr[3], C[3]);
        T = acc ? next_T : -fabsf(T);
        asm volatile("" : "+v"(mx), "+v"(my));
      }
    } else {
#define PAIR(i, v, n4)                                                                   \
      {                                                                                  \
        float w;                                                                         \
        if constexpr (KIND != 4) w = blend_w(T, p[i]); else w = p[i];                    \
        if constexpr (KIND == 1 || KIND == 4) C = MFMA(fr[v], w, C, n4 + i);             \
        if constexpr (KIND == 2) { C[0] = fmaf(w, fr[0], C[0]); C[1] = fmaf(w, fr[1], C[1]); C[2] = fmaf(w, fr[2], C[2]); C[3] = fmaf(w, fr[3], C[3]); } \
        if constexpr (KIND == 3) C[0] += w;                                              \
      }
#define POWER(p, n)                                                                      \
      if constexpr (KIND != 3) {                                                         \
        p = MFMA(qk[0], one, zero4, n); p = MFMA(qk[1], x, p, n); p = MFMA(qk[2], y, p, n);  \
        p = MFMA(qk[3], xx, p, n); p = MFMA(qk[4], xy, p, n); p = MFMA(qk[5], yy, p, n);     \
      } else { p[0] = qk[0]; p[1] = qk[1]; p[2] = qk[2]; p[3] = qk[3]; asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3])); }
      // software pipeline: the power of group n + 1 is issued with the blend of group n; nothing moves
      // across the stage boundary (sched_barrier), so the live state stays at two power registers sets
#define GROUP(n)                                                                         \
      {                                                                                  \
        f4 p = pn;                                                                       \
        if constexpr ((n) < 15) { POWER(pn, ((n) + 1) & 15) }                            \
        PAIR(0, (n) / 4, 4 * ((n) % 4)) PAIR(1, (n) / 4, 4 * ((n) % 4))                  \
        PAIR(2, (n) / 4, 4 * ((n) % 4)) PAIR(3, (n) / 4, 4 * ((n) % 4))                  \
        __builtin_amdgcn_sched_barrier(0);                                               \
      }
      const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
      f4 pn;
      POWER(pn, 0)
      GROUP(0) GROUP(1) GROUP(2) GROUP(3) GROUP(4) GROUP(5) GROUP(6) GROUP(7)
      GROUP(8) GROUP(9) GROUP(10) GROUP(11) GROUP(12) GROUP(13) GROUP(14) GROUP(15)
#undef POWER
#undef GROUP
#undef PAIR
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = T + C[0] + C[1] + C[2] + C[3];
  if (lane == 0) {
    const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    times[2 * w] = t1 - t0;
    times[2 * w + 1] = r1 - r0;
  }
}


// ---- (3) do the two pipes overlap ACROSS waves of one SIMD? ----------------------------------------
// 512-thread workgroups, one per CU (100 KiB of LDS each): waves w and w + 4 share a SIMD.  Waves 0-3 run the
// VALU part (8 per pair), waves 4-7 the MFMA part (2.5 per pair); MODE 1 / 2 let only one half work.
template <int MODE>
__global__ __launch_bounds__(512) void co_kernel(const float* __restrict__ q, const float* __restrict__ feat,
                                                 float* __restrict__ out, unsigned long long* __restrict__ times, int iters) {
  extern __shared__ float pad[];
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (q == nullptr) pad[threadIdx.x] = 0.f;
  const bool valu_wave = wave < 4;
  if ((MODE == 1 && !valu_wave) || (MODE == 2 && valu_wave)) return;
  const float x = (float)(lane & 7) - 3.5f, y = (float)(lane >> 3) - 3.5f;
  float one = 1.0f, xx = x * x, xy = x * y, yy = y * y;
  asm volatile("" : "+v"(one));
  float qk[6], fr[4];
  for (int k = 0; k < 6; ++k) qk[k] = q[k * 64 + lane];
  for (int v = 0; v < 4; ++v) fr[v] = feat[64 * v + lane];
  float T = 1.f;
  f4 C = {0.f, 0.f, 0.f, 0.f};
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (valu_wave) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 64; ++g) {
        float p = qk[g & 3];
        asm volatile("" : "+v"(p));
        C[0] += blend_w(T, p);
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      asm volatile("" : "+v"(qk[0]), "+v"(qk[1]), "+v"(qk[2]), "+v"(qk[3]), "+v"(qk[4]), "+v"(qk[5]));
#define GROUP(n)                                                                                    \
      {                                                                                             \
        f4 p = MFMA(qk[0], one, zero4, n); p = MFMA(qk[1], x, p, n); p = MFMA(qk[2], y, p, n);      \
        p = MFMA(qk[3], xx, p, n); p = MFMA(qk[4], xy, p, n); p = MFMA(qk[5], yy, p, n);            \
        C = MFMA(fr[(n) / 4], p[0], C, 4 * ((n) % 4)); C = MFMA(fr[(n) / 4], p[1], C, 4 * ((n) % 4) + 1); \
        C = MFMA(fr[(n) / 4], p[2], C, 4 * ((n) % 4) + 2); C = MFMA(fr[(n) / 4], p[3], C, 4 * ((n) % 4) + 3); \
      }
      GROUP(0) GROUP(1) GROUP(2) GROUP(3) GROUP(4) GROUP(5) GROUP(6) GROUP(7)
      GROUP(8) GROUP(9) GROUP(10) GROUP(11) GROUP(12) GROUP(13) GROUP(14) GROUP(15)
#undef GROUP
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = T + C[0] + C[1] + C[2] + C[3];
  if (lane == 0) times[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
void run_co(const float* q, const float* feat, float* d, unsigned long long* dt) {
  const int iters = 200;
  hipFuncSetAttribute((const void*)co_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  co_kernel<MODE><<<256, 512, 100 * 1024>>>(q, feat, d, dt, 5);
  hipDeviceSynchronize();
  hipMemset(dt, 0, 256 * 8 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); co_kernel<MODE><<<256, 512, 100 * 1024>>>(q, feat, d, dt, iters); hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), dt, h.size() * 8, hipMemcpyDeviceToHost);
  double sv = 0, sm = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? sv : sm) += (double)h[b * 8 + w];
  printf("co-residency mode %d (%s): VALU waves %7.2f cycles per pair, MFMA waves %7.2f cycles per pair, launch %.3f ms\n", MODE,
         MODE == 0 ? "both halves work" : MODE == 1 ? "VALU waves only" : "MFMA waves only", sv / 1024 / (iters * 64.0),
         sm / 1024 / (iters * 64.0), ms);
}

static const char* kNames[] = {"shipped blend body, 21 VALU", "8 VALU + 2.5 MFMA (power + colour on the matrix pipe)",
                               "12 VALU + 1.5 MFMA (power on the matrix pipe)", "VALU part alone (8 per pair)",
                               "MFMA part alone (2.5 per pair)"};

template <int KIND, int WPS>
void run(const float* q, const float* feat, float* d, unsigned long long* dt) {
  // 256 CUs x WPS workgroups of four waves (one per SIMD); the register budget is capped for WPS waves per SIMD
  const int wps = WPS, threads = 256, blocks = 256 * WPS;
  const int iters = 200;
  time_kernel<KIND, WPS><<<blocks, threads>>>(q, feat, d, dt, 5);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t nw = (size_t)blocks * threads / 64;
  std::vector<unsigned long long> h(2 * nw);
  double best = 1e30, best_ms = 0, mhz = 0;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0); time_kernel<KIND, WPS><<<blocks, threads>>>(q, feat, d, dt, iters); hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), dt, 2 * nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sc = 0, sr = 0;
    for (size_t w = 0; w < nw; ++w) { sc += (double)h[2 * w]; sr += (double)h[2 * w + 1]; }
    const double cyc = sc / nw / ((double)iters * 64 * wps);
    if (cyc < best) { best = cyc; best_ms = ms; mhz = sc / sr * 100.0; }
  }
  // wall-clock view: the whole launch / (pairs per SIMD), in ns, independent of the in-wave clock reads
  const double ns_pair = best_ms * 1e6 / ((double)iters * 64 * wps);
  printf("%-56s %d w/SIMD: %6.2f shader cycles per pair per SIMD (clock %4.0f MHz; %5.2f ns wall per pair per SIMD)\n",
         kNames[KIND], wps, best, mhz, ns_pair);
  fflush(stdout);
}

template <int KIND>
void sweep(const float* q, const float* feat, float* d, unsigned long long* dt) {
  run<KIND, 1>(q, feat, d, dt); run<KIND, 2>(q, feat, d, dt); run<KIND, 4>(q, feat, d, dt); if (KIND != 2) { run<KIND, 6>(q, feat, d, dt); run<KIND, 8>(q, feat, d, dt); }
}

int main() {
  // ---- layout / bit check
  std::vector<float> hq(6 * 64), hf(256), hw(64 * 64);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& v : hq) v = rnd() * 3.f;
  for (auto& v : hf) v = rnd();
  for (auto& v : hw) v = rnd();
  float *q, *om, *of, *feat, *w, *cm, *cf;
  hipMalloc(&q, hq.size() * 4); hipMalloc(&om, 64 * 64 * 4); hipMalloc(&of, 64 * 64 * 4);
  hipMalloc(&feat, 256 * 4); hipMalloc(&w, 64 * 64 * 4); hipMalloc(&cm, 256 * 4); hipMalloc(&cf, 256 * 4);
  hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(feat, hf.data(), 256 * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), 64 * 64 * 4, hipMemcpyHostToDevice);
  layout_kernel<<<1, 64>>>(q, om, of, feat, w, cm, cf);
  std::vector<float> a(64 * 64), b(64 * 64), c(256), d2(256);
  hipMemcpy(a.data(), om, a.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), of, b.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), cm, 256 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(d2.data(), cf, 256 * 4, hipMemcpyDeviceToHost);
  int bad = 0, badc = 0;
  for (size_t i = 0; i < a.size(); ++i) bad += memcmp(&a[i], &b[i], 4) != 0;
  for (int i = 0; i < 256; ++i) badc += memcmp(&c[i], &d2[i], 4) != 0;
  printf("layout check: power %d of 4096 values differ from the fmaf chain, colour %d of 256 (0 = layout and rounding as assumed)\n",
         bad, badc);
  if (bad) printf("  e.g. mfma %g vs fma %g\n", a[0], b[0]);

  float* d; hipMalloc(&d, 512 * 1024 * 4);
  unsigned long long* dt; hipMalloc(&dt, 2 * 512 * 16 * 8);
  run_co<1>(q, feat, d, dt); run_co<2>(q, feat, d, dt); run_co<0>(q, feat, d, dt);
  sweep<0>(q, feat, d, dt);
  sweep<1>(q, feat, d, dt);
  sweep<2>(q, feat, d, dt);
  sweep<3>(q, feat, d, dt);
  sweep<4>(q, feat, d, dt);
  return 0;
}
