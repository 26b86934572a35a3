// micro-benchmark (round 5): how should the forward raster's inner loop be FED?
//
// The shipped raster_fwd_kernel walks a wave-private LDS queue: per queued Gaussian three broadcast ds_read_b128 bring 12
// wave-uniform operands (q0 q1 q2 A | B C mask idx | f0 f1 f2 f3) into 12 VGPRs, then up to four hand-written quadrant
// bodies run (raster_fwd.hip: blend_pixel_safe_asm, 15 vector instructions per 64 pairs).  PMC has its waves parked on that
// round trip (5 waves per SIMD at 96 VGPRs).  The round-4 review's untried lever: feed the operands through the SCALAR
// path -- the wave stores its culled batch to a per-wave ring in global memory (L2 resident), invalidates the scalar cache,
// and s_load_dwordx8 + x4 brings entry j + 1 into SGPRs while entry j is blended; the body then takes one scalar operand
// per instruction (gfx9: one constant-bus read per VALU op), which costs one v_mov more (16 instead of 15).
//
// This program prices exactly that, body + feed, with everything else of the kernel left out:
//   form L  LDS queue, 3 x ds_read_b128 per entry, the shipped body                       (what ships)
//   form S  global ring, vector stores + s_dcache_inv per batch, s_load one entry ahead, scalar-operand body
//   form S0 the same without the per-batch s_dcache_inv (NOT correct in a real kernel -- stale lines -- shows its cost)
//   form S2 form S with TWO entries in flight (three SGPR sets in turn)
// at 4, 5, 6 and 8 waves per SIMD (occupancy forced with dynamic LDS), on synthetic batches shaped like configs[1]:
// 54 queued entries per batch of 64, quadrant masks with 2.25 bits set on average, alpha >= 1/255 for ~40 % of the lanes.
// Reported: ns per (entry, quadrant) body per wave, SIMD-cycles per vector instruction, and the wall time of the launch.
//
//   hipcc --offload-arch=gfx950 -O3 -o queue_feed.bin queue_feed.hip && ./queue_feed.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f8 __attribute__((ext_vector_type(8)));

struct Px { float T, c0, c1, c2, c3; };
struct Poly { float x, y, xx, xy, yy; };

// ---- the shipped body: every operand in a VGPR -------------------------------------------------------------------
__device__ __forceinline__ void body_v(Px& p, unsigned long long& alive, unsigned& live, const Poly& pp, float q0, float q1,
                                       float q2, float A, float B, float C, float f0, float f1, float f2, float f3, int bit) {
  float dx, t0, t1;
  unsigned long long acc;
  const float amin = 1.0f / 255.0f, tstop = 1e-4f;
  const unsigned clr = ~(1u << bit);
  asm volatile(
      "s_mov_b64 exec, %[alive]\n"
      "v_fma_f32 %[t1], %[q1], %[x], %[q0]\n"
      "v_fmac_f32 %[t1], %[q2], %[y]\n"
      "v_fmac_f32 %[t1], %[A], %[xx]\n"
      "v_fmac_f32 %[t1], %[B], %[xy]\n"
      "v_fmac_f32 %[t1], %[C], %[yy]\n"
      "v_exp_f32 %[t1], %[t1]\n"
      "s_nop 0\n"
      "v_cmpx_le_f32 vcc, %[amin], %[t1]\n"
      "v_fma_f32 %[t0], -%[t1], %[T], %[T]\n"
      "v_mul_f32 %[dx], %[t1], %[T]\n"
      "v_cmpx_lt_f32_e64 %[acc], %[tstop], %[t0]\n"
      "v_mov_b32 %[T], %[t0]\n"
      "v_fmac_f32 %[c0], %[dx], %[f0]\n"
      "v_fmac_f32 %[c1], %[dx], %[f1]\n"
      "v_fmac_f32 %[c2], %[dx], %[f2]\n"
      "v_fmac_f32 %[c3], %[dx], %[f3]\n"
      "s_xor_b64 vcc, vcc, %[acc]\n"
      "s_andn2_b64 %[alive], %[alive], vcc\n"
      "s_cselect_b32 vcc_lo, -1, %[clr]\n"
      "s_and_b32 %[live], %[live], vcc_lo\n"
      "s_mov_b64 exec, -1\n"
      : [dx] "=&v"(dx), [t0] "=&v"(t0), [t1] "=&v"(t1), [acc] "=&s"(acc), [alive] "+s"(alive), [live] "+s"(live),
        [T] "+v"(p.T), [c0] "+v"(p.c0), [c1] "+v"(p.c1), [c2] "+v"(p.c2), [c3] "+v"(p.c3)
      : [q0] "v"(q0), [q1] "v"(q1), [q2] "v"(q2), [x] "v"(pp.x), [y] "v"(pp.y), [xx] "v"(pp.xx), [xy] "v"(pp.xy), [yy] "v"(pp.yy),
        [A] "v"(A), [B] "v"(B), [C] "v"(C), [f0] "v"(f0), [f1] "v"(f1), [f2] "v"(f2), [f3] "v"(f3), [amin] "s"(amin),
        [tstop] "s"(tstop), [clr] "s"(clr)
      : "vcc", "scc");
}

// ---- the same arithmetic with the entry's operands in SGPRs: one scalar operand per instruction, so q0 needs a move ----
__device__ __forceinline__ void body_s(Px& p, unsigned long long& alive, unsigned& live, const Poly& pp, float q0, float q1,
                                       float q2, float A, float B, float C, float f0, float f1, float f2, float f3, int bit) {
  float dx, t0, t1;
  unsigned long long acc;
  const float amin = 1.0f / 255.0f, tstop = 1e-4f;
  const unsigned clr = ~(1u << bit);
  asm volatile(
      "s_mov_b64 exec, %[alive]\n"
      "v_mov_b32 %[t1], %[q0]\n"
      "v_fmac_f32 %[t1], %[q1], %[x]\n"
      "v_fmac_f32 %[t1], %[q2], %[y]\n"
      "v_fmac_f32 %[t1], %[A], %[xx]\n"
      "v_fmac_f32 %[t1], %[B], %[xy]\n"
      "v_fmac_f32 %[t1], %[C], %[yy]\n"
      "v_exp_f32 %[t1], %[t1]\n"
      "s_nop 0\n"
      "v_cmpx_le_f32 vcc, %[amin], %[t1]\n"
      "v_fma_f32 %[t0], -%[t1], %[T], %[T]\n"
      "v_mul_f32 %[dx], %[t1], %[T]\n"
      "v_cmpx_lt_f32_e64 %[acc], %[tstop], %[t0]\n"
      "v_mov_b32 %[T], %[t0]\n"
      "v_fmac_f32 %[c0], %[f0], %[dx]\n"
      "v_fmac_f32 %[c1], %[f1], %[dx]\n"
      "v_fmac_f32 %[c2], %[f2], %[dx]\n"
      "v_fmac_f32 %[c3], %[f3], %[dx]\n"
      "s_xor_b64 vcc, vcc, %[acc]\n"
      "s_andn2_b64 %[alive], %[alive], vcc\n"
      "s_cselect_b32 vcc_lo, -1, %[clr]\n"
      "s_and_b32 %[live], %[live], vcc_lo\n"
      "s_mov_b64 exec, -1\n"
      : [dx] "=&v"(dx), [t0] "=&v"(t0), [t1] "=&v"(t1), [acc] "=&s"(acc), [alive] "+s"(alive), [live] "+s"(live),
        [T] "+v"(p.T), [c0] "+v"(p.c0), [c1] "+v"(p.c1), [c2] "+v"(p.c2), [c3] "+v"(p.c3)
      : [q0] "s"(q0), [q1] "s"(q1), [q2] "s"(q2), [x] "v"(pp.x), [y] "v"(pp.y), [xx] "v"(pp.xx), [xy] "v"(pp.xy), [yy] "v"(pp.yy),
        [A] "s"(A), [B] "s"(B), [C] "s"(C), [f0] "s"(f0), [f1] "s"(f1), [f2] "s"(f2), [f3] "s"(f3), [amin] "s"(amin),
        [tstop] "s"(tstop), [clr] "s"(clr)
      : "vcc", "scc");
}

struct Entry { float q0, q1, q2, A, B, C; unsigned mask; int idx; float f0, f1, f2, f3; float pad[4]; };   // 64 bytes

// synthetic entry of batch b for queue position j (wave-uniform function of (b, j, seed)): exponent in [-9, 0] so that
// ~40 % of the lanes pass alpha >= 1/255 with opacity ~0.5, masks with 2.25 bits on average
__device__ __forceinline__ Entry make_entry(unsigned b, unsigned j, unsigned seed) {
  unsigned h = (b * 64u + j) * 2654435761u + seed * 40503u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  Entry e;
  const float u0 = (float)(h & 1023u) / 1023.f, u1 = (float)((h >> 10) & 1023u) / 1023.f, u2 = (float)((h >> 20) & 1023u) / 1023.f;
  e.A = -0.02f - 0.05f * u0; e.C = -0.02f - 0.05f * u1; e.B = 0.02f * (u2 - 0.5f);
  const float mx = 14.f * (u1 - 0.5f), my = 14.f * (u2 - 0.5f);
  e.q0 = e.A * mx * mx + e.B * mx * my + e.C * my * my - 1.2f;
  e.q1 = -(2.f * e.A * mx + e.B * my);
  e.q2 = -(2.f * e.C * my + e.B * mx);
  const unsigned r = (h >> 7) & 15u;                    // 16 patterns: 3 x one bit... average 2.25 bits
  const unsigned masks[16] = {1, 2, 4, 8, 3, 12, 5, 10, 3, 12, 7, 11, 13, 14, 15, 15};
  e.mask = masks[r];
  e.idx = (int)(b * 64u + j);
  e.f0 = u0; e.f1 = u1; e.f2 = u2; e.f3 = 7.f + u0;
  e.pad[0] = e.pad[1] = e.pad[2] = e.pad[3] = 0.f;
  return e;
}

#define INIT_STATE                                                                                          \
  const unsigned lane = threadIdx.x & 63u;                                                                  \
  Poly pq[4];                                                                                               \
  Px st[4];                                                                                                 \
  unsigned long long alive[4];                                                                              \
  _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                           \
    const float x = (float)(lane & 7) - 7.5f + 8.f * (k & 1), y = (float)(lane >> 3) - 7.5f + 8.f * (k >> 1); \
    pq[k] = Poly{x, y, x * x, x * y, y * y};                                                                \
    st[k] = Px{1.f, 0.f, 0.f, 0.f, 0.f};                                                                    \
    alive[k] = ~0ull;                                                                                       \
  }

#define FINISH                                                                                              \
  float s = 0.f;                                                                                            \
  _Pragma("unroll") for (int k = 0; k < 4; ++k) s += st[k].T + st[k].c0 + st[k].c1 + st[k].c2 + st[k].c3;  \
  out[(size_t)blockIdx.x * 64 + lane] = s;

// every `reset` batches the pixels are reopened (a real tile saturates after ~5 batches; the bench keeps the work per
// body constant instead: all four quadrants stay live)
__global__ __launch_bounds__(64, 8) void feed_lds(float* out, unsigned long long* times, int n_batches, int count, unsigned seed) {
  extern __shared__ unsigned char pad_[];
  __shared__ Entry queue[65];
  INIT_STATE
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int b = 0; b < n_batches; ++b) {
    unsigned live = 0xfu;
    // each lane queues its own entry (the real kernel: after the cull, ballot-compacted)
    if ((int)lane < count) queue[lane] = make_entry((unsigned)b + blockIdx.x * 977u, lane, seed);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int j = 0; j < count; ++j) {
      const f4* e = reinterpret_cast<const f4*>(&queue[j]);
      f4 g0 = e[0], g1 = e[1], g2 = e[2];
      asm volatile("" : "+v"(g0), "+v"(g1), "+v"(g2));
      const unsigned m = __builtin_amdgcn_readfirstlane(__float_as_uint(g1.z)) & live;
      if (m & 1u) body_v(st[0], alive[0], live, pq[0], g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g2.x, g2.y, g2.z, g2.w, 0);
      if (m & 2u) body_v(st[1], alive[1], live, pq[1], g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g2.x, g2.y, g2.z, g2.w, 1);
      if (m & 4u) body_v(st[2], alive[2], live, pq[2], g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g2.x, g2.y, g2.z, g2.w, 2);
      if (m & 8u) body_v(st[3], alive[3], live, pq[3], g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g2.x, g2.y, g2.z, g2.w, 3);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) { alive[k] = ~0ull; st[k].T = st[k].T < 0.05f ? 1.f : st[k].T; }   // keep the pixels open
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) { times[2 * blockIdx.x] = t1 - t0; times[2 * blockIdx.x + 1] = r1 - r0; }
  FINISH
}

// scalar loads of one 48-byte entry: x8 + x4 into SGPR tuples (inline asm: the compiler will not select s_load for
// memory the kernel itself wrote).  The wait is explicit -- SMEM returns out of order, so only lgkmcnt(0) is safe.
__device__ __forceinline__ void s_load_entry(const Entry* p, f8& a, f4& b) {
  // (early clobber: the first load's result must not land on the address pair the second load still has to read)
  asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x20" : "=&s"(a), "=&s"(b) : "s"(p) : "memory");
}
// (the tuples are operands of the wait, so that no use of them can be scheduled ahead of it)
__device__ __forceinline__ void s_wait(f8& a, f4& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b) : : "memory"); }

template <bool INV>
__global__ __launch_bounds__(64, 8) void feed_scalar(float* out, unsigned long long* times, int n_batches, int count, unsigned seed,
                                                  Entry* rings) {
  extern __shared__ unsigned char pad_[];
  INIT_STATE
  Entry* ring = rings + (size_t)blockIdx.x * 64;           // this wave's ring: 4 KB, rewritten every batch
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int b = 0; b < n_batches; ++b) {
    unsigned live = 0xfu;
    if ((int)lane < count) {
      const Entry e = make_entry((unsigned)b + blockIdx.x * 977u, lane, seed);
      f4* d = reinterpret_cast<f4*>(&ring[lane]);
      d[0] = f4{e.q0, e.q1, e.q2, e.A};
      d[1] = f4{e.B, e.C, __uint_as_float(e.mask), __int_as_float(e.idx)};
      d[2] = f4{e.f0, e.f1, e.f2, e.f3};
    }
    // the stores must have reached L2 before the scalar loads (TCP is write-through); the scalar cache may hold the
    // previous batch's lines of the same addresses
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (INV) asm volatile("s_dcache_inv" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    f8 a0, a1; f4 b0, b1;
    s_load_entry(ring, a0, b0);
    for (int j = 0; j < count; j += 2) {
      s_wait(a0, b0);
      s_load_entry(ring + (j + 1 < count ? j + 1 : j), a1, b1);            // entry j + 1 on its way while j is blended
      {
        const unsigned m = __float_as_uint(a0[6]) & live;
        if (m & 1u) body_s(st[0], alive[0], live, pq[0], a0[0], a0[1], a0[2], a0[3], a0[4], a0[5], b0[0], b0[1], b0[2], b0[3], 0);
        if (m & 2u) body_s(st[1], alive[1], live, pq[1], a0[0], a0[1], a0[2], a0[3], a0[4], a0[5], b0[0], b0[1], b0[2], b0[3], 1);
        if (m & 4u) body_s(st[2], alive[2], live, pq[2], a0[0], a0[1], a0[2], a0[3], a0[4], a0[5], b0[0], b0[1], b0[2], b0[3], 2);
        if (m & 8u) body_s(st[3], alive[3], live, pq[3], a0[0], a0[1], a0[2], a0[3], a0[4], a0[5], b0[0], b0[1], b0[2], b0[3], 3);
      }
      if (j + 1 >= count) break;
      s_wait(a1, b1);
      s_load_entry(ring + (j + 2 < count ? j + 2 : j + 1), a0, b0);
      {
        const unsigned m = __float_as_uint(a1[6]) & live;
        if (m & 1u) body_s(st[0], alive[0], live, pq[0], a1[0], a1[1], a1[2], a1[3], a1[4], a1[5], b1[0], b1[1], b1[2], b1[3], 0);
        if (m & 2u) body_s(st[1], alive[1], live, pq[1], a1[0], a1[1], a1[2], a1[3], a1[4], a1[5], b1[0], b1[1], b1[2], b1[3], 1);
        if (m & 4u) body_s(st[2], alive[2], live, pq[2], a1[0], a1[1], a1[2], a1[3], a1[4], a1[5], b1[0], b1[1], b1[2], b1[3], 2);
        if (m & 8u) body_s(st[3], alive[3], live, pq[3], a1[0], a1[1], a1[2], a1[3], a1[4], a1[5], b1[0], b1[1], b1[2], b1[3], 3);
      }
    }
    s_wait(a0, b0);
#pragma unroll
    for (int k = 0; k < 4; ++k) { alive[k] = ~0ull; st[k].T = st[k].T < 0.05f ? 1.f : st[k].T; }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) { times[2 * blockIdx.x] = t1 - t0; times[2 * blockIdx.x + 1] = r1 - r0; }
  FINISH
}

// the same with entries j + 1 AND j + 2 on their way while entry j is blended (three SGPR sets in turn: 36 SGPRs)
#define BODY4(A_, B_)                                                                                                                    \
  {                                                                                                                                      \
    const unsigned m = __float_as_uint(A_[6]) & live;                                                                                    \
    if (m & 1u) body_s(st[0], alive[0], live, pq[0], A_[0], A_[1], A_[2], A_[3], A_[4], A_[5], B_[0], B_[1], B_[2], B_[3], 0);          \
    if (m & 2u) body_s(st[1], alive[1], live, pq[1], A_[0], A_[1], A_[2], A_[3], A_[4], A_[5], B_[0], B_[1], B_[2], B_[3], 1);          \
    if (m & 4u) body_s(st[2], alive[2], live, pq[2], A_[0], A_[1], A_[2], A_[3], A_[4], A_[5], B_[0], B_[1], B_[2], B_[3], 2);          \
    if (m & 8u) body_s(st[3], alive[3], live, pq[3], A_[0], A_[1], A_[2], A_[3], A_[4], A_[5], B_[0], B_[1], B_[2], B_[3], 3);          \
  }
__global__ __launch_bounds__(64, 8) void feed_scalar2(float* out, unsigned long long* times, int n_batches, int count, unsigned seed,
                                                      Entry* rings) {
  extern __shared__ unsigned char pad_[];
  INIT_STATE
  Entry* ring = rings + (size_t)blockIdx.x * 64;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int b = 0; b < n_batches; ++b) {
    unsigned live = 0xfu;
    if ((int)lane < count) {
      const Entry e = make_entry((unsigned)b + blockIdx.x * 977u, lane, seed);
      f4* d = reinterpret_cast<f4*>(&ring[lane]);
      d[0] = f4{e.q0, e.q1, e.q2, e.A};
      d[1] = f4{e.B, e.C, __uint_as_float(e.mask), __int_as_float(e.idx)};
      d[2] = f4{e.f0, e.f1, e.f2, e.f3};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_dcache_inv" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    f8 a0, a1, a2; f4 b0, b1, b2;
    const int last = count - 1;
    s_load_entry(ring, a0, b0);
    s_load_entry(ring + (1 < last ? 1 : last), a1, b1);
    // SMEM returns out of order: every wait is lgkmcnt(0), i.e. for BOTH entries in flight -- the second has had one
    // whole entry's blend to arrive, the first two
    for (int j = 0; j < count; j += 3) {
      s_wait(a0, b0);
      s_load_entry(ring + (j + 2 < last ? j + 2 : last), a2, b2);
      BODY4(a0, b0)
      if (j + 1 >= count) break;
      s_wait(a1, b1);
      s_load_entry(ring + (j + 3 < last ? j + 3 : last), a0, b0);
      BODY4(a1, b1)
      if (j + 2 >= count) break;
      s_wait(a2, b2);
      s_load_entry(ring + (j + 4 < last ? j + 4 : last), a1, b1);
      BODY4(a2, b2)
    }
    s_wait(a0, b0);
    s_wait(a1, b1);
#pragma unroll
    for (int k = 0; k < 4; ++k) { alive[k] = ~0ull; st[k].T = st[k].T < 0.05f ? 1.f : st[k].T; }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) { times[2 * blockIdx.x] = t1 - t0; times[2 * blockIdx.x + 1] = r1 - r0; }
  FINISH
}

int main(int argc, char** argv) {
  const int n_batches = argc > 1 ? atoi(argv[1]) : 48, count = argc > 2 ? atoi(argv[2]) : 54;
  const int only_form = argc > 3 ? atoi(argv[3]) : -1;
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs; %d batches of %d queued entries per wave, 2.25 quadrant bodies per entry on average\n", prop.gcnArchName,
         cus, n_batches, count);
  // bodies per wave (host replica of the mask table: every pattern equally likely, all quadrants live)
  const double bodies_per_entry = (1 + 1 + 1 + 1 + 2 + 2 + 2 + 2 + 2 + 2 + 3 + 3 + 3 + 3 + 4 + 4) / 16.0;
  const double bodies = (double)n_batches * count * bodies_per_entry;
  float* out; unsigned long long* times; Entry* rings;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute((const void*)feed_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192));
  CHECK(hipFuncSetAttribute((const void*)feed_scalar<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192));
  CHECK(hipFuncSetAttribute((const void*)feed_scalar<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192));
  CHECK(hipFuncSetAttribute((const void*)feed_scalar2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192));
  // last column: launch time x SIMDs / bodies of the launch = ns of one SIMD per quadrant body, tails and all
  printf("%-34s %5s %9s %12s %14s %12s %10s %12s\n", "form", "w/SIMD", "waves", "launch us", "ns/body/wave", "cyc/VALU", "clock GHz", "SIMD-ns/body");
  for (int wps : {4, 5, 6, 8}) {
    const int per_cu = 4 * wps;                                   // one-wave workgroups per CU
    // dynamic LDS so that exactly per_cu workgroups fit in 160 KB (static: 65 x 64 B = 4160 B in the LDS form)
    const int rounds = 3;                                         // the launch holds `rounds` x the resident waves
    const int waves = cus * per_cu * rounds;
    CHECK(hipMalloc(&out, (size_t)waves * 64 * 4));
    CHECK(hipMalloc(&times, (size_t)waves * 16));
    CHECK(hipMalloc(&rings, (size_t)waves * 64 * sizeof(Entry)));
    for (int form = 0; form < 4; ++form) {
      if (only_form >= 0 && form != only_form) continue;
      // dynamic LDS such that the occupancy API reports exactly per_cu workgroups per CU (the LDS allocation granule of
      // gfx950 is not assumed): the largest footprint that still lets per_cu fit
      const void* fn = form == 0 ? (const void*)feed_lds : form == 1 ? (const void*)feed_scalar<true>
                       : form == 2 ? (const void*)feed_scalar<false> : (const void*)feed_scalar2;
      int dyn = 160 * 1024 / per_cu / 256 * 256, fit = 0;
      for (; dyn >= 0; dyn -= 256) {
        CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, fn, 64, dyn));
        if (fit >= per_cu) break;
      }
      if (dyn < 0) dyn = 0;
      const double wps_real = fit / 4.0;
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        if (form == 0) hipLaunchKernelGGL(feed_lds, dim3(waves), dim3(64), dyn, 0, out, times, n_batches, count, 7u + rep);
        else if (form == 1) hipLaunchKernelGGL(feed_scalar<true>, dim3(waves), dim3(64), dyn, 0, out, times, n_batches, count, 7u + rep, rings);
        else if (form == 2) hipLaunchKernelGGL(feed_scalar<false>, dim3(waves), dim3(64), dyn, 0, out, times, n_batches, count, 7u + rep, rings);
        else hipLaunchKernelGGL(feed_scalar2, dim3(waves), dim3(64), dyn, 0, out, times, n_batches, count, 7u + rep, rings);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      std::vector<unsigned long long> t((size_t)waves * 2);
      CHECK(hipMemcpy(t.data(), times, (size_t)waves * 16, hipMemcpyDeviceToHost));
      double sc = 0, rc = 0;
      for (int w = 0; w < waves; ++w) { sc += (double)t[2 * w]; rc += (double)t[2 * w + 1]; }
      const double ghz = sc / rc * 0.1;                           // shader cycles per 10 ns tick
      const double wave_ns = rc / waves * 10.0;
      const int valu_per_body = form == 0 ? 15 : 16;
      // SIMD-cycles per vector instruction: the wave's shader cycles / (its instructions x the waves sharing the SIMD)
      const double cyc = (sc / waves) / (bodies * valu_per_body * wps_real);
      const char* names[4] = {"L: LDS queue, 3 x ds_read_b128", "S: ring + s_dcache_inv + s_load", "S0: ring + s_load, no invalidate",
                              "S2: as S, two entries ahead"};
      printf("%-34s %5.2f %9d %12.1f %14.2f %12.2f %10.2f %12.1f\n", names[form], wps_real, waves, best * 1e3, wave_ns / bodies, cyc, ghz,
             best * 1e6 / (waves * bodies) * cus * 4);
    }
    CHECK(hipFree(out)); CHECK(hipFree(times)); CHECK(hipFree(rings));
  }
  return 0;
}
