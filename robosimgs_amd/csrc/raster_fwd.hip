// raster_fwd.hip -- per-tile depth-ordered alpha compositing, forward (A.2 step 9), gfx950.
// Geometry, queue and culling: raster_common.h.  Not bandwidth-bound: 15 vector instructions per 64 pixel-Gaussian
// pairs (five FMAs for the exponent's polynomial about the tile centre, v_exp, the alpha test and the T (1 - alpha) > 1e-4
// test as v_cmpx under an EXEC that starts as the quadrant's open pixels, one move, the colour FMAs: 12 FMA-class + 2
// compare-class + one v_exp = 45 cycles by scripts/ubench/valu_issue.hip) against 44 bytes per tile-Gaussian pair, so
// the design spends its effort on evaluating
// fewer pairs, on cheaper evaluations and on keeping enough waves resident, not on moving bytes.  Two schedules of the
// same blend: raster_fwd_kernel (one wave per 16x16 tile, four pixels per lane: fewest instructions, what several
// frames in flight run) and raster_fwd_q_kernel (one wave per 8x8 block: shortest launch); DESIGN.md 4.3.
#include <type_traits>

#include "raster_common.h"
#include "tile_order.h"
#include "dataset_pixel.h"

#ifndef MGS_RASTER_WAVES
// min waves per SIMD asked of the register allocator (one-wave-per-tile kernel).  5 = at most 96 VGPRs: with the
// hand-written blend body the 4-channel kernels sat at 106-107 registers (4 waves) -- at 5 the same box renders
// 3,818 instead of 3,674 frames/s with three frames in flight (189.5 vs 191.6 us alone); 6 (80 VGPRs) spills: 3,571
#define MGS_RASTER_WAVES 5
#endif
#ifndef MGS_RASTER_PIPE
// 1: the one-wave-per-tile kernel reads queue entry j + 1 from LDS while it blends entry j (two register sets in turn).
// PMC has the waves parked on s_waitcnt for 39 % of their cycles, but both forms of the prefetch lose: 249 us (copies
// between the sets) / 214 us (ping-pong) against 188 us -- 30 more dwords spilled around the cull at the 96-register
// bound, and the read-ahead does not shorten the entry's own dependency chain.  Off.
#define MGS_RASTER_PIPE 0
#endif
#ifndef MGS_RASTER_NO_PREFETCH
#define MGS_RASTER_NO_PREFETCH 1     // 1: the one-wave-per-tile kernel fetches a batch when it needs it instead of one batch ahead (ten
                                     // registers live across the walk less): 188.2 -> 186.2 us alone, 3,819 -> 3,866 frames/s; 0 = prefetch
#endif
#ifndef MGS_RASTER_WG_WAVES
// Independent tiles (waves) per workgroup of the INFERENCE variant; no workgroup barrier is ever
// used.  One-wave workgroups grab every wave slot the moment it frees up and starve the 4-wave
// workgroups of the other frames' binning kernels, which need four slots on one CU at once; with
// 4 tiles per workgroup the frames in flight interleave better: 2873 -> 2956 frames/s (8 / 16
// tiles: 2858 / 2832, load imbalance).  A lone launch is 3 % slower that way (the workgroup lives
// as long as its heaviest tile), so the training variant, which runs alone, keeps one tile.
#define MGS_RASTER_WG_WAVES 4
#endif

namespace mgs {
namespace {

template <int CHT>
struct QueueEntry {
  float4 geo0;                       // q0, q1, q2, A   (raster_common.h: poly_coefs; A,B,C: conic pre-scaled)
  float4 geo1;                       // B, C, quadrant mask (bits), list index (bits)
  float4 feat[(CHT + 3) / 4];
  float4 geo3;                       // mean - tile centre (x, y): read only by batches that test sigma >= 0
};

// Per-pixel state.  Two ways to remember that a pixel is finished (A.2 step 9's "done" flag):
//   * sign of T (kernels without the hand-written body): T > 0 open, T < 0 finished with |T| its final transmittance --
//     no separate flag, and a finished pixel can never accumulate again: T*(1-alpha) < 0 < 1e-4;
//   * `alive` (kMasks kernels: 3 and 4 channels): one 64-bit lane mask per quadrant in SGPRs, T stays positive.  The
//     blend runs with EXEC = alive & (alpha >= 1/255) & (T (1 - alpha) > 1e-4), so "finished" costs no vector
//     instruction at all -- the sign form pays a compare-class select on w and one on T per 64 pairs.
template <int CHT>
struct PixelState {
  float T;
  float C[CHT];
  int last;
#ifdef MGS_RASTER_STATS
  unsigned n_valid = 0, n_acc = 0;
#endif
};

constexpr float kLog2e = 1.4426950408889634f;

#ifdef MGS_RASTER_STATS
// Instrumented build only (python robosimgs_amd/csrc/build.py with MGS_EXTRA_FLAGS=-DMGS_RASTER_STATS):
// counts the work the forward raster really does; read back with mgs_debug_read_raster_stats().
//   0 list entries fetched   1 entries queued after the cull   2 quadrant evaluations (x64 lanes)
//   3 lanes with a valid alpha   4 lanes accumulated   5 batches processed   6 batches in lists
__device__ unsigned long long g_raster_stats[8];
#define MGS_STAT(i, v) stat[i] += (v)
#else
#define MGS_STAT(i, v)
#endif

// One Gaussian against the 64 pixels of one quadrant (one pixel per lane).
//   alpha = opacity exp(-sigma) = exp2(A dx^2 + C dy^2 + B dx dy + L)  with  A = -0.5 log2e a, B = -log2e b,
//   C = -0.5 log2e c, L = log2(opacity): one v_exp_f32, no multiply by the opacity (raster_common.h pair_power);
//   "sigma >= 0" is "pair_power_sign <= 0".
// TRACK_LAST: record the list index of the last blended Gaussian (the backward starts there);
// an inference render drops that select (compares / selects issue at half the FMA rate on gfx950).
template <int CHT, bool TRACK_LAST, bool SAFE = false, bool MASKS = false>
__device__ __forceinline__ void blend_pixel(PixelState<CHT>& px, unsigned long long& alive, const PixelPoly& pp, float q0,
                                            float q1, float q2, float A, float B, float C, float m_x, float m_y,
                                            const float* feat, int idx) {
  // SAFE: the conic cannot round sigma below zero (sigma_sign_is_safe) and opacity <= kSafeOpacity: the sigma test
  // is dead and exp2(power) <= opacity (1 + 2^-22) < 0.999, so the clamp is the identity too
  const float ov = __builtin_amdgcn_exp2f(pair_power_poly(pp, q0, q1, q2, A, B, C));   // the backward repeats it bit for bit
  float alpha = SAFE ? ov : fminf(kAlphaMax, ov);
  bool valid = alpha >= kAlphaMin;
  if (!SAFE) valid = valid && pair_power_sign(m_x - pp.x, m_y - pp.y, A, B, C) <= 0.f;
  if (MASKS) valid = valid && __builtin_amdgcn_inverse_ballot_w64(alive);     // the mask IS the condition register
  // alpha forced to 0 where the Gaussian does not count: an open pixel (T > 1e-4 by invariant)
  // then keeps T and adds nothing, with no second mask to combine
  float a_eff = valid ? alpha : 0.f;
  float next_T = fmaf(-a_eff, px.T, px.T);
  bool acc = next_T > kTStop;                     // false for the closing Gaussian (and, in the sign form, for finished pixels)
  float w = a_eff * px.T;
  w = acc ? w : 0.f;      // (w = |T| - |T_new| -- a subtraction instead of a multiply and a select -- puts the weight
                          //  behind the T select on the dependency chain: measured 218 -> 226 us, 3,575 -> 3,417 frames/s)
#pragma unroll
  for (int c = 0; c < CHT; ++c) px.C[c] = fmaf(w, feat[c], px.C[c]);
  if (MASKS) {
    px.T = acc ? next_T : px.T;                   // a pixel this Gaussian closes keeps the T it had and leaves the mask
    alive &= ~ballot(!acc);
  } else {
    px.T = acc ? next_T : -fabsf(px.T);           // not accumulated: the pixel is (or stays) finished
  }
  if (TRACK_LAST) px.last = (acc && valid) ? idx : px.last;
#ifdef MGS_RASTER_STATS
  px.n_valid += valid && px.T > 0.f;
  px.n_acc += acc && valid;
#endif
}

#ifndef MGS_RASTER_CMPX
#define MGS_RASTER_CMPX 1
#endif
#ifndef MGS_RASTER_MASKS
#define MGS_RASTER_MASKS 1          // 0: the sign of T is the "finished" flag in every kernel (measurement)
#endif
#ifndef MGS_RASTER_Q_BREAK
#define MGS_RASTER_Q_BREAK 0        // one wave per 8x8 block: leave the batch at the entry that closes the block's last pixel
#endif
#ifndef MGS_RASTER_CLOSE_BRANCH
// 1: the lane-mask bookkeeping of a pixel that closes (s_andn2 / s_cselect / s_and) sits behind a scalar branch on "some
// pixel closed" -- nearly always skipped: 4 instead of 6 scalar instructions per quadrant body
#define MGS_RASTER_CLOSE_BRANCH 0
#endif
#ifndef MGS_RASTER_LIVE_BITS
#define MGS_RASTER_LIVE_BITS 1      // one wave per tile: a quadrant whose last pixel closes is skipped for the rest of the batch
#endif
// Measurement only (MGS_RASTER_MASKS=0, profiles/r3/00_experiments.md): the SAFE body with the sign of T as the
// "finished" flag -- 16 vector instructions, two of them selects, no scalar bookkeeping.
template <int CHT, bool TRACK_LAST>
__device__ __forceinline__ void blend_pixel_safe_asm_sign(PixelState<CHT>& px, const PixelPoly& pp, float q0, float q1,
                                                     float q2, float A, float B, float C, const float* feat, int idx) {
  static_assert(CHT == 3 || CHT == 4, "hand-written blend: 3 or 4 channels");
  float dx, t0, t1;               // dx: the weight w, t0: T (1 - alpha), t1: exponent, then alpha
  const float amin = kAlphaMin, tstop = kTStop;
  float c3 = CHT == 4 ? px.C[CHT - 1] : 0.f;
  const float f3 = CHT == 4 ? feat[CHT - 1] : 0.f;
  asm volatile(
      "v_fma_f32 %[t1], %[q1], %[x], %[q0]\n"          // pair_power_poly, same order
      "v_fmac_f32 %[t1], %[q2], %[y]\n"
      "v_fmac_f32 %[t1], %[A], %[xx]\n"
      "v_fmac_f32 %[t1], %[B], %[xy]\n"
      "v_fmac_f32 %[t1], %[C], %[yy]\n"
      "v_exp_f32 %[t1], %[t1]\n"
      "s_nop 0\n"
      "v_cmpx_le_f32 vcc, %[amin], %[t1]\n"
      "v_fma_f32 %[t0], -%[t1], %[T], %[T]\n"
      "v_mul_f32 %[dx], %[t1], %[T]\n"
      "v_cmp_lt_f32 vcc, %[tstop], %[t0]\n"
      "s_nop 1\n"
      "v_cndmask_b32 %[dx], 0, %[dx], vcc\n"
      "v_cndmask_b32_e64 %[T], -|%[T]|, %[t0], vcc\n"
      "v_fmac_f32 %[c0], %[dx], %[f0]\n"
      "v_fmac_f32 %[c1], %[dx], %[f1]\n"
      "v_fmac_f32 %[c2], %[dx], %[f2]\n"
      ".if %[four]\n"
      "v_fmac_f32 %[c3], %[dx], %[f3]\n"
      ".endif\n"
      ".if %[track]\n"
      "v_cndmask_b32 %[last], %[last], %[idx], vcc\n"     // EXEC = valid lanes, VCC = accumulated
      ".endif\n"
      "s_mov_b64 exec, -1\n"
      : [dx] "=&v"(dx), [t0] "=&v"(t0), [t1] "=&v"(t1),
        [T] "+v"(px.T), [c0] "+v"(px.C[0]), [c1] "+v"(px.C[1]), [c2] "+v"(px.C[2]), [c3] "+v"(c3), [last] "+v"(px.last)
      : [q0] "v"(q0), [q1] "v"(q1), [q2] "v"(q2), [x] "v"(pp.x), [y] "v"(pp.y), [xx] "v"(pp.xx), [xy] "v"(pp.xy),
        [yy] "v"(pp.yy), [A] "v"(A), [B] "v"(B), [C] "v"(C),
        [f0] "v"(feat[0]), [f1] "v"(feat[1]), [f2] "v"(feat[2]), [f3] "v"(f3), [amin] "s"(amin), [tstop] "s"(tstop),
        [four] "n"(CHT == 4 ? 1 : 0), [track] "n"(TRACK_LAST ? 1 : 0), [idx] "v"(idx)
      : "vcc");
  if (CHT == 4) px.C[CHT - 1] = c3;
}

// Kernels of 3 and 4 channels keep "finished" in lane masks (PixelState above) and run the SAFE blend as the
// hand-written body below.
template <int CHT>
constexpr bool kMasks = MGS_RASTER_MASKS != 0 && MGS_RASTER_CMPX != 0 && (CHT == 3 || CHT == 4);

// The SAFE blend (3 or 4 channels; inference and, with one more move for last_ids, training) as hand-written gfx950
// code: the same arithmetic in the same order as blend_pixel<CHT, ., true> -- bit-identical pixels -- under an EXEC that
// is narrowed three times instead of selects: to the quadrant's open pixels (s_mov from `alive`), by the alpha >= 1/255
// test (v_cmpx; a Gaussian that does not count leaves the pixel untouched) and by the T (1 - alpha) > 1e-4 test
// (v_cmpx again: what is left accumulates; the lanes the second test dropped are the pixels this Gaussian closes and
// leave `alive`).  15 vector instructions per 64 pairs -- 12 FMA-class, two compares, one v_exp: 45 cycles by
// scripts/ubench/valu_issue.hip; the round-3 form with the sign of T as the flag had two selects more (16, 51 cycles).
// gfx940+ needs one wait state after a transcendental before its result is read.
// LIVE_BIT >= 0 (one wave per tile): bit LIVE_BIT of `live` is cleared the moment the quadrant's last pixel closes, so
// that the rest of the batch skips the quadrant (the caller ands every entry's quadrant mask with `live`: scalar
// instructions only; re-deriving the live quadrants with ballots every 8 / 16 / 32 entries was a loss in round 3).
template <int CHT, bool TRACK_LAST, int LIVE_BIT = -1>
__device__ __forceinline__ void blend_pixel_safe_asm(PixelState<CHT>& px, unsigned long long& alive, unsigned& live,
                                                     const PixelPoly& pp,
                                                     float q0, float q1, float q2, float A, float B, float C,
                                                     const float* feat, int idx) {
  static_assert(CHT == 3 || CHT == 4, "hand-written blend: 3 or 4 channels");
  float dx, t0, t1;               // dx: the weight w, t0: T (1 - alpha), t1: exponent, then alpha
  unsigned long long acc;         // lanes that accumulate
  const float amin = kAlphaMin, tstop = kTStop;
  float c3 = CHT == 4 ? px.C[CHT - 1] : 0.f;
  const float f3 = CHT == 4 ? feat[CHT - 1] : 0.f;
  asm volatile(
      "s_mov_b64 exec, %[alive]\n"
      "v_fma_f32 %[t1], %[q1], %[x], %[q0]\n"          // pair_power_poly, same order
      "v_fmac_f32 %[t1], %[q2], %[y]\n"
      "v_fmac_f32 %[t1], %[A], %[xx]\n"
      "v_fmac_f32 %[t1], %[B], %[xy]\n"
      "v_fmac_f32 %[t1], %[C], %[yy]\n"
      "v_exp_f32 %[t1], %[t1]\n"
      "s_nop 0\n"
      "v_cmpx_le_f32 vcc, %[amin], %[t1]\n"            // EXEC = VCC = open pixels the Gaussian counts for
      "v_fma_f32 %[t0], -%[t1], %[T], %[T]\n"
      "v_mul_f32 %[dx], %[t1], %[T]\n"
      "v_cmpx_lt_f32_e64 %[acc], %[tstop], %[t0]\n"    // EXEC = acc = those of them that accumulate
      "v_mov_b32 %[T], %[t0]\n"
      "v_fmac_f32 %[c0], %[dx], %[f0]\n"
      "v_fmac_f32 %[c1], %[dx], %[f1]\n"
      "v_fmac_f32 %[c2], %[dx], %[f2]\n"
      ".if %[four]\n"
      "v_fmac_f32 %[c3], %[dx], %[f3]\n"
      ".endif\n"
      ".if %[track]\n"
      "v_mov_b32 %[last], %[idx]\n"
      ".endif\n"
      "s_xor_b64 vcc, vcc, %[acc]\n"                   // counted but not accumulated: the pixels this Gaussian closes
      ".if %[closebranch]\n"                           // (SCC = some pixel closes: rare -- the bookkeeping sits behind a branch)
      "s_cbranch_scc0 1f\n"
      ".endif\n"
      "s_andn2_b64 %[alive], %[alive], vcc\n"          // (SCC = some pixel of the quadrant is still open)
      ".if %[livebit] >= 0\n"
      "s_cselect_b32 vcc_lo, -1, %[clr]\n"
      "s_and_b32 %[live], %[live], vcc_lo\n"
      ".endif\n"
      "1:\n"
      "s_mov_b64 exec, -1\n"
      : [dx] "=&v"(dx), [t0] "=&v"(t0), [t1] "=&v"(t1), [acc] "=&s"(acc), [alive] "+s"(alive), [live] "+s"(live),
        [T] "+v"(px.T), [c0] "+v"(px.C[0]), [c1] "+v"(px.C[1]), [c2] "+v"(px.C[2]), [c3] "+v"(c3), [last] "+v"(px.last)
      : [q0] "v"(q0), [q1] "v"(q1), [q2] "v"(q2), [x] "v"(pp.x), [y] "v"(pp.y), [xx] "v"(pp.xx), [xy] "v"(pp.xy),
        [yy] "v"(pp.yy), [A] "v"(A), [B] "v"(B), [C] "v"(C),
        [f0] "v"(feat[0]), [f1] "v"(feat[1]), [f2] "v"(feat[2]), [f3] "v"(f3), [amin] "s"(amin), [tstop] "s"(tstop),
        [four] "n"(CHT == 4 ? 1 : 0), [track] "n"(TRACK_LAST ? 1 : 0), [idx] "v"(idx), [livebit] "n"(LIVE_BIT),
        [clr] "n"(LIVE_BIT >= 0 ? ~(1 << LIVE_BIT) : -1), [closebranch] "n"(MGS_RASTER_CLOSE_BRANCH)
      : "vcc", "scc");            // (s_xor / s_andn2 write SCC: the loop counter's compare must not straddle the body)
  if (CHT == 4) px.C[CHT - 1] = c3;
}

// DATASET (4 channels, inference, "ED"): the epilogue writes the dataset frame the reference reads -- RGBA8 + ray distance,
// dataset_pixel.h -- instead of (or, when render / alphas are given, beside) the float frame: 6 - 12 bytes per pixel leave
// the kernel instead of 20, and no conversion pass reads them back.  A separate instantiation: the float-frame kernels'
// code is what it was.
struct NoDataset {};
template <int CHT, bool TRACK_LAST, bool DATASET = false>
__global__ __launch_bounds__(64 * (TRACK_LAST ? 1 : MGS_RASTER_WG_WAVES), (CHT <= 4 ? MGS_RASTER_WAVES : 1)) void raster_fwd_kernel(
    const float* __restrict__ means2d, const float* __restrict__ conics,
    const float* __restrict__ feats, const float* __restrict__ opacities,
    const float4* __restrict__ splats, const float* __restrict__ background, int channels,
    int width, int height, int tile_w, int n_tiles, const int32_t* __restrict__ tile_offsets,
    const int32_t* __restrict__ flatten_ids, float* __restrict__ render,
    float* __restrict__ alphas, int32_t* __restrict__ last_ids, int cull, int expected_last, int opts,
    const int32_t* __restrict__ group_order, float* __restrict__ ckpt, int ckpt_shift,
    std::conditional_t<DATASET, DatasetOut, NoDataset> ds) {
  constexpr int kWgWaves = TRACK_LAST ? 1 : MGS_RASTER_WG_WAVES;
  __shared__ QueueEntry<CHT> queues[kWgWaves][kQueue + 1];
  QueueEntry<CHT>* queue = queues[threadIdx.x >> 6];
  const int tile = tile_of_unit(blockIdx.x * kWgWaves + (int)(threadIdx.x >> 6), n_tiles, group_order);   // tile_order.h
  if (tile < 0) return;
  const unsigned lane = threadIdx.x & 63u;
  const int tx = tile % tile_w, ty = tile / tile_w;
  const float tile_x = (float)(tx * 16), tile_y = (float)(ty * 16);
  const int start = tile_offsets[tile], end = tile_offsets[tile + 1];
  if (opts & 1) {
    // Issue priority by list length.  A tile is one wave's serial job; while every SIMD is full each of
    // its five waves advances at a fifth of the SIMD's rate, so the longest lists -- started no earlier
    // than the others -- set the kernel's duration.  Waves with long lists get priority and finish at
    // their own issue limit; short ones fill the gaps.
    const int avg = (tile_offsets[n_tiles] - tile_offsets[0]) / n_tiles, len = end - start;   // this camera's own total
    if (len > 2 * avg) __builtin_amdgcn_s_setprio(3);
    else if (2 * len > 3 * avg) __builtin_amdgcn_s_setprio(2);
    else if (len > avg) __builtin_amdgcn_s_setprio(1);
  }

  // pixel centres of quadrant 0; quadrant k adds (8*(k&1), 8*(k>>1))
  const int ix = tx * 16 + (int)(lane & 7), iy = ty * 16 + (int)(lane >> 3);
  // offsets of this lane's four pixels from the tile centre, and their products (raster_common.h: PixelPoly)
  const float xo = (float)(lane & 7) - 7.5f, yo = (float)(lane >> 3) - 7.5f;
  PixelPoly pq[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) pq[k] = pixel_poly(xo + 8.f * (k & 1), yo + 8.f * (k >> 1));
  const float ctr_x = tile_x + 8.f, ctr_y = tile_y + 8.f;

#ifdef MGS_RASTER_STATS
  unsigned long long stat[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  stat[6] = (unsigned)(end - start + kQueue - 1) / kQueue;
#endif
  PixelState<CHT> st[4];
  unsigned long long alive[4];                   // kMasks: the quadrant's open pixels
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool inside = ix + 8 * (k & 1) < width && iy + 8 * (k >> 1) < height;
    st[k].T = (inside || kMasks<CHT>) ? 1.f : -1.f;    // pixels outside the image start finished
    alive[k] = ballot(inside);
    st[k].last = 0;
#pragma unroll
    for (int c = 0; c < CHT; ++c) st[k].C[c] = 0.f;
  }

  // raw batch registers (software prefetch of the next 64 list entries)
  int r_idx = start + (int)lane;
  bool r_ok = r_idx < end;
  float2 r_xy = make_float2(0.f, 0.f);
  float r_ca = 1.f, r_cb = 0.f, r_cc = 1.f, r_op = 0.f;
  float r_feat[CHT];
#pragma unroll
  for (int c = 0; c < CHT; ++c) r_feat[c] = 0.f;
  auto fetch = [&](int idx, bool ok) {
    if (ok && CHT <= 4 && splats) {   // one packed 48-byte record instead of four gathers
      int g = flatten_ids[idx];
      const float4 p0 = splats[3 * (size_t)g], p1 = splats[3 * (size_t)g + 1], p2 = splats[3 * (size_t)g + 2];
      r_xy = make_float2(p0.x, p0.y);
      r_ca = p0.z; r_cb = p0.w; r_cc = p1.x; r_op = p1.y;
      const float ff[4] = {p1.z, p1.w, p2.x, p2.y};
#pragma unroll
      for (int c = 0; c < CHT; ++c) r_feat[c] = ff[c & 3];
    } else if (ok) {
      int g = flatten_ids[idx];
      r_xy = reinterpret_cast<const float2*>(means2d)[g];
      r_ca = conics[3 * (size_t)g + 0];
      r_cb = conics[3 * (size_t)g + 1];
      r_cc = conics[3 * (size_t)g + 2];
      r_op = opacities[g];
#pragma unroll
      for (int c = 0; c < CHT; ++c) r_feat[c] = c < channels ? feats[(size_t)g * channels + c] : 0.f;
    }
  };
  fetch(r_idx, r_ok);

  for (int b = start; b < end; b += kQueue) {
    // which quadrants still have an unfinished pixel
    unsigned live = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if ((kMasks<CHT> ? alive[k] : ballot(st[k].T > 0.f)) != 0ull) live |= 1u << k;
    if (live == 0) break;
    if constexpr (TRACK_LAST) {
      // first batch of a segment (raster_common.h: checkpoints): the state in front of it, for the segmented backward
      if (ckpt && b != start && ((b - start) & ((1 << ckpt_shift) - 1)) == 0) {
        float* cp = ckpt + ckpt_header_floats(n_tiles) + ckpt_unit(start, tile, (b - start) >> ckpt_shift, ckpt_shift) * (size_t)(1 + channels) * 256 + lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          cp[64 * k] = fabsf(st[k].T);
#pragma unroll
          for (int c = 0; c < CHT; ++c)
            if (c < channels) cp[256 * (1 + c) + 64 * k] = st[k].C[c];
        }
      }
    }

    // take the prefetched batch, start the next one
#if MGS_RASTER_NO_PREFETCH
    if (b != start) { r_idx = b + (int)lane; r_ok = r_idx < end; fetch(r_idx, r_ok); }
#endif
    const int c_idx = r_idx;
    const bool c_ok = r_ok;
    const float2 c_xy = r_xy;
    const float c_ca = r_ca, c_cb = r_cb, c_cc = r_cc, c_op = r_op;
    float c_feat[CHT];
#pragma unroll
    for (int c = 0; c < CHT; ++c) c_feat[c] = r_feat[c];
#if !MGS_RASTER_NO_PREFETCH
    r_idx = b + kQueue + (int)lane;
    r_ok = r_idx < end;
    fetch(r_idx, r_ok);
#endif

    unsigned qmask = 0;
    if (c_ok) qmask = (cull ? quadrant_mask(c_xy.x, c_xy.y, c_ca, c_cb, c_cc, c_op, tile_x, tile_y) : 0xfu) & live;
    const unsigned long long keep = ballot(qmask != 0u);
    // every queued Gaussian of this batch has a well conditioned conic and an opacity <= 0.999 (nearly
    // always): the sigma >= 0 test and the 0.999 clamp are dead for the whole batch and the walk below runs
    // without them (raster_common.h: sigma_sign_is_safe) -- two compare / min class instructions less per 64 pairs
    const bool all_safe = ballot(qmask != 0u && !entry_is_safe(c_ca, c_cb, c_cc, c_op)) == 0ull;
    const int count = __popcll(keep);
    MGS_STAT(0, __popcll(ballot(c_ok)));
    MGS_STAT(1, count);
    MGS_STAT(5, 1);
    if (qmask != 0u) {
      QueueEntry<CHT>& e = queue[mask_rank(keep)];
      const float sA = -0.5f * kLog2e * c_ca, sB = -kLog2e * c_cb, sC = -0.5f * kLog2e * c_cc;
      const float m_x = c_xy.x - ctr_x, m_y = c_xy.y - ctr_y;
      const PolyCoef q = poly_coefs(m_x, m_y, sA, sB, sC, __log2f(c_op));
      e.geo0 = make_float4(q.q0, q.q1, q.q2, sA);
      e.geo1 = make_float4(sB, sC, __uint_as_float(qmask), __int_as_float(c_idx));
      e.geo3 = make_float4(m_x, m_y, 0.f, 0.f);
#pragma unroll
      for (int f = 0; f < (CHT + 3) / 4; ++f) {
        float4 v;
        v.x = c_feat[4 * f];
        v.y = 4 * f + 1 < CHT ? c_feat[4 * f + 1] : 0.f;
        v.z = 4 * f + 2 < CHT ? c_feat[4 * f + 2] : 0.f;
        v.w = 4 * f + 3 < CHT ? c_feat[4 * f + 3] : 0.f;
        e.feat[f] = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // (measured and rejected: reading entry j+1 while blending entry j, 307 vs 292 us; two
    //  entries per loop trip, 291 vs 288 us)
    auto blend_entry = [&](auto safe_tag, const float4& g0, const float4& g1, const float4* ef, const float4& g3) {
      constexpr bool SAFE = decltype(safe_tag)::value;
      float feat[CHT];
#pragma unroll
      for (int f = 0; f < (CHT + 3) / 4; ++f) {
        feat[4 * f] = ef[f].x;
        if (4 * f + 1 < CHT) feat[4 * f + 1] = ef[f].y;
        if (4 * f + 2 < CHT) feat[4 * f + 2] = ef[f].z;
        if (4 * f + 3 < CHT) feat[4 * f + 3] = ef[f].w;
      }
      // (kMasks: `live` loses a quadrant's bit the moment its last pixel closes, blend_pixel_safe_asm)
      const unsigned m = __builtin_amdgcn_readfirstlane(__float_as_uint(g1.z)) & (kMasks<CHT> ? live : 0xfu);
      const int idx = __float_as_int(g1.w);
      MGS_STAT(2, __popc(m));
      // (measured and rejected, profiles/r3/00_experiments.md: one straight-line body per quadrant SET behind a
      //  switch on the mask, 268-278 us against 197; the live quadrants re-derived every 8 / 16 / 32 entries, +2 %)
      auto quad = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if (m & (1u << k)) {
          if constexpr (kMasks<CHT> && SAFE)
            blend_pixel_safe_asm<CHT, TRACK_LAST, MGS_RASTER_LIVE_BITS ? k : -1>(st[k], alive[k], live, pq[k], g0.x, g0.y, g0.z,
                                                                                g0.w, g1.x, g1.y, feat, idx);
          else if constexpr (!MGS_RASTER_MASKS && MGS_RASTER_CMPX && SAFE && (CHT == 3 || CHT == 4))
            blend_pixel_safe_asm_sign<CHT, TRACK_LAST>(st[k], pq[k], g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, feat, idx);
          else
            blend_pixel<CHT, TRACK_LAST, SAFE, kMasks<CHT>>(st[k], alive[k], pq[k], g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g3.x,
                                                             g3.y, feat, idx);
        }
      };
      quad(std::integral_constant<int, 0>{});
      quad(std::integral_constant<int, 1>{});
      quad(std::integral_constant<int, 2>{});
      quad(std::integral_constant<int, 3>{});
    };
    auto walk = [&](auto safe_tag) {
      constexpr bool SAFE = decltype(safe_tag)::value;
#if MGS_RASTER_PIPE
      // entry j + 1 is read from LDS while entry j is blended (the queue has one spare slot: the read past the last
      // entry is harmless): PMC had the waves parked on s_waitcnt for 39 % of their cycles, most of it these reads
      if constexpr (CHT <= 4) {
        // two register sets in turn (no copies): while set A is blended set B is on its way, and vice versa
        float4 a0 = queue[0].geo0, a1 = queue[0].geo1, af = queue[0].feat[0], a3 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 b0, b1, bf, b3 = a3;
        if constexpr (!SAFE) a3 = queue[0].geo3;
        for (int j = 0; j < count; j += 2) {
          {
            const QueueEntry<CHT>& e = queue[j + 1];
            b0 = e.geo0; b1 = e.geo1; bf = e.feat[0];
            if constexpr (!SAFE) b3 = e.geo3;
          }
          blend_entry(safe_tag, a0, a1, &af, a3);
          if (j + 1 >= count) break;
          {
            const QueueEntry<CHT>& e = queue[min(j + 2, kQueue)];
            a0 = e.geo0; a1 = e.geo1; af = e.feat[0];
            if constexpr (!SAFE) a3 = e.geo3;
          }
          blend_entry(safe_tag, b0, b1, &bf, b3);
        }
        return;
      }
#endif
      for (int j = 0; j < count; ++j) {
        const QueueEntry<CHT>& e = queue[j];
        float4 g0, g1, ef[(CHT + 3) / 4];
        if constexpr (CHT <= 4) {
          lds_read_3f4(&e.geo0, &e.geo1, &e.feat[0], g0, g1, ef[0]);
        } else {
          g0 = e.geo0; g1 = e.geo1;
#pragma unroll
          for (int f = 0; f < (CHT + 3) / 4; ++f) ef[f] = e.feat[f];
        }
        float4 g3 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (!SAFE) g3 = e.geo3;            // the mean's offset: only the sigma >= 0 test reads it
        blend_entry(safe_tag, g0, g1, ef, g3);
      }
    };
    if (all_safe) walk(std::true_type{}); else walk(std::false_type{});
    __builtin_amdgcn_wave_barrier();   // queue is rewritten by the next batch
  }

#ifdef MGS_RASTER_STATS
  {
    unsigned nv = 0, na = 0;
    for (int k = 0; k < 4; ++k) { nv += st[k].n_valid; na += st[k].n_acc; }
    for (int d = 32; d >= 1; d >>= 1) { nv += __shfl_xor(nv, d); na += __shfl_xor(na, d); }
    stat[3] = nv; stat[4] = na;
    if (lane == 0)
      for (int i = 0; i < 7; ++i) atomicAdd(&g_raster_stats[i], stat[i]);
  }
#endif
  if constexpr (TRACK_LAST) {
    if (ckpt) {       // where the backward's walk of this tile ends (header of the checkpoint buffer)
      int h = max(max(st[0].last, st[1].last), max(st[2].last, st[3].last));
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) h = max(h, __shfl_xor(h, d));
      if (lane < 4) reinterpret_cast<int32_t*>(ckpt)[4 * tile + lane] = lane == 0 ? h : -1;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int x = ix + 8 * (k & 1), y = iy + 8 * (k >> 1);
    if (x < width && y < height) {
      const size_t p = (size_t)y * width + x;
      const float alpha = 1.0f - fabsf(st[k].T);
      // "ED": the last channel (depth sum) leaves as the expected depth, A.2 step 9
      const float inv_alpha = expected_last ? 1.0f / fmaxf(alpha, 1e-10f) : 1.0f;
      if constexpr (DATASET) {
        static_assert(CHT == 4 && !TRACK_LAST, "dataset epilogue: RGB + expected depth, inference");
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = st[k].C[c] + (background ? fabsf(st[k].T) * background[c] : 0.f);
        v[3] *= inv_alpha;
        dataset_store(ds, p, x, y, v[0], v[1], v[2], v[3], alpha);
        if (render) {
#pragma unroll
          for (int c = 0; c < 4; ++c) render[p * 4 + c] = v[c];
          alphas[p] = alpha;
        }
        continue;
      }
#pragma unroll
      for (int c = 0; c < CHT; ++c)
        if (c < channels) {
          float v = st[k].C[c] + (background ? fabsf(st[k].T) * background[c] : 0.f);
          if (c == channels - 1) v *= inv_alpha;
          render[p * channels + c] = v;
        }
      alphas[p] = alpha;
      if (TRACK_LAST) last_ids[p] = st[k].last;
    }
  }
}


// ---- one wave per 8x8 block ------------------------------------------------------------------
// Same blend, finer work units: a workgroup is one tile, wave k of it owns quadrant k with ONE pixel
// per lane.  Four times the waves, each a quarter of the serial work and half the registers (8 waves
// per SIMD instead of 5): the chip is filled in ~4 even rounds instead of 1.6 uneven ones, no
// quadrant branches in the inner loop, and a quadrant that saturates frees its wave slot at once.
// The price is that the list is fetched and culled by each of the four waves (L2 hits) and that the
// pixel offset (2 subtractions) is paid per evaluation instead of per queue entry.
template <int CHT, bool TRACK_LAST, bool DATASET = false>
__global__ __launch_bounds__(256) void raster_fwd_q_kernel(    // (bounded to 64 VGPRs / 8 waves per SIMD: 166-172 -> 182 us, left free: 66)
    const float* __restrict__ means2d, const float* __restrict__ conics,
    const float* __restrict__ feats, const float* __restrict__ opacities,
    const float4* __restrict__ splats, const float* __restrict__ background, int channels,
    int width, int height, int tile_w, int n_tiles, const int32_t* __restrict__ tile_offsets,
    const int32_t* __restrict__ flatten_ids, float* __restrict__ render,
    float* __restrict__ alphas, int32_t* __restrict__ last_ids, int cull, int expected_last,
    const int32_t* __restrict__ group_order, float* __restrict__ ckpt, int ckpt_shift,
    std::conditional_t<DATASET, DatasetOut, NoDataset> ds) {
  __shared__ QueueEntry<CHT> queues[4][kQueue];
#ifdef MGS_RASTER_Q_VGPR_CLOBBER
  // occupancy experiment: naming a high VGPR raises the kernel's register allocation (and lowers its
  // waves per SIMD) without touching LDS, which the other frames' kernels need
  asm volatile("" ::: MGS_RASTER_Q_VGPR_CLOBBER);
#endif
  const int k = (int)(threadIdx.x >> 6);
  QueueEntry<CHT>* queue = queues[k];
  const int tile = tile_of_unit(blockIdx.x, n_tiles, group_order);      // tile_order.h
  if (tile < 0) return;
  const unsigned lane = threadIdx.x & 63u;
  const int tx = tile % tile_w, ty = tile / tile_w;
  const int start = tile_offsets[tile], end = tile_offsets[tile + 1];
  const int ix = tx * 16 + 8 * (k & 1) + (int)(lane & 7), iy = ty * 16 + 8 * (k >> 1) + (int)(lane >> 3);
  // this lane's pixel as an offset from the TILE centre (the same coordinates as the one-wave-per-tile kernel: same bits)
  const PixelPoly pp = pixel_poly((float)(8 * (k & 1) + (int)(lane & 7)) - 7.5f, (float)(8 * (k >> 1) + (int)(lane >> 3)) - 7.5f);
  const bool inside = ix < width && iy < height;
  QuadRect rect;
  rect.x0 = (float)(tx * 16 + 8 * (k & 1)) + 0.5f; rect.x1 = rect.x0 + 7.f;
  rect.y0 = (float)(ty * 16 + 8 * (k >> 1)) + 0.5f; rect.y1 = rect.y0 + 7.f;
  const float tile_x = (float)(tx * 16), tile_y = (float)(ty * 16);

  PixelState<CHT> st;
  st.T = (inside || kMasks<CHT>) ? 1.f : -1.f;
  unsigned long long alive = ballot(inside);     // kMasks: the block's open pixels
  st.last = 0;
#pragma unroll
  for (int c = 0; c < CHT; ++c) st.C[c] = 0.f;

  int r_idx = start + (int)lane;
  bool r_ok = r_idx < end;
  float2 r_xy = make_float2(0.f, 0.f);
  float r_ca = 1.f, r_cb = 0.f, r_cc = 1.f, r_op = 0.f;
  float r_feat[CHT];
#pragma unroll
  for (int c = 0; c < CHT; ++c) r_feat[c] = 0.f;
  auto fetch = [&](int idx, bool ok) {
    if (ok && CHT <= 4 && splats) {
      int g = flatten_ids[idx];
      const float4 p0 = splats[3 * (size_t)g], p1 = splats[3 * (size_t)g + 1], p2 = splats[3 * (size_t)g + 2];
      r_xy = make_float2(p0.x, p0.y);
      r_ca = p0.z; r_cb = p0.w; r_cc = p1.x; r_op = p1.y;
      const float ff[4] = {p1.z, p1.w, p2.x, p2.y};
#pragma unroll
      for (int c = 0; c < CHT; ++c) r_feat[c] = ff[c & 3];
    } else if (ok) {
      int g = flatten_ids[idx];
      r_xy = reinterpret_cast<const float2*>(means2d)[g];
      r_ca = conics[3 * (size_t)g + 0];
      r_cb = conics[3 * (size_t)g + 1];
      r_cc = conics[3 * (size_t)g + 2];
      r_op = opacities[g];
#pragma unroll
      for (int c = 0; c < CHT; ++c) r_feat[c] = c < channels ? feats[(size_t)g * channels + c] : 0.f;
    }
  };
  fetch(r_idx, r_ok);

  for (int b = start; b < end; b += kQueue) {
    if ((kMasks<CHT> ? alive : ballot(st.T > 0.f)) == 0ull) break;          // every pixel of the block is finished
    if constexpr (TRACK_LAST) {
      // first batch of a segment (raster_common.h: checkpoints): this block's state in front of it.  A block that is
      // finished by then stores nothing: none of its pixels has a last_id in or past the segment, and the backward
      // takes a checkpoint only for pixels that do.
      if (ckpt && b != start && ((b - start) & ((1 << ckpt_shift) - 1)) == 0) {
        float* cp = ckpt + ckpt_header_floats(n_tiles) + ckpt_unit(start, tile, (b - start) >> ckpt_shift, ckpt_shift) * (size_t)(1 + channels) * 256 + 64 * k + lane;
        cp[0] = fabsf(st.T);
#pragma unroll
        for (int c = 0; c < CHT; ++c)
          if (c < channels) cp[256 * (1 + c)] = st.C[c];
      }
    }
    const int c_idx = r_idx;
    const bool c_ok = r_ok;
    const float2 c_xy = r_xy;
    const float c_ca = r_ca, c_cb = r_cb, c_cc = r_cc, c_op = r_op;
    float c_feat[CHT];
#pragma unroll
    for (int c = 0; c < CHT; ++c) c_feat[c] = r_feat[c];
    r_idx = b + kQueue + (int)lane;
    r_ok = r_idx < end;
    fetch(r_idx, r_ok);

    // same exact test as quadrant_mask (raster_common.h), for this block only
    bool keep_me = false;
    if (c_ok) {
      keep_me = true;
      if (cull) {
        keep_me = false;
        if (c_op >= kAlphaMin) {
          // (single v_log_f32 / v_rcp_f32, as in quadrant_mask: the slack is orders above their error)
#ifdef MGS_Q_CULL_IEEE      // measurement: the divides and the denormal-safe logf
          const float thr = __logf(255.0f * c_op);
#else
          const float thr = 0.6931471805599453f * __builtin_amdgcn_logf(255.0f * c_op);
#endif
          const float fx = fmaxf(fabsf(tile_x - c_xy.x), fabsf(tile_x + 16.f - c_xy.x));
          const float fy = fmaxf(fabsf(tile_y - c_xy.y), fabsf(tile_y + 16.f - c_xy.y));
          const float slack = 0.05f + 4e-6f * (fabsf(c_ca) + fabsf(c_cc) + 2.f * fabsf(c_cb)) * (fx * fx + fy * fy);
#ifdef MGS_Q_CULL_IEEE
          const float smin = rect_min_sigma(c_xy.x, c_xy.y, c_ca, c_cb, c_cc, 1.0f / c_ca, 1.0f / c_cc, rect);
#else
          const float smin = rect_min_sigma(c_xy.x, c_xy.y, c_ca, c_cb, c_cc, __builtin_amdgcn_rcpf(c_ca), __builtin_amdgcn_rcpf(c_cc), rect);
#endif
          keep_me = !(smin > thr + slack);
        }
      }
    }
    const unsigned long long keep = ballot(keep_me);
    const int count = __popcll(keep);
    const bool all_safe = ballot(keep_me && !entry_is_safe(c_ca, c_cb, c_cc, c_op)) == 0ull;
    if (keep_me) {
      QueueEntry<CHT>& e = queue[mask_rank(keep)];
      const float sA = -0.5f * kLog2e * c_ca, sB = -kLog2e * c_cb, sC = -0.5f * kLog2e * c_cc;
      const float m_x = c_xy.x - (tile_x + 8.f), m_y = c_xy.y - (tile_y + 8.f);
      const PolyCoef q = poly_coefs(m_x, m_y, sA, sB, sC, __log2f(c_op));
      e.geo0 = make_float4(q.q0, q.q1, q.q2, sA);
      e.geo1 = make_float4(sB, sC, 0.f, __int_as_float(c_idx));
      e.geo3 = make_float4(m_x, m_y, 0.f, 0.f);
#pragma unroll
      for (int f = 0; f < (CHT + 3) / 4; ++f) {
        float4 v;
        v.x = c_feat[4 * f];
        v.y = 4 * f + 1 < CHT ? c_feat[4 * f + 1] : 0.f;
        v.z = 4 * f + 2 < CHT ? c_feat[4 * f + 2] : 0.f;
        v.w = 4 * f + 3 < CHT ? c_feat[4 * f + 3] : 0.f;
        e.feat[f] = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    auto walk = [&](auto safe_tag) {
      constexpr bool SAFE = decltype(safe_tag)::value;
      // A block whose 64 pixels are all finished stops at the next multiple of 8 entries instead of the end
      // of the 64-entry batch -- in the training variant only: measured 206 -> 194 us there, 184 -> 191 us
      // for the inference variant (whose loop the compiler unrolls four times when left whole).
      constexpr int kChunk = TRACK_LAST ? 8 : kQueue;
      for (int j0 = 0; j0 < count; j0 += kChunk) {
        if (j0 && (kMasks<CHT> ? alive : ballot(st.T > 0.f)) == 0ull) break;
        const int j1 = min(j0 + kChunk, count);
      for (int j = j0; j < j1; ++j) {
#if MGS_RASTER_Q_BREAK
        if (kMasks<CHT> && alive == 0ull) break;     // (a scalar compare: the block's last pixel closed)
#endif
        const QueueEntry<CHT>& e = queue[j];
        float4 g0, g1, ef0;
        if constexpr (CHT <= 4) lds_read_3f4(&e.geo0, &e.geo1, &e.feat[0], g0, g1, ef0);
        else { g0 = e.geo0; g1 = e.geo1; ef0 = e.feat[0]; }
        float feat[CHT];
#pragma unroll
        for (int f = 0; f < (CHT + 3) / 4; ++f) {
          const float4 v = f == 0 ? ef0 : e.feat[f];
          feat[4 * f] = v.x;
          if (4 * f + 1 < CHT) feat[4 * f + 1] = v.y;
          if (4 * f + 2 < CHT) feat[4 * f + 2] = v.z;
          if (4 * f + 3 < CHT) feat[4 * f + 3] = v.w;
        }
        if constexpr (kMasks<CHT> && SAFE) {
          unsigned unused = 0;
          blend_pixel_safe_asm<CHT, TRACK_LAST>(st, alive, unused, pp, g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, feat, __float_as_int(g1.w));
        } else if constexpr (!MGS_RASTER_MASKS && MGS_RASTER_CMPX && SAFE && (CHT == 3 || CHT == 4)) {
          blend_pixel_safe_asm_sign<CHT, TRACK_LAST>(st, pp, g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, feat, __float_as_int(g1.w));
        } else {
          float4 g3 = make_float4(0.f, 0.f, 0.f, 0.f);
          if constexpr (!SAFE) g3 = e.geo3;
          blend_pixel<CHT, TRACK_LAST, SAFE, kMasks<CHT>>(st, alive, pp, g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g3.x, g3.y, feat,
                                                           __float_as_int(g1.w));
        }
      }
      }
    };
    if (all_safe) walk(std::true_type{}); else walk(std::false_type{});
    __builtin_amdgcn_wave_barrier();
  }

  if constexpr (TRACK_LAST) {
    if (ckpt) {       // where the backward's walk of this block ends (header of the checkpoint buffer)
      int h = inside ? st.last : -1;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) h = max(h, __shfl_xor(h, d));
      if (lane == 0) reinterpret_cast<int32_t*>(ckpt)[4 * tile + k] = h;
    }
  }
  if (inside) {
    const size_t p = (size_t)iy * width + ix;
    const float alpha = 1.0f - fabsf(st.T);
    const float inv_alpha = expected_last ? 1.0f / fmaxf(alpha, 1e-10f) : 1.0f;
    if constexpr (DATASET) {
      static_assert(CHT == 4 && !TRACK_LAST, "dataset epilogue: RGB + expected depth, inference");
      float v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = st.C[c] + (background ? fabsf(st.T) * background[c] : 0.f);
      v[3] *= inv_alpha;
      dataset_store(ds, p, ix, iy, v[0], v[1], v[2], v[3], alpha);
      if (render) {
#pragma unroll
        for (int c = 0; c < 4; ++c) render[p * 4 + c] = v[c];
        alphas[p] = alpha;
      }
      return;
    }
#pragma unroll
    for (int c = 0; c < CHT; ++c)
      if (c < channels) {
        float v = st.C[c] + (background ? fabsf(st.T) * background[c] : 0.f);
        if (c == channels - 1) v *= inv_alpha;
        render[p * channels + c] = v;
      }
    alphas[p] = alpha;
    if (TRACK_LAST) last_ids[p] = st.last;
  }
}

}  // namespace
}  // namespace mgs

using namespace mgs;

// Test hook: 0 disables the exact quadrant cull (every listed Gaussian is evaluated against every
// live quadrant).  The image must not change; tests/test_gpu_forward.py checks that bit for bit.
// Both knobs exist in libmgs_debug.so only (-DMGS_DEBUG_HOOKS; tests and A/B scripts load that build): the shipped
// libmgs.so has no process-global state -- two renderers in one process cannot race on a knob (include/mgs.h:
// "stateless and re-entrant").
#ifdef MGS_DEBUG_HOOKS
static int g_raster_cull = 1;
extern "C" void mgs_debug_set_raster_cull(int enabled) { g_raster_cull = enabled; }
#else
static constexpr int g_raster_cull = 1;
#endif
// Scheduling knobs for A/B measurements (scripts/raster_ab.py); they never change a pixel.
//   bit 0: issue priority by tile-list length in the one-wave-per-tile kernel (default on)
//   bit 1: honour MGS_RASTER_LATENCY (one wave per 8x8 block, raster_fwd_q_kernel, <= 4 channels); default on
//   bit 2: use that kernel whatever the flags say
//   bit 3: ignore tile_group_order (index-order launch); bit 4: ignore it in the one-wave-per-tile inference kernel only
//   bits 8..: KiB of unused dynamic LDS per workgroup of that kernel (caps its occupancy: experiments)
#ifdef MGS_DEBUG_HOOKS
static int g_raster_opts = 3;
extern "C" void mgs_debug_set_raster_opts(int opts) { g_raster_opts = opts; }
#else
static constexpr int g_raster_opts = 3;
#endif

#ifdef MGS_RASTER_STATS
// instrumented build only: copy the counters out (synchronous) and zero them
extern "C" int mgs_debug_read_raster_stats(unsigned long long* out8) {
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_raster_stats), sizeof(zero)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_raster_stats), zero, sizeof(zero)) != hipSuccess) return -1;
  return 0;
}
#endif

extern "C" size_t mgs_raster_checkpoint_floats(uint32_t isect_capacity, int tile_w, int tile_h, int channels,
                                               int checkpoint_interval) {
  if (checkpoint_interval < 64 || (checkpoint_interval & (checkpoint_interval - 1)) || tile_w <= 0 || tile_h <= 0 || channels < 1)
    return 0;
  int shift = 0;
  while ((1 << shift) < checkpoint_interval) ++shift;
  return ckpt_header_floats(tile_w * tile_h) + ckpt_units(isect_capacity, tile_w * tile_h, shift) * (size_t)(1 + channels) * 256;
}

extern "C" int mgs_rasterize_fwd(int n, const float* means2d, const float* conics,
                                 const float* feats, const float* opacities, const float* splats,
                                 const float* background, int channels, int width, int height,
                                 int tile_w, int tile_h, const int32_t* tile_offsets,
                                 const int32_t* flatten_ids, const int32_t* tile_group_order, int flags,
                                 float* render, float* alphas, int32_t* last_ids, float* checkpoints,
                                 int checkpoint_interval, uint8_t* ds_rgba, void* ds_distance, int ds_distance_type,
                                 const double* ds_Kinv_host, mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && width > 0 && height > 0, "rasterize_fwd: bad sizes");
  MGS_REQUIRE(channels >= 1 && channels <= MGS_MAX_CHANNELS, "rasterize_fwd: channels %d outside 1..%d", channels, MGS_MAX_CHANNELS);
  MGS_REQUIRE(tile_w == (width + 15) / 16 && tile_h == (height + 15) / 16,
              "rasterize_fwd: tile grid %dx%d does not match %dx%d at tile size 16", tile_w, tile_h, width, height);
  MGS_REQUIRE(!splats || channels <= 4, "rasterize_fwd: packed splats carry at most 4 channels");
  MGS_REQUIRE((n == 0 || splats || (means2d && conics && feats && opacities)) && tile_offsets &&
                  flatten_ids && ((render && alphas) || (ds_rgba && !render && !alphas)), "rasterize_fwd: null pointer");
  DatasetOut ds{};
  if (ds_rgba || ds_distance) {
    // the dataset frame (RGBA8 + ray distance) straight out of the raster: "RGB+ED" inference frames only
    MGS_REQUIRE(ds_rgba && channels == 4 && (flags & MGS_RASTER_EXPECTED_LAST) && !last_ids,
                "rasterize_fwd: the dataset output needs ds_rgba, 4 channels, MGS_RASTER_EXPECTED_LAST and no last_ids");
    MGS_REQUIRE(((uintptr_t)ds_rgba & 3) == 0, "rasterize_fwd: ds_rgba must be 4-byte aligned");
    MGS_REQUIRE(!ds_distance || ds_Kinv_host, "rasterize_fwd: the distance map needs K^-1 (host pointer, 9 doubles)");
    MGS_REQUIRE(ds_distance_type >= 0 && ds_distance_type <= 2, "rasterize_fwd: distance type %d not in {0: f32, 1: f64, 2: f16}",
                ds_distance_type);
    ds.rgba = reinterpret_cast<uint32_t*>(ds_rgba);
    ds.dist = ds_distance;
    ds.type = ds_distance_type;
    for (int i = 0; i < 9; ++i) ds.ki.m[i] = ds_Kinv_host ? ds_Kinv_host[i] : 0.0;
  }
  int ckpt_shift = 0;
  if (checkpoints) {
    MGS_REQUIRE(last_ids, "rasterize_fwd: checkpoints are written by the training variant (last_ids given)");
    MGS_REQUIRE(checkpoint_interval >= 64 && (checkpoint_interval & (checkpoint_interval - 1)) == 0,
                "rasterize_fwd: checkpoint_interval %d is not a power of two >= 64", checkpoint_interval);
    while ((1 << ckpt_shift) < checkpoint_interval) ++ckpt_shift;
  }
  const int n_tiles = tile_w * tile_h;
  hipStream_t s = (hipStream_t)stream;
  const bool per_block = (g_raster_opts & 4) || ((g_raster_opts & 2) && (flags & MGS_RASTER_LATENCY));
  if (g_raster_opts & 8) tile_group_order = nullptr;
  if ((g_raster_opts & 16) && !per_block && !last_ids) tile_group_order = nullptr;
  const int n_units = tile_group_order ? (n_tiles + 3) / 4 * 4 : n_tiles;       // tile slots of the launch
#define MGS_RF_LAUNCH_T(C, T)                                                                  \
  hipLaunchKernelGGL((raster_fwd_kernel<C, T>), dim3(div_up(n_units, (T) ? 1 : MGS_RASTER_WG_WAVES)),   \
                     dim3(64 * ((T) ? 1 : MGS_RASTER_WG_WAVES)), (T) ? 0 : (size_t)(g_raster_opts >> 8) * 1024, s, means2d, conics, \
                     feats, opacities, reinterpret_cast<const float4*>(splats), background,     \
                     channels, width, height, tile_w, n_tiles,                                 \
                     tile_offsets, flatten_ids, render, alphas, last_ids, g_raster_cull,            \
                     (flags & MGS_RASTER_EXPECTED_LAST) ? 1 : 0, g_raster_opts, tile_group_order, checkpoints, ckpt_shift, NoDataset{})
#define MGS_RQ_LAUNCH_T(C, T)                                                                  \
  hipLaunchKernelGGL((raster_fwd_q_kernel<C, T>), dim3(n_units), dim3(256), (size_t)(g_raster_opts >> 8) * 1024, s, means2d, conics, feats,  \
                     opacities, reinterpret_cast<const float4*>(splats), background, channels, width,      \
                     height, tile_w, n_tiles, tile_offsets, flatten_ids, render, alphas, last_ids,         \
                     g_raster_cull, (flags & MGS_RASTER_EXPECTED_LAST) ? 1 : 0, tile_group_order, checkpoints, ckpt_shift, NoDataset{})
#define MGS_RF_LAUNCH(C) do {                                                                      \
    if (per_block && (C) <= 4) { if (last_ids) MGS_RQ_LAUNCH_T(C, true); else MGS_RQ_LAUNCH_T(C, false); } \
    else if (last_ids) MGS_RF_LAUNCH_T(C, true); else MGS_RF_LAUNCH_T(C, false); } while (0)
  if (ds.rgba) {          // 4 channels, inference, "ED": the dataset instantiations of the two schedules
    if (per_block)
      hipLaunchKernelGGL((raster_fwd_q_kernel<4, false, true>), dim3(n_units), dim3(256), (size_t)(g_raster_opts >> 8) * 1024, s,
                         means2d, conics, feats, opacities, reinterpret_cast<const float4*>(splats), background, channels, width,
                         height, tile_w, n_tiles, tile_offsets, flatten_ids, render, alphas, last_ids, g_raster_cull, 1,
                         tile_group_order, checkpoints, ckpt_shift, ds);
    else
      hipLaunchKernelGGL((raster_fwd_kernel<4, false, true>), dim3(div_up(n_units, MGS_RASTER_WG_WAVES)),
                         dim3(64 * MGS_RASTER_WG_WAVES), (size_t)(g_raster_opts >> 8) * 1024, s, means2d, conics, feats, opacities,
                         reinterpret_cast<const float4*>(splats), background, channels, width, height, tile_w, n_tiles,
                         tile_offsets, flatten_ids, render, alphas, last_ids, g_raster_cull, 1, g_raster_opts, tile_group_order,
                         checkpoints, ckpt_shift, ds);
    return check_launch("rasterize_fwd");
  }
  if (channels == 1) MGS_RF_LAUNCH(1);
  else if (channels == 2) MGS_RF_LAUNCH(2);
  else if (channels == 3) MGS_RF_LAUNCH(3);
  else if (channels == 4) MGS_RF_LAUNCH(4);
  else if (channels <= 8) MGS_RF_LAUNCH(8);
  else if (channels <= 16) MGS_RF_LAUNCH(16);
  else MGS_RF_LAUNCH(32);
#undef MGS_RF_LAUNCH
#undef MGS_RF_LAUNCH_T
#undef MGS_RQ_LAUNCH_T
  return check_launch("rasterize_fwd");
}
