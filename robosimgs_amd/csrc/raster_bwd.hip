// raster_bwd.hip -- per-tile alpha compositing, backward (A.2 step 10), gfx950.
//
// Same geometry as the forward (raster_common.h): one wave64 per 16x16 tile, four pixels per
// lane, the tile's list walked BACK TO FRONT in batches of 64 with the same exact quadrant
// cull and ballot-compacted wave-private LDS queue.  Per Gaussian each lane sums, over its (up to) four pixels, the
// six moments of d loss / d sigma about the tile centre and the colour gradients (GaussGrad), and the wave reduces those
// 6 + channels partial sums.  Two ways out:
//   * records (mgs_rasterize_bwd_det, the render path's default): the sums go through a wave-private LDS transpose
//     (eight lanes finish one value each) and are stored as one record of `record_floats` floats per (tile, Gaussian)
//     pair; reduce_records_kernel turns every record into the pair's gradients (the slot tells the tile, hence the
//     mean's offset from its centre) and sums each Gaussian's contiguous slots.  No atomics, bit-reproducible.
//   * atomics (mgs_rasterize_bwd, for externally supplied tile lists): plain DPP reduction to
//     lane 63, one hardware float atomic per component.  Scattered device atomics sustain only
//     ~25-30 G/s on MI355X, which made this variant 1.38 ms against 0.78 ms for the records.
#include <algorithm>
#include <type_traits>

#include "raster_common.h"
#include "tile_order.h"

namespace mgs {
namespace {

template <int CHT, bool WIDE>
struct BwdEntry {
  float4 geo0;                       // q0, q1, q2 of the exponent's polynomial about the tile centre (raster_common.h), A
  float4 geo1;                       // B, C, quadrant mask (bits), list index (bits); A = -0.5 log2e a, B = -log2e b, C = -0.5 log2e c
  float4 feat[(CHT + 3) / 4];
  float4 geo2;                       // record slot / Gaussian id (bits), mean - tile centre (x, y), L = log2(opacity)
  float4 geo3[WIDE ? 1 : 0];         // conic a, b, c: absgrad and the atomic path only (64 bytes per entry without it at 4 channels)
};

template <int CHT>
struct BwdPixel {
  float T;            // transmittance in front of the Gaussian being processed
  float bv;           // (colour accumulated BEHIND the current Gaussian) . v_c  -  T_final * d loss / d alpha_out (with the
                      // background term): both only ever appear as this difference times 1 / (1 - alpha), so the walk
                      // starts bv at minus the second term instead of keeping it in a register of its own
  float v_c[CHT];     // d loss / d render
  int last;
};

// What one (tile, Gaussian) pair accumulates over the tile's pixels: the MOMENTS of v_sigma = d loss / d sigma about the
// tile centre (x, y = the pixel's offset from it, PixelPoly) and the colour gradient.  Everything else follows from
// them once per pair (moments_to_mean below): with d = m - (x, y), m = mean - tile centre,
//   sum v_sigma dx = m_x s - s_x,   sum v_sigma dx^2 = m_x (m_x s - s_x) - (m_x s_x - s_xx),   ...
// and opacity * d loss / d opacity = -s.  Round 2 accumulated sum p, sum q, sum p dx, ... with p = v_sigma dx per
// pixel: two subtractions and two products more per (pixel, Gaussian), plus an exp2 per pair for 1 / opacity.
template <int CHT>
struct GaussGrad {
  float s, s_x, s_y, s_xx, s_xy, s_yy, a_x, a_y;
  float v_f[CHT];
};

// One Gaussian against the 64 pixels of one quadrant, backward.  Wave-uniform skip when no lane
// contributes; inside, the per-lane condition is folded into two masked factors (alpha_eff and
// the opacity*vis product) so that every update is an unconditional FMA into the accumulators --
// the branchy form made the compiler zero-initialise and merge nine temporaries per quadrant.
// Algebraic savings over the textbook per-pixel form (A.2 step 10):
//   * colour only enters through dot products with the pixel's v_c, so the "colour behind"
//     buffer is kept as the scalar bv = buffer . v_c and the Gaussian's colour as fv = feat . v_c;
//   * the geometric gradients are polynomials in the pixel's offset from the tile centre times v_sigma: only the six
//     moments above are accumulated (the products x^2, xy, y^2 are per-lane constants the exponent's polynomial
//     already keeps in registers) and the CONSUMER turns them into gradients once per pair and applies the conic
//     once per Gaussian (moments_to_mean, finish_geo).
// SAFE (chosen per 64-entry batch, as in the forward): every queued Gaussian has a conic that cannot round sigma
// below zero and an opacity <= kSafeOpacity, so the sigma test and the 0.999 clamp are dead -- alpha = ov, nothing is
// ever clamped, one select serves a_eff and ov_eff.  Same values bit for bit, six vector instructions less.
template <int CHT, bool ABSGRAD, bool SAFE = false>
__device__ __forceinline__ void grad_pixel(BwdPixel<CHT>& px, GaussGrad<CHT>& gg, const PixelPoly& pp,
                                           float mx, float my, float ca, float cb,
                                           float cc, float A, float B, float C, float q0, float q1, float q2,
                                           const float* feat, int idx, int& any) {
  // mx, my: the mean's offset from the tile centre; pp: the pixel's (raster_common.h)
  // the forward's own evaluation (raster_fwd.hip blend_pixel, raster_common.h pair_power_poly), bit for bit:
  // ov = opacity * exp(-sigma) as one exp2 of the exponent's polynomial about the tile centre
  float ov = __builtin_amdgcn_exp2f(pair_power_poly(pp, q0, q1, q2, A, B, C));
  float alpha = SAFE ? ov : fminf(kAlphaMax, ov);
  bool valid = idx <= px.last && alpha >= kAlphaMin;
  if (!SAFE) valid = valid && pair_power_sign(mx - pp.x, my - pp.y, A, B, C) <= 0.f;
  if (ballot(valid) == 0ull) return;
  any = 1;        // (wave-uniform, set on the taken side of a scalar branch: one s_mov.  As a returned bool the compiler
                  //  rebuilt it from the lane mask with a v_cndmask, a v_cmp and two scalar instructions per quadrant body)
  float a_eff = valid ? alpha : 0.f;                       // 0 => T, bv and v_f stay untouched
  bool grad_geo = valid && ov <= kAlphaMax;                // alpha not clamped: sigma/opacity get grads
  float ov_eff = SAFE ? a_eff : (grad_geo ? ov : 0.f);
  float ra = __builtin_amdgcn_rcpf(1.0f - a_eff);          // v_rcp_f32; an IEEE divide is 11 instructions
  px.T *= ra;
  float fac = a_eff * px.T;
  float fv = feat[0] * px.v_c[0];
#pragma unroll
  for (int c = 1; c < CHT; ++c) fv = fmaf(feat[c], px.v_c[c], fv);
  float v_alpha = fmaf(-px.bv, ra, fv * px.T);
  asm volatile("" : "+v"(px.bv), "+v"(v_alpha));   // bv is updated in place AFTER its last use (the compiler formed
  px.bv = fmaf(fv, fac, px.bv);                    // the new value early in a temporary and copied it back)
#pragma unroll
  for (int c = 0; c < CHT; ++c) gg.v_f[c] = fmaf(fac, px.v_c[c], gg.v_f[c]);
  float v_sigma = -ov_eff * v_alpha;
  gg.s += v_sigma;
  gg.s_x = fmaf(v_sigma, pp.x, gg.s_x);
  gg.s_y = fmaf(v_sigma, pp.y, gg.s_y);
  gg.s_xx = fmaf(v_sigma, pp.xx, gg.s_xx);
  gg.s_xy = fmaf(v_sigma, pp.xy, gg.s_xy);
  gg.s_yy = fmaf(v_sigma, pp.yy, gg.s_yy);
  if (ABSGRAD) {
    float p = v_sigma * (mx - pp.x), q = v_sigma * (my - pp.y);
    gg.a_x += fabsf(fmaf(cb, q, ca * p));
    gg.a_y += fabsf(fmaf(cc, q, cb * p));
  }
}

// Moments of v_sigma about the tile centre -> sums about the Gaussian's mean, m = mean - tile centre:
//   P = sum v_sigma dx, Q = sum v_sigma dy, Vaa = sum v_sigma dx^2, Vab = sum v_sigma dx dy, Vbb = sum v_sigma dy^2.
// Written as nested differences (m_x P - (m_x s_x - s_xx)) so that the large m^2 s term never appears on its own.
__device__ __forceinline__ void moments_to_mean(float mx, float my, float s, float s_x, float s_y, float s_xx,
                                                float s_xy, float s_yy, float& P, float& Q, float& Vaa, float& Vab,
                                                float& Vbb) {
  P = fmaf(mx, s, -s_x);
  Q = fmaf(my, s, -s_y);
  Vaa = fmaf(mx, P, -fmaf(mx, s_x, -s_xx));
  Vab = fmaf(my, P, -fmaf(mx, s_y, -s_xy));
  Vbb = fmaf(my, Q, -fmaf(my, s_y, -s_yy));
}

// Record layout: one record per (tile, Gaussian) slot, `record_floats` floats apart -- the 6 moments, the CHT colour
// gradients (channels past `channels` are zero), the absgrad pair, padded to a multiple of FOUR floats so that the
// reduce kernel reads a record as 16-byte pieces (three `global_load_dwordx4` per slot at 4 channels instead of ten
// scattered `global_load_dword`: its lanes each walk their own Gaussian's slots, so every load instruction touches 64
// different places and the kernel is bound by the number of those, not by bytes).  A value-major layout (value p of
// slot s at records[p * n_slots + s]) was measured in round 3: the ten 4-byte stores of a record then land in ten
// different lines and the raster backward pays more than the reduce gains (547 -> 686 us).
__host__ __device__ constexpr int record_floats(int cht, bool absgrad) { return (6 + cht + (absgrad ? 2 : 0) + 3) / 4 * 4; }
__host__ __device__ constexpr int padded_channels(int channels) {
  return channels <= 4 ? channels : channels <= 8 ? 8 : channels <= 16 ? 16 : 32;
}

// Sums of grad_pixel's raw geometric accumulators -> gradients of mean2d and conic.
__device__ __forceinline__ void finish_geo(float ca, float cb, float cc, float& v_x, float& v_y,
                                           float& v_ca, float& v_cc) {
  const float s0 = v_x, s1 = v_y;
  v_x = fmaf(cb, s1, ca * s0);
  v_y = fmaf(cc, s1, cb * s0);
  v_ca *= 0.5f;
  v_cc *= 0.5f;
}

// RECORDS == false: lane 63 adds the wave-reduced sums to the per-Gaussian outputs with float
// atomics.  RECORDS == true: it stores them as one record at the pair's slot (pair_info) and
// sets the slot's flag; reduce_records_kernel then sums each Gaussian's slots -- no atomics,
// bit-reproducible.
#ifndef MGS_RASTER_BWD_WG_WAVES
#define MGS_RASTER_BWD_WG_WAVES 1      // independent tiles (waves) per workgroup; 2 / 4 measured slower (564 / 554 vs 537 us)
#endif
#ifndef MGS_RASTER_BWD_MIN_WAVES
// min waves per SIMD asked of the register allocator for the record kernels of up to 4 channels without absgrad.
// 5 = at most 96 VGPRs (113 when left free: 4 waves), 28 bytes spilled outside the walk; the kernel's LDS (64-byte queue
// entries, ten rows of reduction buffer: 6.5 KB per wave) allows 24 waves per CU.  Measured with dynamic-LDS caps:
// 3 waves per SIMD 544 us, 4 waves 474-484, 5 waves 468-475 (memset + backward + reduce); 6 waves (80 VGPRs) spill: 694.
// The round-3 first pass tried 96 VGPRs with 9 KB of LDS per wave -- 17 waves per CU at most, so only the spills showed.
#define MGS_RASTER_BWD_MIN_WAVES 5
#endif
// HALF (record path only; measured and NOT the default): a wave owns HALF a tile -- two 8x8 blocks side by
// side, two pixels per lane -- and the pair's record slot is doubled (slot * 2 + half): twice the waves, each
// half the serial work, 96 instead of 112 VGPRs, against one more wave reduction and record for the pairs
// that reach both halves.  618 -> 724 us for the whole backward at config 2: the extra reductions and the
// doubled slots of the reduce cost more than the finer schedule returns (what paid in the forward, where a
// block's result needs no cross-lane sum, does not pay here).
#ifndef MGS_RASTER_BWD_LDS_PAD
#define MGS_RASTER_BWD_LDS_PAD 0       // bytes of unused dynamic LDS per workgroup: caps the waves per CU (occupancy experiments)
#endif
#ifndef MGS_RASTER_BWD_HALF
#define MGS_RASTER_BWD_HALF 0
#endif
// PIPE (record path, 9..16 reduced values): the wave reduction of list entry j is finished while entry j+1 is
// evaluated -- the partial sums parked in LDS by entry j are read back at the top of the next trip, the ~100
// vector instructions of that entry's evaluation cover the round trip, and the sums are then finished and
// stored.  Summation order and records are unchanged.  Measured and NOT the default: 572.4 us with it, 569.3 without --
// the kernel is not waiting on that round trip.
#ifndef MGS_RASTER_BWD_PIPE
#define MGS_RASTER_BWD_PIPE 0
#endif
#ifndef MGS_RASTER_BWD_QUAD
#define MGS_RASTER_BWD_QUAD 1
#endif
#ifndef MGS_RASTER_BWD_XCD_RUN
#define MGS_RASTER_BWD_XCD_RUN 4       // segmented launch: consecutive units per XCD (see raster_bwd_kernel); 1 = plain numbering.  FETCH_SIZE 422 / 387 / 360 / 344 MiB for 1 / 2 / 4 / 8, same time
#endif
#ifndef MGS_RASTER_BWD_ORDER
#define MGS_RASTER_BWD_ORDER 1         // launch the tiles by falling list length (tile_order_kernel): 623 -> 572 us
#endif
#ifdef MGS_RASTER_BWD_TIMING          // measurement build (scripts/dbg/bwd_timeline.py): per tile {start, end} on the 100 MHz clock, entries, pairs
__device__ unsigned long long g_bwd_times[5 * 65536];   // per unit: enter, walk begins, end, entries, pairs
#endif
// SPLIT (record path, <= 4 channels): a unit of the launch is one SEGMENT of a tile's list (raster_common.h:
// checkpoints), found through seg_table; a segment that has a successor some pixel reaches starts those pixels from the
// forward's checkpoint -- T as the forward had it, colour behind = final - checkpoint -- instead of from the end of the
// list.  Every list entry lies in exactly one segment and the batches are those of the whole walk, so records, slots
// and the reduce are unchanged; only the first T and bv of a front segment differ from the whole walk's (by rounding:
// the forward's T is the exact product, the whole walk's a chain of v_rcp_f32).  Why: a tile used to be one wave's
// serial job, 8,160 jobs of 200-560 evaluated pairs are two uneven rounds of the wave slots, and the last fifth of
// the launch ran at under one wave per SIMD.
// Tables of the segmented launch (unit_table_kernel below): 64 counters, the whole segments, the partial ones by class.
constexpr int kUnitClasses = 32;
struct UnitTables {
  uint32_t* counts;      // [0] whole segments, [1 + c] partial segments of class c
  int4* whole;           // {tile, segment, the list's start, the end of the tile's walk (its largest last_id)}
  int4* part;            // [class][n_tiles]
};
__host__ __device__ inline size_t unit_tables_bytes(int n_tiles, int shift, uint32_t capacity) {
  return 256 + (ckpt_units(capacity, n_tiles, shift) + (size_t)kUnitClasses * n_tiles) * sizeof(int4);
}
__device__ __forceinline__ UnitTables unit_tables(const int32_t* base, int n_tiles, int shift, uint32_t capacity) {
  UnitTables u;
  u.counts = reinterpret_cast<uint32_t*>(const_cast<int32_t*>(base));
  u.whole = reinterpret_cast<int4*>(const_cast<int32_t*>(base) + 64);
  u.part = u.whole + ckpt_units(capacity, n_tiles, shift);
  return u;
}

template <int CHT, bool ABSGRAD, bool RECORDS, bool HALF = false, bool SPLIT = false>
__device__ __forceinline__ void raster_bwd_unit(const int unit,
    const float* __restrict__ means2d, const float* __restrict__ conics,
    const float* __restrict__ feats, const float* __restrict__ opacities,
    const float4* __restrict__ splats, const float* __restrict__ background, int channels,
    int width, int height, int tile_w, int n_tiles, const int32_t* __restrict__ tile_offsets,
    const int32_t* __restrict__ flatten_ids, const float* __restrict__ alphas,
    const int32_t* __restrict__ last_ids, const float* __restrict__ v_render,
    const float* __restrict__ v_alphas, float* __restrict__ v_means2d,
    float* __restrict__ v_means2d_abs, float* __restrict__ v_conics,
    float* __restrict__ v_feats, float* __restrict__ v_opacities,
    const int4* __restrict__ pair_info, float* __restrict__ records,
    uint8_t* __restrict__ flags, uint32_t capacity, const float* __restrict__ expected_render,
    const int32_t* __restrict__ tile_order, const float* __restrict__ ckpt, int ckpt_shift,
    const int32_t* __restrict__ seg_table, const float* __restrict__ render_out, int splat_slots) {
  constexpr bool WIDE = ABSGRAD || !RECORDS;
  constexpr int NVR = 6 + CHT + (ABSGRAD ? 2 : 0);      // values reduced over the wave per list entry
  constexpr int kRedRows = !RECORDS ? 1 : (NVR > 8 && NVR <= 16) ? NVR : 8;
  __shared__ BwdEntry<CHT, WIDE> queues[MGS_RASTER_BWD_WG_WAVES][kQueue];
  __shared__ float reds[MGS_RASTER_BWD_WG_WAVES][kRedRows][64];  // wave-private transpose buffer of the record reduction
  BwdEntry<CHT, WIDE>* queue = queues[threadIdx.x >> 6];
  float (*red)[64] = reds[threadIdx.x >> 6];
  static_assert(!HALF || RECORDS, "half tiles exist on the record path only");
  static_assert(!SPLIT || (RECORDS && !HALF && CHT <= 4), "segments exist on the record path of up to 4 channels");
  constexpr int NQ = HALF ? 2 : 4;                 // 8x8 blocks per wave
#ifdef MGS_RASTER_BWD_TIMING
  const unsigned long long t_enter = __builtin_amdgcn_s_memrealtime();
#endif
  // tile_order: groups of four tiles by falling list length (tile_order.h), so that the longest walks start first
  int tile, seg = 0, seg_start = 0, seg_hi = 0;
  if constexpr (SPLIT) {
    // the launch is sized by the capacity; the tables (unit_table_kernel) list the segments some pixel reaches: whole
    // segments first, then every tile's last, partly walked one by falling length class
    const UnitTables ut = unit_tables(seg_table, n_tiles, ckpt_shift, capacity);
    uint32_t cnt[1 + kUnitClasses];                   // (wave-uniform: 33 SGPRs from three scalar loads, no dependent chain)
#pragma unroll
    for (int i = 0; i <= kUnitClasses; ++i) cnt[i] = ut.counts[i];
    const int4 e_whole = ut.whole[unit];              // (in flight with the counts: two thirds of the units are whole segments)
    int r = unit - (int)cnt[0], cls = -1, at = 0;
#pragma unroll
    for (int c = 0; c < kUnitClasses; ++c) {
      const int n = (int)cnt[1 + c];
      if (cls < 0 && r >= 0 && r < n) { cls = c; at = r; }
      if (r >= 0) r -= n;
    }
    if (r >= 0 && cls < 0) return;                    // past the last unit (the launch is sized by the capacity)
    const int4 e = cls < 0 ? e_whole : ut.part[(size_t)cls * n_tiles + at];
    tile = e.x;
    seg = e.y;
    seg_start = e.z;
    seg_hi = e.w;
  } else {
    tile = HALF ? (unit >> 1 < n_tiles ? unit >> 1 : -1) : tile_of_unit(unit, n_tiles, tile_order);
  }
  const int half = HALF ? unit & 1 : 0;
  if (tile < 0) return;
  const unsigned lane = threadIdx.x & 63u;
  const int tx = tile % tile_w, ty = tile / tile_w;
  const float tile_x = (float)(tx * 16), tile_y = (float)(ty * 16);
  const int start = SPLIT ? seg_start : tile_offsets[tile], end = SPLIT ? seg_hi + 1 : tile_offsets[tile + 1];
  if (end <= start) return;
  // SPLIT: this unit's segment [lo, nxt) of the list (the table entry carries the list's start and the end of the walk)
  const int lo = SPLIT ? start + (seg << ckpt_shift) : start;
  const int nxt = SPLIT ? lo + (1 << ckpt_shift) : end;
  // (also consumes `capacity` up here: a scalar load still outstanding at the head of the walk would turn every
  //  LDS wait inside it into a wait for everything -- scalar loads return out of order)
  if (RECORDS && capacity == 0u) return;
  const int ix = tx * 16 + (int)(lane & 7), iy = ty * 16 + 8 * half + (int)(lane >> 3);
  // this lane's pixels as offsets from the tile centre, with their products (raster_common.h: PixelPoly)
  PixelPoly pq[NQ];
#pragma unroll
  for (int k = 0; k < NQ; ++k)
    pq[k] = pixel_poly((float)(lane & 7) - 7.5f + 8.f * (k & 1), (float)(8 * half + (int)(lane >> 3)) - 7.5f + 8.f * (k >> 1));

  BwdPixel<CHT> st[NQ];
  int hi = -1;
  bool has_next = false;           // (wave-uniform) some pixel's list goes on behind this segment: checkpoints apply
  const float* cp = nullptr;
  if constexpr (SPLIT) {
    hi = seg_hi;                     // where the tile's walk ends (the forward left it in the checkpoint header)
    if (hi < lo) return;             // (the table lists no such unit)
    has_next = hi >= nxt;
    cp = ckpt + ckpt_header_floats(n_tiles) + ckpt_unit(start, tile, seg + 1, ckpt_shift) * (size_t)(1 + channels) * 256 + lane;
    hi = min(hi, nxt - 1);           // this unit walks [lo, min(hi, nxt - 1)]
  }
  // the frame's values of this lane's pixels: every load is issued before the first use (one round trip, not one per
  // quadrant: a segment's prologue is paid three to five times per tile)
  bool inside[NQ];
  float px_alpha[NQ], px_va[NQ], px_fin[NQ][CHT], px_ck[NQ][CHT + 1];
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    const int x = ix + 8 * (k & 1), y = iy + 8 * (k >> 1);
    inside[k] = x < width && y < height;
    const size_t p = inside[k] ? (size_t)y * width + x : 0;
    px_alpha[k] = inside[k] ? alphas[p] : 0.f;
    st[k].last = inside[k] ? last_ids[p] : -1;
    px_va[k] = (inside[k] && v_alphas) ? v_alphas[p] : 0.f;      // v_alphas == nullptr: no loss term on alpha
    if (CHT == 4 && channels == 4) {           // (16-byte rows)
      const float4 v = inside[k] ? reinterpret_cast<const float4*>(v_render)[p] : make_float4(0.f, 0.f, 0.f, 0.f);
      st[k].v_c[0] = v.x; st[k].v_c[1 % CHT] = v.y; st[k].v_c[2 % CHT] = v.z; st[k].v_c[3 % CHT] = v.w;
    } else {
#pragma unroll
      for (int c = 0; c < CHT; ++c)
        st[k].v_c[c] = (inside[k] && c < channels) ? v_render[p * channels + c] : 0.f;
    }
    // the forward's frame: "ED" needs its last channel, a segment with a successor all of it
    const float* frame = SPLIT ? (has_next ? render_out : expected_render) : expected_render;
#pragma unroll
    for (int c = 0; c < CHT; ++c) px_fin[k][c] = 0.f;
    if (frame && inside[k]) {
      if (CHT == 4 && channels == 4) {
        const float4 v = reinterpret_cast<const float4*>(frame)[p];
        px_fin[k][0] = v.x; px_fin[k][1 % CHT] = v.y; px_fin[k][2 % CHT] = v.z; px_fin[k][3 % CHT] = v.w;
      } else {
#pragma unroll
        for (int c = 0; c < CHT; ++c)
          if (c < channels && (SPLIT ? (has_next || c == channels - 1) : c == channels - 1)) px_fin[k][c] = frame[p * channels + c];
      }
    }
    if constexpr (SPLIT) {
      // (unconditional per lane: the buffer exists for every unit; what a finished block never wrote is not used)
#pragma unroll
      for (int c = 0; c <= CHT; ++c) px_ck[k][c] = (has_next && c <= channels) ? cp[256 * c + 64 * k] : 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    st[k].T = 1.0f - px_alpha[k];                        // starts at the pixel's final transmittance (1 outside the image)
    float va = px_va[k];
    if (expected_render && inside[k]) {
      // the forward divided the last channel by max(alpha, 1e-10) ("ED"): ED = D / a, so
      // dL/dD = v_ED / a and dL/dalpha -= v_ED ED / a (where a > 1e-10)
      const float a = px_alpha[k], inv = 1.0f / fmaxf(a, 1e-10f);
#pragma unroll
      for (int c = 0; c < CHT; ++c)
        if (c == channels - 1) {
          const float v_ed = st[k].v_c[c];
          if (a > 1e-10f) va -= v_ed * px_fin[k][c] * inv;
          st[k].v_c[c] = v_ed * inv;
        }
    }
#pragma unroll
    for (int c = 0; c < CHT; ++c)
      if (background && c < channels) va -= background[c] * st[k].v_c[c];
    st[k].bv = -(st[k].T * va);
    if constexpr (SPLIT) {
      // a pixel whose last contributor lies in or past the next segment starts from the forward's state in front of
      // that segment: T as stored; colour behind = final - accumulated so far.  The final sums are taken back out of
      // the frame: render = C + T_final * background, last channel times 1 / max(alpha, 1e-10) in "ED" mode.
      if (has_next && st[k].last >= nxt) {
        const float a = px_alpha[k], t_final = st[k].T;
        float behind = 0.f;
#pragma unroll
        for (int c = 0; c < CHT; ++c)
          if (c < channels) {
            float fin = px_fin[k][c];
            if (expected_render && c == channels - 1) fin *= fmaxf(a, 1e-10f);
            if (background) fin = fmaf(-t_final, background[c], fin);
            behind = fmaf(fin - px_ck[k][1 + c], st[k].v_c[c], behind);
          }
        st[k].bv += behind;
        st[k].T = px_ck[k][0];
      }
    } else {
      hi = max(hi, st[k].last);
    }
  }
  if constexpr (!SPLIT) {
    // wave-wide maximum of the last contributing index
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) hi = max(hi, __shfl_xor(hi, d));
    hi = min(hi, end - 1);
    if (hi < start) return;
  }
#ifdef MGS_RASTER_BWD_TIMING
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
  unsigned long long n_walked = 0, n_pairs = 0;
#endif
#ifdef MGS_RASTER_BWD_PRIO
  {   // issue priority by the length of the walk (see raster_fwd.hip)
    const int avg = tile_offsets[n_tiles] / n_tiles, len = hi - start + 1;
    if (len > 2 * avg) __builtin_amdgcn_s_setprio(3);
    else if (2 * len > 3 * avg) __builtin_amdgcn_s_setprio(2);
    else if (len > avg) __builtin_amdgcn_s_setprio(1);
  }
#endif

  constexpr int NV = 6 + CHT + (ABSGRAD ? 2 : 0);      // values reduced over the wave per list entry
  constexpr bool PIPE = RECORDS && MGS_RASTER_BWD_PIPE != 0 && NV > 8 && NV <= 16;
  // QUAD (9..16 reduced values): FOUR lanes finish a value -- each adds a quarter of the value's row, two DPP steps, and
  // the NV totals leave in one store instruction; the other form gives eight lanes a value and takes two groups of
  // eight values side by side (the second group mostly idle at 10 values): 4 + 15 + 2 + 1 instead of 4 + 14 + 6 + 2
  constexpr bool QUAD = MGS_RASTER_BWD_QUAD != 0 && NV > 8 && NV <= 16;
  constexpr int RSP = record_floats(CHT, ABSGRAD);     // floats per record (stride)
  // record position of value j: the values in the order they are reduced (padding channels hold zeros)
  auto rec_pos = [&](int j) { return j; };
  // the two halves of the 9..16-value reduction (see below): read back the parked partial sums ...
  auto red_load = [&](float4& a0, float4& b0, float4& a1, float4& b1) {
    if constexpr (QUAD) {
      // the lane's quarter of its value's row, 16-byte piece ((value & 3) ^ step) at each step: in every step the 16
      // lanes the LDS serves together -- four values x four quarters -- read 16 different columns of the 64 banks
      // (rows are 256 bytes apart: all rows alias, only the column tells lanes apart)
      const int qv = (int)(lane >> 2), x = qv & 3;
      const float* q = &red[qv < NV ? qv : NV - 1][16 * (lane & 3)];
      a0 = *reinterpret_cast<const float4*>(q + 4 * x);
      b0 = *reinterpret_cast<const float4*>(q + 4 * (x ^ 1));
      a1 = *reinterpret_cast<const float4*>(q + 4 * (x ^ 2));
      b1 = *reinterpret_cast<const float4*>(q + 4 * (x ^ 3));
      return;
    }
    const int v1 = 8 + (int)(lane >> 3);                          // second group's value for this lane
    const float4* s0 = reinterpret_cast<const float4*>(&red[lane >> 3][(lane & 7) * 8]);
    const float4* s1 = reinterpret_cast<const float4*>(&red[v1 < NV ? v1 : 8][(lane & 7) * 8]);
    a0 = s0[0]; b0 = s0[1]; a1 = s1[0]; b1 = s1[1];
  };
  // ... and finish the sums and store the record
  auto red_finish = [&](size_t rslot, const float4& a0, const float4& b0, const float4& a1, const float4& b1) {
    if constexpr (QUAD) {
      float t = (((a0.x + a0.y) + (a0.z + a0.w)) + ((b0.x + b0.y) + (b0.z + b0.w))) +
                (((a1.x + a1.y) + (a1.z + a1.w)) + ((b1.x + b1.y) + (b1.z + b1.w)));
      // (as text, see below: the compiler sinks the last add into the storing lanes' branch and unfolds its DPP move)
      asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(t));
      // (as text: the record's address is uniform + the lane's word, and the store takes the uniform part from an SGPR
      //  pair -- the compiler formed a 64-bit address per lane with a quarter-rate v_mad_i64_i32 per pair)
      const float* rec = records + rslot * RSP;
      if ((lane & 3) == 0 && (int)(lane >> 2) < NV)                          // one store: NV consecutive words
        asm volatile("global_store_dword %0, %1, %2" : : "v"((unsigned)(lane >> 2) * 4u), "v"(t), "s"(rec) : "memory");
      if (lane == 0) flags[rslot] = 1;
      return;
    }
    const int v1 = 8 + (int)(lane >> 3);
    float t0 = ((a0.x + a0.y) + (a0.z + a0.w)) + ((b0.x + b0.y) + (b0.z + b0.w));
    float t1 = ((a1.x + a1.y) + (a1.z + a1.w)) + ((b1.x + b1.y) + (b1.z + b1.w));
    t0 += dpp_f(t0, kDppXor1);
    t1 += dpp_f(t1, kDppXor1);
    t0 += dpp_f(t0, kDppXor2);
    t1 += dpp_f(t1, kDppXor2);
    // (as text: the compiler sinks the last add into the storing lanes' branch and then cannot fold the DPP move into
    //  it -- two v_mov 0, two v_mov_dpp and two adds instead of two v_add_f32_dpp)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(t0), "+v"(t1));
    // (uniform record base + the lane's 32-bit position: the store takes the base from SGPRs instead of a 64-bit
    //  address per lane kept across the walk)
    float* rec = records + rslot * RSP;
    if ((lane & 7) == 0) {
      rec[lane >> 3] = t0;
      if (v1 < NV) rec[(unsigned)v1] = t1;
    }
    if (lane == 0) flags[rslot] = 1;
  };
  bool pend = false;            // PIPE: an entry's partial sums are parked in `red`, its record not yet stored
  size_t pend_slot = 0;

  for (int q = (hi - start) / kQueue; q >= (lo - start) / kQueue; --q) {
    const int b = start + q * kQueue;
    // quadrants that still have a pixel with something left at or above this batch
    unsigned live = 0;
#pragma unroll
    for (int k = 0; k < NQ; ++k)
      if (ballot(st[k].last >= b) != 0ull) live |= 1u << k;
    if (live == 0) continue;

    const int idx = b + (int)lane;
    unsigned qmask = 0;
    int g = 0;
    float2 xy = make_float2(0.f, 0.f);
    float ca = 1.f, cb = 0.f, cc = 1.f, op = 0.f;
    float pf[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t slot_base = 0u, slot_rect = 0u;          // splat_slots: the pair's record slot out of the record itself
    const bool packed = CHT <= 4 && splats != nullptr;
    if (idx <= hi) {
      g = flatten_ids[idx];
      if (packed) {
        const float4 p0 = splats[3 * (size_t)g], p1 = splats[3 * (size_t)g + 1], p2 = splats[3 * (size_t)g + 2];
        xy = make_float2(p0.x, p0.y);
        ca = p0.z; cb = p0.w; cc = p1.x; op = p1.y;
        pf[0] = p1.z; pf[1] = p1.w; pf[2] = p2.x; pf[3] = p2.y;
        slot_base = __float_as_uint(p2.z); slot_rect = __float_as_uint(p2.w);
      } else {
        xy = reinterpret_cast<const float2*>(means2d)[g];
        ca = conics[3 * (size_t)g + 0];
        cb = conics[3 * (size_t)g + 1];
        cc = conics[3 * (size_t)g + 2];
        op = opacities[g];
      }
      qmask = quadrant_mask(xy.x, xy.y, ca, cb, cc, op, tile_x, tile_y);
      if (HALF) qmask = (qmask >> (2 * half)) & 3u;       // this half's two blocks
      qmask &= live;
    }
    const unsigned long long keep = ballot(qmask != 0u);
    const int count = __popcll(keep);
#ifdef MGS_RASTER_BWD_TIMING
    n_walked += (unsigned long long)min(64, hi - b + 1);
    n_pairs += (unsigned long long)count;
#endif
    const bool all_safe = ballot(qmask != 0u && !entry_is_safe(ca, cb, cc, op)) == 0ull;
    if (qmask != 0u) {
      BwdEntry<CHT, WIDE>& e = queue[mask_rank(keep)];
      constexpr float kLog2e = 1.4426950408889634f;
      const float sA = -0.5f * kLog2e * ca, sB = -kLog2e * cb, sC = -0.5f * kLog2e * cc, L = __log2f(op);
      const float m_x = xy.x - (tile_x + 8.f), m_y = xy.y - (tile_y + 8.f);
      const PolyCoef pc = poly_coefs(m_x, m_y, sA, sB, sC, L);
      e.geo0 = make_float4(pc.q0, pc.q1, pc.q2, sA);
      e.geo1 = make_float4(sB, sC, __uint_as_float(qmask), __int_as_float(idx));
      int gid = g;
      if (RECORDS) {
        if (packed && splat_slots) {        // (uniform) mgs_isect_tiles left {slot base, x0 | y0 << 10 | w << 20} in the record
          gid = (int)slot_base + (ty - (int)((slot_rect >> 10) & 1023u)) * (int)(slot_rect >> 20) + (tx - (int)(slot_rect & 1023u));
        } else {
          const int4 info = pair_info[g];
          gid = info.x + (ty - info.z) * (info.w & 0xffff) + (tx - info.y);   // the pair's slot
        }
      }
      e.geo2 = make_float4(__int_as_float(gid), m_x, m_y, L);
      if constexpr (WIDE) e.geo3[0] = make_float4(ca, cb, cc, 0.f);
      float f[((CHT + 3) / 4) * 4];
#pragma unroll
      for (int c = 0; c < ((CHT + 3) / 4) * 4; ++c)
        f[c] = (c < CHT && c < channels) ? (packed ? pf[c & 3] : feats[(size_t)g * channels + c]) : 0.f;
#pragma unroll
      for (int j = 0; j < (CHT + 3) / 4; ++j)
        e.feat[j] = make_float4(f[4 * j], f[4 * j + 1], f[4 * j + 2], f[4 * j + 3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    auto walk = [&](auto safe_tag) {
    constexpr bool SAFE = decltype(safe_tag)::value;
    for (int j = count - 1; j >= 0; --j) {
      const BwdEntry<CHT, WIDE>& e = queue[j];
      const float4 g0 = e.geo0, g1 = e.geo1, g2 = e.geo2;
      float4 g3 = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (WIDE) g3 = e.geo3[0];
      float feat[CHT];
#pragma unroll
      for (int f = 0; f < (CHT + 3) / 4; ++f) {
        float4 v = e.feat[f];
        feat[4 * f] = v.x;
        if (4 * f + 1 < CHT) feat[4 * f + 1] = v.y;
        if (4 * f + 2 < CHT) feat[4 * f + 2] = v.z;
        if (4 * f + 3 < CHT) feat[4 * f + 3] = v.w;
      }
      float4 pa0, pb0, pa1, pb1;
      if (PIPE) {     // unconditional (stale words when nothing is pending): a branch here would make the compiler
        red_load(pa0, pb0, pa1, pb1);         // wait for these loads together with the entry's own at the join
        asm volatile("" ::: "memory");        // keep the read-back up here, ahead of the evaluation
      }
      const unsigned m = __builtin_amdgcn_readfirstlane(__float_as_uint(g1.z));
      const int gi = __float_as_int(g1.w);
      const int gid = __builtin_amdgcn_readfirstlane(__float_as_int(g2.x));
      GaussGrad<CHT> gg;
      gg.s = gg.s_x = gg.s_y = gg.s_xx = gg.s_xy = gg.s_yy = gg.a_x = gg.a_y = 0.f;
#pragma unroll
      for (int c = 0; c < CHT; ++c) gg.v_f[c] = 0.f;
      int any = 0;
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        if (m & (1u << k))
          grad_pixel<CHT, ABSGRAD, SAFE>(st[k], gg, pq[k], g2.y, g2.z, g3.x, g3.y, g3.z, g0.w, g1.x, g1.y,
                                         g0.x, g0.y, g0.z, feat, gi, any);
      }
      if (PIPE && pend) {
        red_finish(pend_slot, pa0, pb0, pa1, pb1);
        pend = false;
      }
      if (!any) continue;          // (uniform: no lane of any quadrant took the Gaussian -- nothing to reduce)
      if constexpr (RECORDS) {
        // overflowed tile lists (status word set by the binning): slot bases run up to the true
        // n_isect, the workspace only to the capacity -- nothing is written past it
        if ((uint32_t)gid >= capacity) continue;
        const size_t rslot = HALF ? 2 * (size_t)gid + half : (size_t)gid;
        // reduce-scatter butterfly: 8 values at a time, totals land in 8 lanes that store the
        // record slice with one instruction
        float vals[NV];
        // the record holds the pair's raw moments; reduce_records_kernel turns them into gradients (it knows the
        // pair's tile from the slot, hence m)
        vals[0] = gg.s; vals[1] = gg.s_x; vals[2] = gg.s_y; vals[3] = gg.s_xx;
        vals[4] = gg.s_xy; vals[5] = gg.s_yy;
#pragma unroll
        for (int c = 0; c < CHT; ++c) vals[6 + c] = gg.v_f[c];
        if constexpr (ABSGRAD) { vals[6 + CHT] = gg.a_x; vals[7 + CHT] = gg.a_y; }
        int done = 0;
        // Through LDS, eight values per group: every lane parks its partial sums lane-linear
        // (red[i][lane], conflict-free), lane L then reads the eight partials red[L >> 3][8 (L & 7) ..]
        // with two ds_read_b128, adds them, and three DPP steps finish the sum over the eight lanes
        // that share a value: 10 VALU per eight values against 26 for the all-DPP butterfly (LDS
        // instructions of one wave execute in order, so the wave-private buffer needs no barrier
        // beyond the compiler fences).  Up to 16 values (every case up to 8 channels) go in ONE round
        // trip: all values are parked at once and each lane finishes two groups side by side -- the
        // tail values used to take a DPP chain plus two ds_bpermute round trips per list entry.
        if constexpr (NV > 8 && NV <= 16) {
#pragma unroll
          for (int i = 0; i < NV; ++i) red[i][lane] = vals[i];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if constexpr (PIPE) {
            pend = true;                            // finished during the next entry (or after the walk)
            pend_slot = rslot;
            continue;
          } else {
            float4 a0, b0, a1, b1;
            red_load(a0, b0, a1, b1);
            red_finish(rslot, a0, b0, a1, b1);
            __builtin_amdgcn_wave_barrier();        // the next entry overwrites red
            continue;
          }
        }
        while (NV - done >= 8) {
#pragma unroll
          for (int i = 0; i < 8; ++i) red[i][lane] = vals[done + i];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const float4* src = reinterpret_cast<const float4*>(&red[lane >> 3][(lane & 7) * 8]);
          const float4 a = src[0], b = src[1];
          float t = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
          t += dpp_f(t, kDppXor1);
          t += dpp_f(t, kDppXor2);
          t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x141, 0xf, 0xf, false));  // row_half_mirror
          if ((lane & 7) == 0) {
            int pos = rec_pos(done + (int)(lane >> 3));
            if (pos >= 0) records[rslot * RSP + pos] = t;
          }
          __builtin_amdgcn_wave_barrier();          // the next round overwrites red
          done += 8;
        }
#define MGS_RS_CHUNK(V)                                                        \
        while (NV - done >= V) {                                               \
          float t = wave_reduce_scatter<V>(vals + done, lane);                 \
          int i = wave_reduce_scatter_index<V>(lane);                          \
          if (i >= 0) {                                                        \
            int pos = rec_pos(done + i);                                       \
            if (pos >= 0) records[rslot * RSP + pos] = t;                                        \
          }                                                                    \
          done += V;                                                           \
        }
        MGS_RS_CHUNK(4)
        MGS_RS_CHUNK(2)
        MGS_RS_CHUNK(1)
#undef MGS_RS_CHUNK
        if (lane == 0) flags[rslot] = 1;
      } else {
        // plain wave reduction, then one atomic per component from lane 63
        const float ws = wave_reduce_to_lane63(gg.s), wx = wave_reduce_to_lane63(gg.s_x);
        const float wy = wave_reduce_to_lane63(gg.s_y), wxx = wave_reduce_to_lane63(gg.s_xx);
        const float wxy = wave_reduce_to_lane63(gg.s_xy), wyy = wave_reduce_to_lane63(gg.s_yy);
        float rx, ry, ra, rb, rc;
        moments_to_mean(g2.y, g2.z, ws, wx, wy, wxx, wxy, wyy, rx, ry, ra, rb, rc);
        const float ro = -ws * __builtin_amdgcn_exp2f(-g2.w);     // opacity * d/d opacity = -sum v_sigma; opacity = 2^L
        float rf[CHT];
#pragma unroll
        for (int c = 0; c < CHT; ++c) rf[c] = wave_reduce_to_lane63(gg.v_f[c]);
        float ax = 0.f, ay = 0.f;
        if (ABSGRAD) { ax = wave_reduce_to_lane63(gg.a_x); ay = wave_reduce_to_lane63(gg.a_y); }
        if (lane == 63) {
          finish_geo(g3.x, g3.y, g3.z, rx, ry, ra, rc);
          unsafeAtomicAdd(&v_means2d[2 * (size_t)gid + 0], rx);
          unsafeAtomicAdd(&v_means2d[2 * (size_t)gid + 1], ry);
          unsafeAtomicAdd(&v_conics[3 * (size_t)gid + 0], ra);
          unsafeAtomicAdd(&v_conics[3 * (size_t)gid + 1], rb);
          unsafeAtomicAdd(&v_conics[3 * (size_t)gid + 2], rc);
          unsafeAtomicAdd(&v_opacities[gid], ro);
#pragma unroll
          for (int c = 0; c < CHT; ++c)
            if (c < channels) unsafeAtomicAdd(&v_feats[(size_t)gid * channels + c], rf[c]);
          if (ABSGRAD) {
            unsafeAtomicAdd(&v_means2d_abs[2 * (size_t)gid + 0], ax);
            unsafeAtomicAdd(&v_means2d_abs[2 * (size_t)gid + 1], ay);
          }
        }
      }
    }
    };
#ifdef MGS_RASTER_BWD_NO_SAFE     // measurement: every batch through the general form
    walk(std::false_type{});
#else
    if (all_safe) walk(std::true_type{}); else walk(std::false_type{});
#endif
    __builtin_amdgcn_wave_barrier();
  }
  if (PIPE && pend) {
    float4 a0, b0, a1, b1;
    red_load(a0, b0, a1, b1);
    red_finish(pend_slot, a0, b0, a1, b1);
  }
#ifdef MGS_RASTER_BWD_TIMING
  if (lane == 0 && unit < 65536) {
    g_bwd_times[5 * unit + 0] = t_enter;
    g_bwd_times[5 * unit + 1] = t_begin;
    g_bwd_times[5 * unit + 2] = __builtin_amdgcn_s_memrealtime();
    g_bwd_times[5 * unit + 3] = n_walked;
    g_bwd_times[5 * unit + 4] = n_pairs;
  }
#endif
}

template <int CHT, bool ABSGRAD, bool RECORDS, bool HALF = false, bool SPLIT = false>
__global__ __launch_bounds__(64 * MGS_RASTER_BWD_WG_WAVES, (CHT <= 4 && RECORDS && !ABSGRAD && !HALF) ? MGS_RASTER_BWD_MIN_WAVES : 1) void raster_bwd_kernel(
    const float* __restrict__ means2d, const float* __restrict__ conics,
    const float* __restrict__ feats, const float* __restrict__ opacities,
    const float4* __restrict__ splats, const float* __restrict__ background, int channels,
    int width, int height, int tile_w, int n_tiles, const int32_t* __restrict__ tile_offsets,
    const int32_t* __restrict__ flatten_ids, const float* __restrict__ alphas,
    const int32_t* __restrict__ last_ids, const float* __restrict__ v_render,
    const float* __restrict__ v_alphas, float* __restrict__ v_means2d,
    float* __restrict__ v_means2d_abs, float* __restrict__ v_conics,
    float* __restrict__ v_feats, float* __restrict__ v_opacities,
    const int4* __restrict__ pair_info, float* __restrict__ records,
    uint8_t* __restrict__ flags, uint32_t capacity, const float* __restrict__ expected_render,
    const int32_t* __restrict__ tile_order, const float* __restrict__ ckpt, int ckpt_shift,
    const int32_t* __restrict__ seg_table, const float* __restrict__ render_out, int splat_slots) {
#define MGS_RB_ARGS means2d, conics, feats, opacities, splats, background, channels, width, height, tile_w, n_tiles, \
    tile_offsets, flatten_ids, alphas, last_ids, v_render, v_alphas, v_means2d, v_means2d_abs, v_conics, v_feats,    \
    v_opacities, pair_info, records, flags, capacity, expected_render, tile_order, ckpt, ckpt_shift, seg_table, render_out, splat_slots
  int unit = blockIdx.x * MGS_RASTER_BWD_WG_WAVES + (int)(threadIdx.x >> 6);
  if constexpr (SPLIT && MGS_RASTER_BWD_XCD_RUN > 1 && MGS_RASTER_BWD_WG_WAVES == 1) {
    // a tile's whole segments are neighbours in the unit table and re-read the same 10 KB of frame state; workgroup b
    // runs on XCD b % 8 (observed placement, speed only), so runs of R consecutive units go to one XCD's L2:
    // blocks of 8 R workgroups, unit = block + (b % 8) R + (b / 8) % R -- the launch order is kept to within a block
    constexpr int R = MGS_RASTER_BWD_XCD_RUN;
    const int b = unit, in = b % (8 * R);
    unit = b - in + (in & 7) * R + (in >> 3);
  }
  raster_bwd_unit<CHT, ABSGRAD, RECORDS, HALF, SPLIT>(unit, MGS_RB_ARGS);
#undef MGS_RB_ARGS
}

// ---- rectangles of kBigPairs tiles and more (round 6) ---------------------------------------------------------------------
// reduce_records_rows_kernel hands a wave 64 Gaussians and deals their rectangles' rows to its lanes: fine for rectangles
// of a dozen tiles, but a needle's bounding box of 50 x 50 tiles or a screen-filling Gaussian's 120 x 68 are thousands of
// slots, and where an export keeps such Gaussians together in the index range a few hundred waves own all of them (a
// clustered scene: 410 us as given, 185 us in Morton order, against 49 us for the uniform scene).  Such a Gaussian is
// ONE WAVE's job instead, wherever it sits: a launch that runs before the raster lists them (big_discover: extra
// workgroups of unit_table_kernel, kBigLists lists so that no counter is hot), the reduce launch starts with kBigWaves
// waves that take them one at a time -- lanes = rows, the rows then added in row order by NV lanes -- and its other waves
// skip them.  The order of the additions is the one of the rows kernel (within a row by column, then the rows): the same
// bits.
#ifndef MGS_REDUCE_BIG
#define MGS_REDUCE_BIG 256
#endif
#ifndef MGS_REDUCE_BIG_WGS
#define MGS_REDUCE_BIG_WGS 1024
#endif
constexpr int kBigPairs = MGS_REDUCE_BIG, kBigLists = 64, kBigWGs = MGS_REDUCE_BIG_WGS, kDiscoverPerThread = 16;
__host__ __device__ inline uint32_t big_list_room(uint32_t capacity) { return capacity / (uint32_t)kBigPairs + 1u; }
__host__ __device__ inline size_t big_lists_bytes(uint32_t capacity) { return (size_t)kBigLists * big_list_room(capacity) * sizeof(uint32_t); }

// counts: kBigLists words (zeroed by the caller's memset); items: kBigLists lists of big_list_room(capacity) Gaussian ids
__device__ __forceinline__ void big_discover(int wg, int n, const int4* __restrict__ pair_info, uint32_t capacity,
                                             uint32_t* __restrict__ counts, uint32_t* __restrict__ items) {
  const int t = threadIdx.x;
  const uint32_t list = (uint32_t)(wg * 4 + (t >> 6)) % kBigLists, room = big_list_room(capacity);
  int4 info[kDiscoverPerThread];
#pragma unroll
  for (int j = 0; j < kDiscoverPerThread; ++j) {
    const int g = (wg * kDiscoverPerThread + j) * 256 + t;
    info[j] = g < n ? pair_info[g] : make_int4(0, 0, 0, 0);
  }
  // one returning atomic per wave, whatever it found (sixteen dependent ones took the kernel from 6 to 16 us)
  unsigned long long mm[kDiscoverPerThread];
  uint32_t run[kDiscoverPerThread], wave_total = 0;
#pragma unroll
  for (int j = 0; j < kDiscoverPerThread; ++j) {
    mm[j] = ballot((info[j].w & 0xffff) * (int)((unsigned)info[j].w >> 16) >= kBigPairs);
    run[j] = wave_total;
    wave_total += (uint32_t)__popcll(mm[j]);
  }
  if (wave_total == 0u) return;                         // (uniform; nearly every wave)
  uint32_t base = 0;
  if ((t & 63) == 0) base = atomicAdd(&counts[list], wave_total);
  base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
  for (int j = 0; j < kDiscoverPerThread; ++j) {
    const uint32_t at = base + run[j] + mask_rank(mm[j]);
    if (((mm[j] >> (t & 63)) & 1ull) && at < room) items[(size_t)list * room + at] = (uint32_t)((wg * kDiscoverPerThread + j) * 256 + t);
  }
}

// SPLIT: the units of the segmented launch, one thread per tile: the segments some pixel of the tile reaches (the forward
// left the end of the tile's walk in the checkpoint header).  WHOLE segments go to one table (a cursor per workgroup
// scan), every tile's last, partly walked segment to the table of its length class (32 classes, longest first); the
// raster's unit u is whole[u] or, past those, the (u - n_whole)-th partial one counted through the classes.  All
// whole segments cost about the same, so with the short ones at the end of the launch the chip drains in the time of
// a short unit instead of a whole one (a unit is one wave's serial job; scripts/dbg/bwd_timeline.py).
// counts[0] = whole segments, counts[1 + c] = partial segments of class c: zeroed by the caller's memset.
// Workgroups past the tiles' (n_big_wgs of them): big_discover.
__global__ __launch_bounds__(256) void unit_table_kernel(int n_tiles, const int32_t* __restrict__ tile_offsets,
                                                         const int32_t* __restrict__ tile_hi, int shift, uint32_t capacity,
                                                         int32_t* __restrict__ tables, int n, const int4* __restrict__ pair_info,
                                                         uint32_t* __restrict__ big_counts, uint32_t* __restrict__ big_items) {
  __shared__ uint32_t wave_tot[4], base, cls_n[kUnitClasses], cls_at[kUnitClasses];
  const int n_tile_wgs = (n_tiles + 255) / 256;
  if ((int)blockIdx.x >= n_tile_wgs) {
    big_discover((int)blockIdx.x - n_tile_wgs, n, pair_info, capacity, big_counts, big_items);
    return;
  }
  const UnitTables ut = unit_tables(tables, n_tiles, shift, capacity);
  const int tile = blockIdx.x * 256 + threadIdx.x, t = threadIdx.x;
  if (t < kUnitClasses) cls_n[t] = 0u;
  __syncthreads();
  int n_seg = 0, last_len = 0, u_start = 0, u_hi = 0;
  if (tile < n_tiles) {
    const int start = tile_offsets[tile], end = tile_offsets[tile + 1];
    const int4 h4 = reinterpret_cast<const int4*>(tile_hi)[tile];
    const int hi = min(max(max(h4.x, h4.y), max(h4.z, h4.w)), end - 1);
    u_start = start;
    u_hi = hi;
    if (hi >= start) {
      n_seg = ((hi - start) >> shift) + 1;
      last_len = ((hi - start) & ((1 << shift) - 1)) + 1;
    }
  }
  const uint32_t mine = n_seg > 0 ? (uint32_t)(n_seg - 1) : 0u;
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d);
    if ((t & 63) >= d) incl += up;
  }
  if ((t & 63) == 63) wave_tot[t >> 6] = incl;
  __syncthreads();
  uint32_t at = incl - mine;
  for (int w = 0; w < (t >> 6); ++w) at += wave_tot[w];
  // the workgroup counts its partial segments per class in LDS and takes ONE place per class from the global counters
  // (a returning atomic per tile on 32 addresses serialised: the kernel took 20 us, profiles/r4/03)
  const int c = n_seg > 0 ? ((1 << shift) - last_len) * kUnitClasses >> shift : 0;          // 0 = the longest
  uint32_t rank = 0;
  if (n_seg > 0) rank = atomicAdd(&cls_n[c], 1u);
  if (t == 255) base = atomicAdd(&ut.counts[0], at + mine);
  __syncthreads();
  if (t < kUnitClasses && cls_n[t]) cls_at[t] = atomicAdd(&ut.counts[1 + t], cls_n[t]);
  __syncthreads();
  at += base;
  for (int sg = 0; sg + 1 < n_seg; ++sg) ut.whole[at + sg] = make_int4(tile, sg, u_start, u_hi);
  if (n_seg > 0) ut.part[(size_t)c * n_tiles + cls_at[c] + rank] = make_int4(tile, n_seg - 1, u_start, u_hi);
}
// (the whole-list walk has no unit tables: the discovery alone)
__global__ __launch_bounds__(256) void big_discover_kernel(int n, const int4* __restrict__ pair_info, uint32_t capacity,
                                                           uint32_t* __restrict__ big_counts, uint32_t* __restrict__ big_items) {
  big_discover((int)blockIdx.x, n, pair_info, capacity, big_counts, big_items);
}

// Sum the records of each Gaussian's slots (its tile rectangle, emit order) into the outputs.  A record holds the
// pair's moments of v_sigma about ITS tile's centre (GaussGrad); the slot's position in the rectangle gives the tile,
// hence m = mean - tile centre (the very expression the raster kernel queued), moments_to_mean the pair's sums about the
// mean; those add up over the pairs, then the conic is applied once (finish_geo) and -sum s / opacity is the opacity's
// gradient.
#ifndef MGS_REDUCE_EXP
#define MGS_REDUCE_EXP 0
#endif
template <int CHT, bool ABSGRAD, int SLOTS = 1>      // SLOTS: record slots per (tile, Gaussian) pair (2: half tiles)
__global__ __launch_bounds__(256) void reduce_records_kernel(
    int n, const int4* __restrict__ pair_info, const float* __restrict__ records,
    const uint8_t* __restrict__ flags, uint32_t capacity, const float* __restrict__ means2d,
    const float* __restrict__ conics, const float* __restrict__ opacities,
    const float4* __restrict__ splats, int channels, float* __restrict__ v_means2d,
    float* __restrict__ v_means2d_abs, float* __restrict__ v_conics,
    float* __restrict__ v_feats, float* __restrict__ v_opacities) {
  int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= n) return;
  const int4 info = pair_info[g];
  const int rect_w = info.w & 0xffff;
  const int npair = rect_w * ((unsigned)info.w >> 16);
  const int cnt = npair * SLOTS;
  const size_t first = (size_t)info.x * SLOTS;
  constexpr int RSP = record_floats(CHT, ABSGRAD), R4 = RSP / 4;
  const float4* rec4 = reinterpret_cast<const float4*>(records);
  float acc[6], af[CHT], ab[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[i] = 0.f;
#pragma unroll
  for (int c = 0; c < CHT; ++c) af[c] = 0.f;
  float mean_x = 0.f, mean_y = 0.f, ca = 1.f, cb = 0.f, cc = 1.f, op = 1.f;
  if (cnt > 0) {
    if (splats) {
      const float4 p0 = splats[3 * (size_t)g], p1 = splats[3 * (size_t)g + 1];
      mean_x = p0.x; mean_y = p0.y; ca = p0.z; cb = p0.w; cc = p1.x; op = p1.y;
    } else {
      mean_x = means2d[2 * (size_t)g]; mean_y = means2d[2 * (size_t)g + 1];
      ca = conics[3 * (size_t)g]; cb = conics[3 * (size_t)g + 1]; cc = conics[3 * (size_t)g + 2];
      op = opacities[g];
    }
  }
  // (loading the records unconditionally and selecting by the flag afterwards -- one round trip instead
  //  of two -- was measured: 618 -> 676 us for the whole backward; the extra 64 MB cost more)
  // four slots per trip with predicated loads: the flag and record loads of a trip are all in
  // flight together instead of one dependent round trip per slot (summation order is unchanged)
  // The four flags of a trip are ONE aligned 4-byte load: trips cover the aligned groups of four slots that overlap
  // the Gaussian's range (a group's slots outside the range are skipped), not four byte loads whose 64 lanes each
  // touch a different place -- the kernel is bound by the number of such scattered loads (round-3 ablation: 34 of its
  // 65 us remained with the record loads removed).
  int col = 0, row = 0;                // the pair's tile inside the rectangle (slots are row-major, SLOTS per pair)
  const int lead = (int)(first & 3);   // slots of the first group that belong to the previous Gaussian
  for (int sl = -lead; sl < cnt; sl += 4) {
    bool on[4];
#if MGS_REDUCE_EXP == 2      // measurement: no flag loads (every slot read)
    const uint32_t fw = 0x01010101u;
#else
    // (first + sl) is a multiple of 4; groups past the capacity (overflowed lists) lie outside the workspace
    const uint32_t fw = first + sl < (size_t)capacity * SLOTS ? *reinterpret_cast<const uint32_t*>(flags + (first + sl)) : 0u;
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i)      // slots at or past the capacity do not exist (overflowed lists)
      on[i] = sl + i >= 0 && sl + i < cnt && (uint32_t)(info.x + (sl + i) / SLOTS) < capacity && ((fw >> (8 * i)) & 0xffu) != 0;
    float r[4][RSP];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t slot = first + sl + i;
#pragma unroll
      for (int k = 0; k < R4; ++k) {
#if MGS_REDUCE_EXP == 1      // measurement: no record loads
        const float4 v = make_float4(on[i] ? 1.f : 0.f, 0.f, 0.f, 0.f);
#else
        const float4 v = on[i] ? rec4[slot * R4 + k] : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
        r[i][4 * k] = v.x; r[i][4 * k + 1] = v.y; r[i][4 * k + 2] = v.z; r[i][4 * k + 3] = v.w;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // m exactly as the raster kernel formed it: mean - (16 tile + 8)
      const float mx = mean_x - ((float)((info.y + col) * 16) + 8.f), my = mean_y - ((float)((info.z + row) * 16) + 8.f);
      float P, Q, Vaa, Vab, Vbb;
      moments_to_mean(mx, my, r[i][0], r[i][1], r[i][2], r[i][3], r[i][4], r[i][5], P, Q, Vaa, Vab, Vbb);
      acc[0] += P; acc[1] += Q; acc[2] += Vaa; acc[3] += Vab; acc[4] += Vbb; acc[5] += r[i][0];
#pragma unroll
      for (int c = 0; c < CHT; ++c) af[c] += r[i][6 + c];
      if constexpr (ABSGRAD) { ab[0] += r[i][6 + CHT]; ab[1] += r[i][7 + CHT]; }
      if (sl + i >= 0 && (SLOTS == 1 || ((sl + i) % SLOTS) == SLOTS - 1))
        if (++col == rect_w) { col = 0; ++row; }
    }
  }
  if (cnt > 0) {   // apply the Gaussian's conic once, here; opacity * d/d opacity = -sum v_sigma
    finish_geo(ca, cb, cc, acc[0], acc[1], acc[2], acc[4]);
    acc[5] = op > 0.f ? -acc[5] / op : 0.f;     // (an opacity of exactly 0 owns slots under classic tile bounds: s is 0 there, not 0 / 0)
  }
#if MGS_REDUCE_EXP == 3      // measurement: (nearly) no stores
  if (acc[0] != 12345.f) return;
#endif
  reinterpret_cast<float2*>(v_means2d)[g] = make_float2(acc[0], acc[1]);
  v_conics[3 * (size_t)g + 0] = acc[2];
  v_conics[3 * (size_t)g + 1] = acc[3];
  v_conics[3 * (size_t)g + 2] = acc[4];
  v_opacities[g] = acc[5];
#pragma unroll
  for (int c = 0; c < CHT; ++c)
    if (c < channels) v_feats[(size_t)g * channels + c] = af[c];
  if (ABSGRAD) reinterpret_cast<float2*>(v_means2d_abs)[g] = make_float2(ab[0], ab[1]);
}


#ifndef MGS_REDUCE_TRIP
#define MGS_REDUCE_TRIP 4       // slots of a row fetched together (reduce_records_rows_kernel)
#endif
#ifndef MGS_REDUCE_RUN
#define MGS_REDUCE_RUN 16       // reduce_records_rows_kernel: consecutive Gaussians per run of a wave (four runs; 0: 64 consecutive)
#endif
#ifndef MGS_REDUCE_ROWS
// 1: reduce_records_rows_kernel (below) for up to 4 channels and one slot per pair; 0: reduce_records_kernel everywhere
#define MGS_REDUCE_ROWS 1
#endif
// The same sums with the ROWS of the tile rectangles as the units of work.  reduce_records_kernel gives every lane one
// Gaussian and lets it walk all its slots: a wave takes as long as its largest rectangle (25 trips of four slots where
// the average lane needs 1-2), one dependent round trip after another -- 14 SIMD-cycles per vector instruction, pure
// latency.  Here a wave still owns 64 consecutive Gaussians, but their rectangles' rows (2-3 slots each) are dealt out
// to the lanes, 64 rows per round: every lane does one short trip per round, and the Gaussians then add up their own
// rows of the round out of LDS, in row order.  The order of additions is fixed by the tiles' coordinates alone -- within
// a row by column, then the rows by row -- so the result is bit-reproducible and does not depend on which rectangle
// (classic or tightened: the extra slots are unflagged, the extra rows add exact zeros) lists the tiles.
template <int CHT, bool ABSGRAD>
__global__ __launch_bounds__(256) void reduce_records_rows_kernel(
    int n, const int4* __restrict__ pair_info, const float* __restrict__ records,
    const uint8_t* __restrict__ flags, uint32_t capacity, const float* __restrict__ means2d,
    const float* __restrict__ conics, const float* __restrict__ opacities,
    const float4* __restrict__ splats, int channels, float* __restrict__ v_means2d,
    float* __restrict__ v_means2d_abs, float* __restrict__ v_conics,
    float* __restrict__ v_feats, float* __restrict__ v_opacities,
    const uint32_t* __restrict__ big_counts, const uint32_t* __restrict__ big_items) {
  constexpr int RSP = record_floats(CHT, ABSGRAD), R4 = RSP / 4, NV = 6 + CHT + (ABSGRAD ? 2 : 0);
  constexpr int PITCH = NV | 1;                       // odd pitch: the lanes' row sums fall into different banks
  __shared__ int s_row0[4][64];                       // first row (task) of each Gaussian of the wave
  __shared__ int4 s_info[4][64];
  __shared__ float2 s_mean[4][64];
  __shared__ float s_rows[4][64 * PITCH];
  const int wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
  const float4* rec4 = reinterpret_cast<const float4*>(records);
  // one row of a rectangle: its slots' moments about the mean and colour gradients, added column by column
  auto row_sum = [&](const int4 oi, const float2 om, const int r, float (&rs)[NV]) {
    const int ow = oi.w & 0xffff;
    const uint32_t slot0 = (uint32_t)oi.x + (uint32_t)(r * ow);
    const float my = om.y - ((float)((oi.z + r) * 16) + 8.f);     // m exactly as the raster kernel formed it
    constexpr int TR = MGS_REDUCE_TRIP;
    // the flags of a trip out of the two aligned words that hold them (slots at or past the capacity -- overflowed lists --
    // do not exist and lie outside the workspace); fetched one trip AHEAD: a trip is then one round trip (its records)
    // instead of two dependent ones -- a 120-tile row of a screen-filling Gaussian is 30 trips
    auto trip_flags = [&](const uint32_t s0) -> uint32_t {
      const uint32_t a0 = s0 & ~3u, sh = (s0 & 3u) * 8u;
      const uint32_t w0 = a0 < capacity ? *reinterpret_cast<const uint32_t*>(flags + a0) : 0u;
      const uint32_t w1 = (sh && a0 + 4u < capacity) ? *reinterpret_cast<const uint32_t*>(flags + a0 + 4u) : 0u;
      return sh ? (w0 >> sh) | (w1 << (32u - sh)) : w0;
    };
    uint32_t fw_next = trip_flags(slot0);
    for (int c0 = 0; c0 < ow; c0 += TR) {
      const uint32_t s0 = slot0 + (uint32_t)c0;
      const uint32_t fw = fw_next;
      if (c0 + TR < ow) fw_next = trip_flags(s0 + (uint32_t)TR);
      bool on[TR];
      float rr[TR][RSP];
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        on[i] = c0 + i < ow && s0 + (uint32_t)i < capacity && ((fw >> (8 * i)) & 0xffu) != 0;
#pragma unroll
        for (int k = 0; k < R4; ++k) {
          const float4 v = on[i] ? rec4[(size_t)(s0 + (uint32_t)i) * R4 + k] : make_float4(0.f, 0.f, 0.f, 0.f);
          rr[i][4 * k] = v.x; rr[i][4 * k + 1] = v.y; rr[i][4 * k + 2] = v.z; rr[i][4 * k + 3] = v.w;
        }
      }
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        const float mx = om.x - ((float)((oi.y + c0 + i) * 16) + 8.f);
        float P, Q, Vaa, Vab, Vbb;
        moments_to_mean(mx, my, rr[i][0], rr[i][1], rr[i][2], rr[i][3], rr[i][4], rr[i][5], P, Q, Vaa, Vab, Vbb);
        rs[0] += P; rs[1] += Q; rs[2] += Vaa; rs[3] += Vab; rs[4] += Vbb; rs[5] += rr[i][0];
#pragma unroll
        for (int c = 0; c < CHT; ++c) rs[6 + c] += rr[i][6 + c];
        if constexpr (ABSGRAD) { rs[6 + CHT] += rr[i][6 + CHT]; rs[7 + CHT] += rr[i][7 + CHT]; }
      }
    }
  };
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  if ((int)blockIdx.x < kBigWGs) {
    // ---- the big rectangles: one wave each ------------------------------------------------------------------------------
    const uint32_t room = big_list_room(capacity);
    uint32_t cnt = lane < kBigLists ? big_counts[lane] : 0u;      // (kBigLists == 64: one list per lane)
    cnt = cnt < room ? cnt : room;
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    const uint32_t total = __shfl(incl, 63);
    for (uint32_t k = (uint32_t)blockIdx.x * 4u + (uint32_t)wv; k < total; k += (uint32_t)kBigWGs * 4u) {
      const int l = __ffsll((long long)ballot(incl > k)) - 1;      // the list that holds item k
      const uint32_t first = __shfl(incl - cnt, l);
      const int g = (int)big_items[(size_t)l * room + (k - first)];
      const int4 oi = pair_info[g];
      const int h = (int)((unsigned)oi.w >> 16);
      float2 om;
      float ca, cb, cc, op;
      if (splats) {
        const float4 p0 = splats[3 * (size_t)g], p1 = splats[3 * (size_t)g + 1];
        om = make_float2(p0.x, p0.y); ca = p0.z; cb = p0.w; cc = p1.x; op = p1.y;
      } else {
        om = make_float2(means2d[2 * (size_t)g], means2d[2 * (size_t)g + 1]);
        ca = conics[3 * (size_t)g]; cb = conics[3 * (size_t)g + 1]; cc = conics[3 * (size_t)g + 2];
        op = opacities[g];
      }
      float acc = 0.f;                                              // lane i < NV: component i, the rows in row order
      for (int base = 0; base < h; base += 64) {
        float rs[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) rs[i] = 0.f;
        if (base + lane < h) row_sum(oi, om, base + lane, rs);
#pragma unroll
        for (int i = 0; i < NV; ++i) s_rows[wv][lane * PITCH + i] = rs[i];
        wave_sync();
        const int rows = min(64, h - base);
        if (lane < NV)
          for (int t2 = 0; t2 < rows; ++t2) acc += s_rows[wv][t2 * PITCH + lane];
        __builtin_amdgcn_wave_barrier();                            // the next round overwrites s_rows
      }
      float a[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) a[i] = __shfl(acc, i);
      finish_geo(ca, cb, cc, a[0], a[1], a[2], a[4]);
      a[5] = op > 0.f ? -a[5] / op : 0.f;
      if (lane == 0) {
        reinterpret_cast<float2*>(v_means2d)[g] = make_float2(a[0], a[1]);
        v_conics[3 * (size_t)g + 0] = a[2];
        v_conics[3 * (size_t)g + 1] = a[3];
        v_conics[3 * (size_t)g + 2] = a[4];
        v_opacities[g] = a[5];
#pragma unroll
        for (int c = 0; c < CHT; ++c)
          if (c < channels) v_feats[(size_t)g * channels + c] = a[6 + c];
        if constexpr (ABSGRAD) reinterpret_cast<float2*>(v_means2d_abs)[g] = make_float2(a[6 + CHT], a[7 + CHT]);
      }
    }
    return;
  }
  const int block = (int)blockIdx.x - kBigWGs, n_blocks = (int)gridDim.x - kBigWGs;
  // Which 64 Gaussians a wave owns: four RUNS of MGS_REDUCE_RUN consecutive ones, a whole launch of waves apart (run r
  // of wave W is run r * n_waves + W of the index space) instead of 64 consecutive.  Large rectangles that sit together
  // in the index range -- appended by a densifier, neighbours in a Morton order -- then land in many waves, each with a
  // few, instead of a few waves with 64 each (a wave's time is its row tasks; profiles/r5/00_experiments.md 18, 22).
  // A run still reads whole lines of pair_info and writes whole sectors of the outputs.  Training step, ms (runs of
  // 64 = consecutive / 32 / 16 / 8): configs[2] as given 0.823 / 0.823 / 0.824 / 0.827, in Morton order 0.786 / 0.780 /
  // 0.779 / 0.777; a clustered scene as given 1.526 / 1.460 / 1.397 / 1.376, in Morton order 1.082 / 1.056 / 1.002 / 0.998.
#if MGS_REDUCE_RUN > 0
  constexpr int kRun = MGS_REDUCE_RUN, kRunsPerWave = 64 / kRun;
  const int wave_global = block * 4 + wv, n_waves = n_blocks * 4;
  const long long g_ll = ((long long)(lane / kRun) * n_waves + wave_global) * kRun + lane % kRun;
  const int g = g_ll < n ? (int)g_ll : n;
  static_assert(64 % kRun == 0 && kRunsPerWave >= 1, "runs of a power of two up to 64");
#else
  const int g = block * 256 + (int)threadIdx.x;
#endif
  int4 info = make_int4(0, 0, 0, 0);
  int h = 0;
  bool is_big = false;
  float mean_x = 0.f, mean_y = 0.f, ca = 1.f, cb = 0.f, cc = 1.f, op = 1.f;
  if (g < n) {
    info = pair_info[g];
    h = (int)((unsigned)info.w >> 16);
    if ((info.w & 0xffff) == 0) h = 0;
    is_big = (info.w & 0xffff) * h >= kBigPairs;        // (a wave of the launch's first workgroups takes it)
    if (is_big) h = 0;
    if (h > 0) {
      if (splats) {
        const float4 p0 = splats[3 * (size_t)g], p1 = splats[3 * (size_t)g + 1];
        mean_x = p0.x; mean_y = p0.y; ca = p0.z; cb = p0.w; cc = p1.x; op = p1.y;
      } else {
        mean_x = means2d[2 * (size_t)g]; mean_y = means2d[2 * (size_t)g + 1];
        ca = conics[3 * (size_t)g]; cb = conics[3 * (size_t)g + 1]; cc = conics[3 * (size_t)g + 2];
        op = opacities[g];
      }
    }
  }
  int incl = h;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
  const int row0 = incl - h, n_rows = __shfl(incl, 63);
  s_row0[wv][lane] = row0;
  s_info[wv][lane] = info;
  s_mean[wv][lane] = make_float2(mean_x, mean_y);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.f;
  for (int base = 0; base < n_rows; base += 64) {
    const int t = base + lane;
    float rs[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) rs[i] = 0.f;
    if (t < n_rows) {
      int o = 0;                                      // the last Gaussian whose first row is <= t: the row's owner
#pragma unroll
      for (int step = 32; step >= 1; step >>= 1)
        if (s_row0[wv][o + step] <= t) o += step;     // (o + step <= 63)
      row_sum(s_info[wv][o], s_mean[wv][o], t - s_row0[wv][o], rs);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) s_rows[wv][lane * PITCH + i] = rs[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // every Gaussian adds its rows of this round, in row order
    const int a = max(row0, base), b = min(row0 + h, base + 64);
    for (int t2 = a; t2 < b; ++t2) {
#pragma unroll
      for (int i = 0; i < NV; ++i) acc[i] += s_rows[wv][(t2 - base) * PITCH + i];
    }
    __builtin_amdgcn_wave_barrier();                  // the next round overwrites s_rows
  }
  if (g >= n || is_big) return;
  if (h > 0) {      // apply the Gaussian's conic once, here; opacity * d/d opacity = -sum v_sigma
    finish_geo(ca, cb, cc, acc[0], acc[1], acc[2], acc[4]);
    acc[5] = op > 0.f ? -acc[5] / op : 0.f;
  }
  reinterpret_cast<float2*>(v_means2d)[g] = make_float2(acc[0], acc[1]);
  v_conics[3 * (size_t)g + 0] = acc[2];
  v_conics[3 * (size_t)g + 1] = acc[3];
  v_conics[3 * (size_t)g + 2] = acc[4];
  v_opacities[g] = acc[5];
#pragma unroll
  for (int c = 0; c < CHT; ++c)
    if (c < channels) v_feats[(size_t)g * channels + c] = acc[6 + c];
  if constexpr (ABSGRAD) reinterpret_cast<float2*>(v_means2d_abs)[g] = make_float2(acc[6 + CHT], acc[7 + CHT]);
}

}  // namespace
}  // namespace mgs

using namespace mgs;

#ifdef MGS_RASTER_BWD_TIMING
extern "C" int mgs_debug_bwd_times(unsigned long long* host, int n_words) {      // copies the stamps out and clears them
  hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bwd_times), (size_t)n_words * 8);
  if (e == hipSuccess) {
    void* dev = nullptr;
    e = hipGetSymbolAddress(&dev, HIP_SYMBOL(g_bwd_times));
    if (e == hipSuccess) e = hipMemset(dev, 0, sizeof(g_bwd_times));
  }
  return e == hipSuccess ? 0 : (int)e;
}
#endif
extern "C" int mgs_rasterize_bwd(int n, const float* means2d, const float* conics,
                                 const float* feats, const float* opacities,
                                 const float* background, int channels, int width, int height,
                                 int tile_w, int tile_h, const int32_t* tile_offsets,
                                 const int32_t* flatten_ids, const float* alphas,
                                 const int32_t* last_ids, const float* v_render,
                                 const float* v_alphas, float* v_means2d, float* v_means2d_abs,
                                 float* v_conics, float* v_feats, float* v_opacities,
                                 mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && width > 0 && height > 0, "rasterize_bwd: bad sizes");
  MGS_REQUIRE(channels >= 1 && channels <= MGS_MAX_CHANNELS, "rasterize_bwd: channels %d outside 1..%d", channels, MGS_MAX_CHANNELS);
  MGS_REQUIRE(tile_w == (width + 15) / 16 && tile_h == (height + 15) / 16,
              "rasterize_bwd: tile grid does not match the image at tile size 16");
  if (n == 0) return MGS_OK;
  MGS_REQUIRE(means2d && conics && feats && opacities && tile_offsets && flatten_ids && alphas &&
                  last_ids && v_render && v_alphas && v_means2d && v_conics && v_feats &&
                  v_opacities, "rasterize_bwd: null pointer");
  const int n_tiles = tile_w * tile_h;
  hipStream_t s = (hipStream_t)stream;
#define MGS_RB_LAUNCH(C, A)                                                                    \
  hipLaunchKernelGGL((raster_bwd_kernel<C, A, false>), dim3(div_up(n_tiles, MGS_RASTER_BWD_WG_WAVES)), dim3(64 * MGS_RASTER_BWD_WG_WAVES), 0, s, means2d,  \
                     conics, feats, opacities, (const float4*)nullptr, background, channels,   \
                     width, height, tile_w,                                                    \
                     n_tiles, tile_offsets, flatten_ids, alphas, last_ids, v_render, v_alphas, \
                     v_means2d, v_means2d_abs, v_conics, v_feats, v_opacities,                 \
                     (const int4*)nullptr, (float*)nullptr, (uint8_t*)nullptr, 0u, (const float*)nullptr, \
                     (const int32_t*)nullptr, (const float*)nullptr, 0, (const int32_t*)nullptr, (const float*)nullptr, 0)
#define MGS_RB(C) if (v_means2d_abs) MGS_RB_LAUNCH(C, true); else MGS_RB_LAUNCH(C, false)
  if (channels == 1) { MGS_RB(1); }
  else if (channels == 2) { MGS_RB(2); }
  else if (channels == 3) { MGS_RB(3); }
  else if (channels == 4) { MGS_RB(4); }
  else if (channels <= 8) { MGS_RB(8); }
  else if (channels <= 16) { MGS_RB(16); }
  else { MGS_RB(32); }
#undef MGS_RB
#undef MGS_RB_LAUNCH
  return check_launch("rasterize_bwd");
}

extern "C" int mgs_rasterize_bwd_det(int n, const float* means2d, const float* conics,
                                     const float* feats, const float* opacities,
                                     const float* splats, const float* background, int channels,
                                     int width,
                                     int height, int tile_w, int tile_h,
                                     const int32_t* tile_offsets, const int32_t* flatten_ids,
                                     const float* alphas, const int32_t* last_ids,
                                     const float* v_render, const float* v_alphas,
                                     const float* expected_render,
                                     const int32_t* pair_info, const int32_t* tile_group_order,
                                     uint32_t isect_capacity, const float* render_out,
                                     const float* checkpoints, int checkpoint_interval, int call_flags,
                                     float* v_means2d, float* v_means2d_abs, float* v_conics,
                                     float* v_feats, float* v_opacities, void* workspace,
                                     size_t* workspace_bytes, mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && width > 0 && height > 0, "rasterize_bwd_det: bad sizes");
  MGS_REQUIRE(channels >= 1 && channels <= MGS_MAX_CHANNELS, "rasterize_bwd_det: channels %d outside 1..%d", channels, MGS_MAX_CHANNELS);
  MGS_REQUIRE(tile_w == (width + 15) / 16 && tile_h == (height + 15) / 16,
              "rasterize_bwd_det: tile grid does not match the image at tile size 16");
  MGS_REQUIRE(workspace_bytes, "rasterize_bwd_det: workspace_bytes is null");
  const int rs = record_floats(padded_channels(channels), v_means2d_abs != nullptr);   // floats per record
  const size_t cap = isect_capacity ? isect_capacity : 1;
  constexpr bool kHalf = MGS_RASTER_BWD_HALF != 0;
  constexpr size_t kSlots = kHalf ? 2 : 1;
  const size_t rec_bytes = align_up(cap * kSlots * rs * sizeof(float), 256);
  const size_t flag_bytes = align_up(cap * kSlots, 256);
  // segmented walk (checkpoints from mgs_rasterize_fwd): up to 4 channels, whole tiles
  int ckpt_shift = 0;
  const bool split = checkpoints != nullptr && channels <= 4 && !kHalf;
  if (checkpoints) {
    MGS_REQUIRE(checkpoint_interval >= 64 && (checkpoint_interval & (checkpoint_interval - 1)) == 0,
                "rasterize_bwd_det: checkpoint_interval %d is not a power of two >= 64", checkpoint_interval);
    MGS_REQUIRE(render_out, "rasterize_bwd_det: checkpoints need render_out (the forward's frame)");
    MGS_REQUIRE(!expected_render || expected_render == render_out, "rasterize_bwd_det: expected_render and render_out are the same frame");
    while ((1 << ckpt_shift) < checkpoint_interval) ++ckpt_shift;
  }
  const size_t n_seg_units = checkpoints ? ckpt_units((uint32_t)cap, tile_w * tile_h, ckpt_shift) : 0;
  // behind the flags: the launch order of the whole-tile walk, or the unit tables of the segmented one (their 64
  // counters first: zeroed together with the flags)
  const size_t order_bytes = align_up(std::max((size_t)tile_w * tile_h * sizeof(int32_t),
                                               checkpoints ? unit_tables_bytes(tile_w * tile_h, ckpt_shift, (uint32_t)cap) : 0), 256);
  // the lists of big rectangles (reduce_records_rows_kernel): their kBigLists counters right behind the flags (zeroed with
  // them), the items at the end
  constexpr size_t big_hdr = 256;
  static_assert(kBigLists * sizeof(uint32_t) <= big_hdr, "the big lists' counters fit their header");
  const bool big_path = MGS_REDUCE_ROWS && channels <= 4 && kSlots == 1;
  const size_t big_bytes = big_path ? align_up(big_lists_bytes((uint32_t)cap), 256) : 0;
  const size_t need = rec_bytes + flag_bytes + big_hdr + order_bytes + big_bytes;
  if (!workspace) {
    *workspace_bytes = need;
    return MGS_OK;
  }
  if (*workspace_bytes < need)
    return set_error(MGS_ERR_WORKSPACE_TOO_SMALL, "rasterize_bwd_det: workspace %zu < %zu bytes",
                     *workspace_bytes, need);
  if (n == 0) return MGS_OK;
  MGS_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15u) == 0, "rasterize_bwd_det: workspace must be 16-byte aligned");
  MGS_REQUIRE(!splats || channels <= 4, "rasterize_bwd_det: packed splats carry at most 4 channels");
  MGS_REQUIRE(channels != 4 || (((reinterpret_cast<uintptr_t>(v_render) | reinterpret_cast<uintptr_t>(expected_render) |
                                  reinterpret_cast<uintptr_t>(render_out)) & 15u) == 0),
              "rasterize_bwd_det: 4-channel frames (v_render, expected_render, render_out) must be 16-byte aligned");
  const bool records_only = (call_flags & MGS_RASTER_BWD_RECORDS_ONLY) != 0;
  const int splat_slots = (call_flags & MGS_RASTER_BWD_SPLAT_SLOTS) != 0 ? 1 : 0;
  MGS_REQUIRE(!splat_slots || splats, "rasterize_bwd_det: MGS_RASTER_BWD_SPLAT_SLOTS needs the splat records");
  MGS_REQUIRE((splats || (means2d && conics && feats && opacities)) && tile_offsets && flatten_ids &&
                  alphas && last_ids && v_render && pair_info &&
                  (records_only || (v_means2d && v_conics && v_feats && v_opacities)), "rasterize_bwd_det: null pointer");
  const int n_tiles = tile_w * tile_h;
  hipStream_t s = (hipStream_t)stream;
  float* records = static_cast<float*>(workspace);
  uint8_t* flags = static_cast<uint8_t*>(workspace) + rec_bytes;
  uint32_t* big_counts = reinterpret_cast<uint32_t*>(flags + flag_bytes);
  uint32_t* big_items = reinterpret_cast<uint32_t*>(flags + flag_bytes + big_hdr + order_bytes);
  hipError_t e = hipMemsetAsync(flags, 0, flag_bytes + big_hdr + (split ? 256 : 0), s);
  if (e != hipSuccess) return set_error((int)e, "rasterize_bwd_det: memset: %s", hipGetErrorString(e));
  const int4* info = reinterpret_cast<const int4*>(pair_info);
  const int32_t* order = nullptr;
  int32_t* seg_table = nullptr;
  if (split) {
    seg_table = reinterpret_cast<int32_t*>(static_cast<uint8_t*>(workspace) + rec_bytes + flag_bytes + big_hdr);
    // (+ the workgroups that list the big rectangles for the reduce)
    const unsigned n_disc = big_path && !records_only ? div_up(n, 256 * kDiscoverPerThread) : 0;
    hipLaunchKernelGGL(unit_table_kernel, dim3(div_up(n_tiles, 256) + n_disc), dim3(256), 0, s, n_tiles, tile_offsets,
                       reinterpret_cast<const int32_t*>(checkpoints), ckpt_shift, (uint32_t)cap, seg_table, n, info, big_counts, big_items);
  } else if (!kHalf && MGS_RASTER_BWD_ORDER) {
    order = tile_group_order;
    if (!order) {       // the caller's lists came without one (mgs_isect_tiles writes it): compute it here
      int32_t* mine = reinterpret_cast<int32_t*>(static_cast<uint8_t*>(workspace) + rec_bytes + flag_bytes + big_hdr);
      const int rc = launch_tile_group_order(n_tiles, tile_offsets, mine, s);
      if (rc) return rc;
      order = mine;
    }
  }
  if (!split && big_path && !records_only)
    hipLaunchKernelGGL(big_discover_kernel, dim3(div_up(n, 256 * kDiscoverPerThread)), dim3(256), 0, s, n, info, (uint32_t)cap, big_counts, big_items);
  // (segmented: whole blocks of the XCD-aware unit numbering; units past the live count leave at once)
  const int n_units = split ? (int)((n_seg_units + 8 * MGS_RASTER_BWD_XCD_RUN - 1) / (8 * MGS_RASTER_BWD_XCD_RUN)) * 8 * MGS_RASTER_BWD_XCD_RUN
                            : order ? (n_tiles + 3) / 4 * 4 : n_tiles * (int)kSlots;
#define MGS_RD_RASTER(C, A, SP)                                                                 \
  hipLaunchKernelGGL((raster_bwd_kernel<C, A, true, kHalf, SP>), dim3(div_up(n_units, MGS_RASTER_BWD_WG_WAVES)), dim3(64 * MGS_RASTER_BWD_WG_WAVES), MGS_RASTER_BWD_LDS_PAD, s, means2d,   \
                     conics, feats, opacities, reinterpret_cast<const float4*>(splats),        \
                     background, channels, width, height, tile_w,                              \
                     n_tiles, tile_offsets, flatten_ids, alphas, last_ids, v_render, v_alphas, \
                     (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr,       \
                     (float*)nullptr, info, records, flags, (uint32_t)cap, expected_render,    \
                     (const int32_t*)order, checkpoints, ckpt_shift, (const int32_t*)seg_table, render_out, splat_slots)
#define MGS_RD_LAUNCH(C, A)                                                                     \
  if (split) MGS_RD_RASTER(C, A, ((C) <= 4 && !kHalf)); else MGS_RD_RASTER(C, A, false);        \
  if (!records_only) {                                                                          \
    if constexpr (MGS_REDUCE_ROWS && (C) <= 4 && kSlots == 1)                                   \
      hipLaunchKernelGGL((reduce_records_rows_kernel<((C) <= 4 ? (C) : 4), A>), dim3(kBigWGs + div_up(n, 256)), dim3(256), 0, s, n, \
                         info, records, flags, (uint32_t)cap, means2d, conics, opacities,      \
                         reinterpret_cast<const float4*>(splats),                              \
                         channels, v_means2d, v_means2d_abs, v_conics, v_feats, v_opacities, big_counts, big_items);  \
    else                                                                                        \
      hipLaunchKernelGGL((reduce_records_kernel<C, A, (int)kSlots>), dim3(div_up(n, 256)), dim3(256), 0, s, n,   \
                         info, records, flags, (uint32_t)cap, means2d, conics, opacities,      \
                         reinterpret_cast<const float4*>(splats),                              \
                         channels, v_means2d, v_means2d_abs, v_conics, v_feats, v_opacities); }
#define MGS_RD(C) if (v_means2d_abs) { MGS_RD_LAUNCH(C, true); } else { MGS_RD_LAUNCH(C, false); }
  if (channels == 1) { MGS_RD(1) }
  else if (channels == 2) { MGS_RD(2) }
  else if (channels == 3) { MGS_RD(3) }
  else if (channels == 4) { MGS_RD(4) }
  else if (channels <= 8) { MGS_RD(8) }
  else if (channels <= 16) { MGS_RD(16) }
  else { MGS_RD(32) }
#undef MGS_RD
#undef MGS_RD_LAUNCH
#undef MGS_RD_RASTER
  return check_launch("rasterize_bwd_det");
}
