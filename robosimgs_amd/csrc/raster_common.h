// raster_common.h -- pieces shared by the forward and backward tile raster kernels.
//
// Geometry (raster_fwd_kernel, raster_bwd_kernel; raster_fwd_q_kernel gives each of the four
// quadrants its own wave, one pixel per lane).  One wave64 owns one 16x16 tile.  Lane l holds FOUR pixels, one in each 8x8
// quadrant k of the tile: (x, y) = (8*(k&1) + (l&7), 8*(k>>1) + (l>>3)).  A Gaussian's
// parameters are therefore fetched once per 256 pixel evaluations, the four per-lane pixel
// chains give the VALU independent work, and a quadrant whose 64 pixels cannot be touched
// by a Gaussian (or are all finished) is skipped with a scalar branch.
//
// Queue.  The tile's depth-ordered list is consumed in batches of 64 (one Gaussian per
// lane).  Each lane tests its Gaussian against the four quadrants with an EXACT minimum of
// the conic's quadratic form over the quadrant's pixel-centre rectangle; Gaussians that
// cannot reach alpha >= 1/255 anywhere in the tile are dropped, survivors are compacted
// into a wave-private LDS queue with ballot + mbcnt, and the wave then walks the queue with
// broadcast ds_read_b128.  Dropping a Gaussian never changes a pixel: every pixel it would
// have evaluated fails the alpha >= 1/255 test of A.2 step 9.  No workgroup barrier exists
// anywhere in the kernel.
#ifndef MGS_RASTER_COMMON_H_
#define MGS_RASTER_COMMON_H_

#include "mgs_common.h"

namespace mgs {

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kAlphaMax = 0.999f;
constexpr float kTStop = 1e-4f;
constexpr int kQueue = 64;

struct QuadRect { float x0, x1, y0, y1; };   // pixel-centre extents of an 8x8 quadrant

// 1D clamped minimum of the quadratic along one rectangle edge.
//   fixed offset e (edge coordinate - mean) on the "u" axis with weight wu,
//   free offset in [lo, hi] on the "v" axis with weight wv, cross term b.
__device__ __forceinline__ float edge_min_sigma(float e, float wu, float wv, float b,
                                                float inv_wv, float lo, float hi) {
  float v = fminf(fmaxf(-b * e * inv_wv, lo), hi);
  return 0.5f * (wu * e * e + wv * v * v) + b * e * v;
}

// min over the rectangle of 0.5*(a dx^2 + c dy^2) + b dx dy, (dx,dy) = p - mean.
__device__ __forceinline__ float rect_min_sigma(float mx, float my, float a, float b, float c,
                                                float inv_a, float inv_c, const QuadRect& r) {
  float lx = r.x0 - mx, hx = r.x1 - mx, ly = r.y0 - my, hy = r.y1 - my;
  if (lx <= 0.f && hx >= 0.f && ly <= 0.f && hy >= 0.f) return 0.f;
  float s0 = edge_min_sigma(lx, a, c, b, inv_c, ly, hy);
  float s1 = edge_min_sigma(hx, a, c, b, inv_c, ly, hy);
  float s2 = edge_min_sigma(ly, c, a, b, inv_a, lx, hx);
  float s3 = edge_min_sigma(hy, c, a, b, inv_a, lx, hx);
  return fminf(fminf(s0, s1), fminf(s2, s3));
}

// Bit k set <=> the Gaussian may reach alpha >= 1/255 at some pixel centre of quadrant k.
// Conservative by `slack` (fp32 rounding of both this test and the per-pixel sigma).
__device__ __forceinline__ unsigned quadrant_mask(float mx, float my, float a, float b, float c,
                                                  float opac, float tile_x, float tile_y) {
  if (!(opac >= kAlphaMin)) return 0u;           // alpha <= opac < 1/255 everywhere
#if defined(MGS_CULL_IEEE_DIV)
  float thr = __logf(255.0f * opac);
#else
  // (one v_log_f32: __logf compiles to the denormal-safe sequence of fourteen; 255 opac >= 1 here, and the slack below
  //  is five orders above the instruction's error)
  float thr = 0.6931471805599453f * __builtin_amdgcn_logf(255.0f * opac);
#endif
  // (v_rcp_f32, not the IEEE divide's eleven instructions each: the reciprocals only place the clamped minimiser on an
  //  edge, where the quadratic is flat to second order -- a 1-ulp reciprocal moves sigma by ~1e-13 relative, the test
  //  below carries a slack of 0.05)
#if defined(MGS_CULL_IEEE_DIV)      // measurement: the divides
  float inv_a = 1.0f / a, inv_c = 1.0f / c;
#else
  float inv_a = __builtin_amdgcn_rcpf(a), inv_c = __builtin_amdgcn_rcpf(c);
#endif
  float fx = fmaxf(fabsf(tile_x - mx), fabsf(tile_x + 16.f - mx));
  float fy = fmaxf(fabsf(tile_y - my), fabsf(tile_y + 16.f - my));
  float slack = 0.05f + 4e-6f * (fabsf(a) + fabsf(c) + 2.f * fabsf(b)) * (fx * fx + fy * fy);
  float lim = thr + slack;
  unsigned m = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    QuadRect r;
    r.x0 = tile_x + 8.f * (k & 1) + 0.5f;
    r.x1 = r.x0 + 7.f;
    r.y0 = tile_y + 8.f * (k >> 1) + 0.5f;
    r.y1 = r.y0 + 7.f;
    float s = rect_min_sigma(mx, my, a, b, c, inv_a, inv_c, r);
    if (!(s > lim)) m |= 1u << k;                // NaN keeps the quadrant
  }
  return m;
}

// The blend skips a pair whose sigma comes out negative (A.2 step 9).  For a positive definite conic
// that can only be a rounding artefact, and only if sigma is tiny against its own terms: with
// S = 0.5 (|a| dx^2 + |c| dy^2) + |b dx dy|, sigma >= (lambda_min / 2 lambda_max) S while fp32 evaluates
// it to within ~6e-7 S.  A conic with lambda_min / lambda_max >= 1e-4 (det >= 1e-4 trace^2) can therefore
// never produce sigma < 0 and the test is dead for it.  When every queued conic of a 64-entry batch is
// such (nearly always) the batch is walked by a copy of the loop without the test: one compare and one
// scalar AND less per 64 pairs; a batch holding a needle-like conic keeps the full test.  Results are identical
// (the exponent below is evaluated by the same expression either way).
__device__ __forceinline__ bool sigma_sign_is_safe(float a, float b, float c) {
  const float tr = a + c;
  return a > 0.f && c > 0.f && fmaf(a, c, -b * b) >= 1e-4f * tr * tr;     // false for NaN
}

// ---- alpha of one pixel-Gaussian pair, shared bit for bit by the forward and the backward ------------------
// The queue entry carries the conic pre-scaled by -log2(e) (A = -0.5 log2e a, B = -log2e b, C = -0.5 log2e c) and
// L = log2(opacity), so that   opacity * exp(-sigma) = exp2(A dx^2 + B dx dy + C dy^2 + L)
// is ONE v_exp_f32 of ONE fused expression: L rides in the addend of the last-but-one FMA (no multiply by the
// opacity afterwards).  `pair_power` is that exponent; `pair_power_sign` is the same expression without L, needed
// only where sigma >= 0 must be tested (conics that are not sigma_sign_is_safe).
__device__ __forceinline__ float pair_power(float dx, float dy, float A, float B, float C, float L) {
  return fmaf(dx, fmaf(B, dy, A * dx), fmaf(C * dy, dy, L));
}
__device__ __forceinline__ float pair_power_sign(float dx, float dy, float A, float B, float C) {
  return fmaf(dx, fmaf(B, dy, A * dx), (C * dy) * dy);
}
// The same exponent as a polynomial in the pixel's offset (x, y) from the TILE CENTRE (half-integers up to 7.5: x,
// y, x^2, x y, y^2 are exact per-lane constants).  With m = mean - centre,
//   A (m_x - x)^2 + B (m_x - x)(m_y - y) + C (m_y - y)^2 + L  =  q0 + q1 x + q2 y + A x^2 + B x y + C y^2,
//   q0 = pair_power(m_x, m_y, A, B, C, L),  q1 = -(2 A m_x + B m_y),  q2 = -(2 C m_y + B m_x):
// five FMAs per pair and no subtraction of the mean (the q's cost a dozen instructions per QUEUED Gaussian, one lane
// each).  Evaluation error against exact arithmetic on the same fp32 inputs: median 2e-7, 99.99 % under 6e-6, worst
// 1.2e-5 relative in alpha over 4.4 M visible pairs of the configs[1] scene (the dx / dy form: 7e-8 / 1.2e-6 / 1.1e-5)
// -- both far below what the fp32 rounding of mean2d itself costs (median 8e-6, worst 1.9e-4).  Every kernel that
// needs the alpha of a (Gaussian, pixel) pair -- both forward schedules and the backward -- evaluates exactly this
// chain on exactly these coefficients: same bits everywhere.
struct PolyCoef { float q0, q1, q2; };
__device__ __forceinline__ PolyCoef poly_coefs(float m_x, float m_y, float A, float B, float C, float L) {
  PolyCoef q;
  q.q0 = pair_power(m_x, m_y, A, B, C, L);
  q.q1 = -fmaf(2.f * A, m_x, B * m_y);
  q.q2 = -fmaf(2.f * C, m_y, B * m_x);
  return q;
}
struct PixelPoly { float x, y, xx, xy, yy; };       // offsets from the tile centre and their products (exact)
__device__ __forceinline__ PixelPoly pixel_poly(float x, float y) { return PixelPoly{x, y, x * x, x * y, y * y}; }
__device__ __forceinline__ float pair_power_poly(const PixelPoly& p, float q0, float q1, float q2, float A, float B,
                                                 float C) {
  return fmaf(C, p.yy, fmaf(B, p.xy, fmaf(A, p.xx, fmaf(q2, p.y, fmaf(q1, p.x, q0)))));
}

// Batches whose queued Gaussians all have a well conditioned conic AND an opacity <= kSafeOpacity are walked
// without the sigma >= 0 test and without the 0.999 clamp: exp2(power + L) <= opacity (1 + 2^-22) < 0.999.
constexpr float kSafeOpacity = 0.998f;
// The bound above holds for the polynomial about the tile centre while its absolute error ~2^-24 |A| m^2 (m <= ~11 px)
// stays far below 2^-10: conics of the default eps2d = 0.3 have a, c <= 3.4.  A caller with a tiny eps2d can hand over
// conics of 1e3 and more; those take the clamped, sign-tested body (a + c <= 8 is always true at eps2d >= 0.25).
constexpr float kSafeConicTrace = 8.0f;
__device__ __forceinline__ bool entry_is_safe(float a, float b, float c, float op) {
  return sigma_sign_is_safe(a, b, c) && op <= kSafeOpacity && a + c <= kSafeConicTrace;
}

// ---- forward checkpoints for the segmented backward ---------------------------------------------------------
// A tile's list is cut into SEGMENTS of S = 1 << shift entries (S a multiple of the 64-entry batch, aligned to the
// list's start, so the batches -- and with them every queue and every per-entry value -- are those of the unsegmented
// walk).  The training forward stores, at the first batch of every segment s >= 1 that some pixel of the block still
// reaches, the per-pixel state in front of that batch -- T and the accumulated channels -- and the backward then walks
// every segment as an independent unit of work: a segment that has a successor starts from the successor's
// checkpoint (T as stored; colour behind = final - checkpoint) instead of from the far end of the list.
// Unit of (tile t, segment s) = (start_t >> shift) + t + s: injective (floor((a + L) / S) - floor(a / S) >=
// ceil(L / S) - 1), bounded by (capacity >> shift) + n_tiles, and needs no prefix sum over the tiles.
// Layout: checkpoints[unit][1 + channels][256] floats, pixel k * 64 + lane of the tile (quadrant k, raster_common.h),
// behind a header of 4 int32 per tile: the largest last_id of each of the tile's four 8x8 blocks (-1: not written by
// this forward schedule) -- where the backward's walk of the tile ends, known before any pixel is loaded.
__host__ __device__ constexpr size_t ckpt_units(uint32_t capacity, int n_tiles, int shift) {
  return ((size_t)capacity >> shift) + (size_t)n_tiles + 1;
}
__host__ __device__ constexpr size_t ckpt_header_floats(int n_tiles) { return ((size_t)n_tiles * 4 + 255) / 256 * 256; }
__device__ __forceinline__ size_t ckpt_unit(int start, int tile, int seg, int shift) {
  return (size_t)(start >> shift) + (size_t)tile + (size_t)seg;
}

// A queue entry part is read from LDS as ONE ds_read_b128: an empty asm that "uses" all four lanes of the register
// tuple keeps the compiler from narrowing the load to the components the caller happens to touch (it split a 16-byte
// read into b64 + b32 + 2 x read2_b32: five LDS instructions per entry instead of three).
typedef float mgs_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_read_3f4(const float4* pa, const float4* pb, const float4* pc, float4& a, float4& b,
                                             float4& c) {
  mgs_f4 va = *reinterpret_cast<const mgs_f4*>(pa), vb = *reinterpret_cast<const mgs_f4*>(pb),
         vc = *reinterpret_cast<const mgs_f4*>(pc);
  asm volatile("" : "+v"(va), "+v"(vb), "+v"(vc));      // (one statement: the three reads stay in flight together)
  a = make_float4(va.x, va.y, va.z, va.w);
  b = make_float4(vb.x, vb.y, vb.z, vb.w);
  c = make_float4(vc.x, vc.y, vc.z, vc.w);
}

// full-wave sum: result valid in lane 63
__device__ __forceinline__ float wave_reduce_to_lane63(float v) {
  int i;
#define MGS_DPP_ADD(CTRL, RMASK)                                                               \
  i = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, RMASK, 0xf, false);              \
  v += __int_as_float(i);
  MGS_DPP_ADD(0xB1, 0xf)    // quad_perm [1,0,3,2]
  MGS_DPP_ADD(0x4E, 0xf)    // quad_perm [2,3,0,1]
  MGS_DPP_ADD(0x141, 0xf)   // row_half_mirror
  MGS_DPP_ADD(0x140, 0xf)   // row_mirror
  MGS_DPP_ADD(0x142, 0xa)   // row_bcast15 -> rows 1,3
  MGS_DPP_ADD(0x143, 0xc)   // row_bcast31 -> rows 2,3
#undef MGS_DPP_ADD
  return v;
}


// ---- reduce-scatter butterfly over the wave ----------------------------------------------
// Sums V values (V = 8, 4, 2 or 1) held by every lane over all 64 lanes in about 3V + 5
// instructions instead of 6V.  Scatter steps halve the value count while pairing lanes over
// lane bits 0, 1 and 3 (xor 1 / xor 2 via quad_perm, xor 8 via row_ror:8): a lane keeps the
// half selected by its own bit and adds the partner's copy of that half.  The remaining lane
// bits are then summed with plain shifted adds.  The total of value i ends in the lane of row 0
// with (bit0, bit1, bit3) spelling i as  i = bit3 + 2*bit1 + 4*bit0  (V = 8),
// i = bit1 + 2*bit0 (V = 4), i = bit0 (V = 2), lane 0 (V = 1), and lane bits 2, 4, 5 clear.
__device__ __forceinline__ float dpp_f(float x, int ctrl) {   // ctrl must fold to a constant
  switch (ctrl) {
    case 0xB1: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false));
    case 0x4E: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, false));
    case 0x128: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false));
    default: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x104, 0xf, 0xf, false));
  }
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppRor8 = 0x128, kDppShl4 = 0x104;

template <int V>
__device__ __forceinline__ float wave_reduce_scatter(const float* v, unsigned lane) {
  static_assert(V == 8 || V == 4 || V == 2 || V == 1, "V must be 8, 4, 2 or 1");
  const bool b0 = lane & 1u, b1 = lane & 2u, b3 = lane & 8u;
  float t;
  if constexpr (V == 8) {
    float w[4], u[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (b0 ? v[i + 4] : v[i]) + dpp_f(b0 ? v[i] : v[i + 4], kDppXor1);
#pragma unroll
    for (int i = 0; i < 2; ++i) u[i] = (b1 ? w[i + 2] : w[i]) + dpp_f(b1 ? w[i] : w[i + 2], kDppXor2);
    t = (b3 ? u[1] : u[0]) + dpp_f(b3 ? u[0] : u[1], kDppRor8);
  } else if constexpr (V == 4) {
    float u[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) u[i] = (b0 ? v[i + 2] : v[i]) + dpp_f(b0 ? v[i] : v[i + 2], kDppXor1);
    t = (b1 ? u[1] : u[0]) + dpp_f(b1 ? u[0] : u[1], kDppXor2);
    t += dpp_f(t, kDppRor8);
  } else if constexpr (V == 2) {
    t = (b0 ? v[1] : v[0]) + dpp_f(b0 ? v[0] : v[1], kDppXor1);
    t += dpp_f(t, kDppXor2);
    t += dpp_f(t, kDppRor8);
  } else {
    t = v[0] + dpp_f(v[0], kDppXor1);
    t += dpp_f(t, kDppXor2);
    t += dpp_f(t, kDppRor8);
  }
  t += dpp_f(t, kDppShl4);             // lane bit 2 (valid where bit 2 is clear)
  t += __shfl_down(t, 16);             // rows 1,3 into rows 0,2
  t += __shfl_down(t, 32);             // row 2 into row 0
  return t;
}
// which value index a holder lane (row 0, bit 2 clear) carries; -1 for other lanes
template <int V>
__device__ __forceinline__ int wave_reduce_scatter_index(unsigned lane) {
  if (lane & ~0xBu) return -1;         // only lanes 0,1,2,3,8,9,10,11 can hold a total
  const int b0 = lane & 1, b1 = (lane >> 1) & 1, b3 = (lane >> 3) & 1;
  if (V == 8) return b3 + 2 * b1 + 4 * b0;
  if (V == 4) return b3 ? -1 : b1 + 2 * b0;
  if (V == 2) return (b3 || b1) ? -1 : b0;
  return lane == 0 ? 0 : -1;
}

}  // namespace mgs
#endif
