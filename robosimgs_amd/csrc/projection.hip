// projection.hip -- per-Gaussian stage of the render path for gfx950:
//   * mgs_projection_fwd      world -> screen EWA projection (A.2 steps 1-5)
//   * mgs_sh_fwd              real SH colour (A.2 step 6)
//   * mgs_project_color_fwd   both fused; SH rows of culled Gaussians are never fetched
//
// HBM-bound streaming kernels.  One Gaussian per lane.  The 192-byte SH row of a Gaussian
// (16 coefficients x rgb) would be a 192-byte-strided access per lane; instead each wave
// fetches its 64 rows as 768 consecutive 16-byte pieces (lane-linear, 1 KiB per
// instruction), parks them in LDS with a 208-byte row pitch, and every lane then reads its
// own row back with conflict-free ds_read_b128 (52-dword pitch: 13 is odd, so the 16 lanes
// of a b128 service group land on 16 distinct 4-bank slots).
#include "mgs_common.h"
#include "mgs_math.h"
#include "sh_staging.h"
#include "tile_rect.h"

namespace mgs {
namespace {

// One wave per workgroup: with several frames in flight the raster's one-wave workgroups take every
// wave slot as it frees up, and a 4-wave workgroup (which needs four slots on one CU at once) waits;
// 256 -> 64 threads: 2936 -> 3060 frames/s at three frames in flight, same 55 us alone.
constexpr int kBlock = 64;
constexpr int kWave = kShWave;

__device__ __forceinline__ void load3(const float* p, float v[3]) {
  v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
}

__global__ __launch_bounds__(kBlock) void projection_fwd_kernel(
    int n, const float* __restrict__ means, const float* __restrict__ quats,
    const float* __restrict__ scales, const float* __restrict__ viewmat,
    const float* __restrict__ Kmat, float W, float H, float eps2d, float near_plane,
    float far_plane, float radius_clip, int32_t* __restrict__ radii,
    float* __restrict__ means2d, float* __restrict__ depths, float* __restrict__ conics,
    float* __restrict__ compensations, const float* __restrict__ opacities, int radius_rule,
    int32_t* __restrict__ radii_y) {
  int g = blockIdx.x * kBlock + threadIdx.x;
  if (g >= n) return;
  CameraParams cam = load_camera(viewmat, Kmat);
  float m[3], s[3], q[4];
  load3(means + 3 * (size_t)g, m);
  load3(scales + 3 * (size_t)g, s);
  float4 qq = reinterpret_cast<const float4*>(quats)[g];
  q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
  Projected p = project_gaussian(m, q, s, cam, W, H, eps2d, near_plane, far_plane, radius_clip, radius_rule,
                                 opacities != nullptr, opacities ? opacities[g] : 1.f, compensations != nullptr);
  radii[g] = p.radius;
  if (radii_y) radii_y[g] = p.radius_y;
  reinterpret_cast<float2*>(means2d)[g] = make_float2(p.mean2d[0], p.mean2d[1]);
  depths[g] = p.depth;
  conics[3 * (size_t)g + 0] = p.conic[0];
  conics[3 * (size_t)g + 1] = p.conic[1];
  conics[3 * (size_t)g + 2] = p.conic[2];
  if (compensations) compensations[g] = p.compensation;
}

// Evaluate sum_k Y_k(dir) * coeff_k for KC = (DEG+1)^2 coefficients held in registers.
template <int DEG>
__device__ __forceinline__ void sh_dot(const float dir[3], const float* c /*[KC*3]*/, float rgb[3]) {
  constexpr int KC = (DEG + 1) * (DEG + 1);
  float n2 = dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2];
  float inv = n2 > 0.f ? 1.0f / sqrtf(n2) : 0.f;
  float Y[KC];
  sh_basis(DEG, dir[0] * inv, dir[1] * inv, dir[2] * inv, Y);
  float r = 0.f, g = 0.f, b = 0.f;
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    r += Y[k] * c[3 * k + 0];
    g += Y[k] * c[3 * k + 1];
    b += Y[k] * c[3 * k + 2];
  }
  rgb[0] = r; rgb[1] = g; rgb[2] = b;
}

// Fetch this lane's SH row (first KC*3 floats of a row of `stride_f` floats).
//   STAGED (stride_f == 48): wave-cooperative lane-linear fetch through LDS.
//   otherwise: direct per-lane loads (rows of 3 / 12 / 27 floats are short).
// `active` = this lane needs its row; `wave_mask` = ballot(active); g0 = first Gaussian of
// the wave.  lds points at this wave's kWave*kShPitchF4 float4 slots.
template <int DEG, bool STAGED>
__device__ __forceinline__ void fetch_sh_row(const float* __restrict__ coeffs, int stride_f,
                                             int g, int g0, int n, bool active,
                                             unsigned long long wave_mask, float4* lds,
                                             float* c) {
  constexpr int KC = (DEG + 1) * (DEG + 1);
  if constexpr (STAGED) {
    const unsigned lane = threadIdx.x & (kWave - 1);
    sh_rows_to_lds(coeffs, g0, n, wave_mask, lds);
    __syncthreads();
    constexpr int NF4 = (KC * 3 + 3) / 4;
#pragma unroll
    for (int j = 0; j < NF4; ++j) {
      float4 v = lds[lane * kShPitchF4 + j];
      c[4 * j + 0] = v.x;
      if (4 * j + 1 < KC * 3) c[4 * j + 1] = v.y;
      if (4 * j + 2 < KC * 3) c[4 * j + 2] = v.z;
      if (4 * j + 3 < KC * 3) c[4 * j + 3] = v.w;
    }
  } else {
    if (active) {
      const float* row = coeffs + (size_t)g * stride_f;
#pragma unroll
      for (int k = 0; k < KC * 3; ++k) c[k] = row[k];
    } else {
#pragma unroll
      for (int k = 0; k < KC * 3; ++k) c[k] = 0.f;
    }
  }
}

constexpr int sh_reg_floats(int deg) { return (((deg + 1) * (deg + 1) * 3 + 3) / 4) * 4; }

template <int DEG, bool STAGED>
__global__ __launch_bounds__(kBlock) void sh_fwd_kernel(
    int n, int stride_f, const float* __restrict__ dirs, const float* __restrict__ coeffs,
    const uint8_t* __restrict__ masks, float* __restrict__ colors) {
  __shared__ float4 lds[STAGED ? (kBlock / kWave) * kWave * kShPitchF4 : 1];
  int g = blockIdx.x * kBlock + threadIdx.x;
  bool active = g < n && (masks == nullptr || masks[g] != 0);
  unsigned long long wave_mask = ballot(active);
  int g0 = blockIdx.x * kBlock + (threadIdx.x & ~(kWave - 1));
  float c[sh_reg_floats(DEG)];
  fetch_sh_row<DEG, STAGED>(coeffs, stride_f, g, g0, n, active, wave_mask,
                            lds + (threadIdx.x / kWave) * kWave * kShPitchF4, c);
  if (g >= n) return;
  float rgb[3] = {0.f, 0.f, 0.f};
  if (active) {
    float d[3];
    load3(dirs + 3 * (size_t)g, d);
    sh_dot<DEG>(d, c, rgb);
  }
  colors[3 * (size_t)g + 0] = rgb[0];
  colors[3 * (size_t)g + 1] = rgb[1];
  colors[3 * (size_t)g + 2] = rgb[2];
}

template <int DEG, bool STAGED, int RULE>
__global__ __launch_bounds__(kBlock) void project_color_fwd_kernel(
    int n, const float* __restrict__ means, const float* __restrict__ quats,
    const float* __restrict__ scales, const float* __restrict__ opacities, int stride_f,
    const float* __restrict__ coeffs, const float* __restrict__ viewmat,
    const float* __restrict__ Kmat, float W, float H, float eps2d, float near_plane,
    float far_plane, float radius_clip, int32_t* __restrict__ radii,
    float* __restrict__ means2d, float* __restrict__ depths, float* __restrict__ conics,
    float* __restrict__ opac_out, int feat_stride, float* __restrict__ feats,
    float4* __restrict__ splats, uint2* __restrict__ bin_info, uint32_t* __restrict__ bin_sums,
    int bin_tight, int tile_w, int tile_h, int32_t* __restrict__ radii_y) {
  __shared__ float4 lds[STAGED ? (kBlock / kWave) * kWave * kShPitchF4 : 1];
  int g = blockIdx.x * kBlock + threadIdx.x;
  CameraParams cam = load_camera(viewmat, Kmat);
  float m[3] = {0.f, 0.f, 0.f};
  Projected p;
  p.radius = 0;
  if (g < n) {
    float s[3], q[4];
    load3(means + 3 * (size_t)g, m);
    load3(scales + 3 * (size_t)g, s);
    float4 qq = reinterpret_cast<const float4*>(quats)[g];
    q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
    if (RULE == MGS_RADIUS_CLASSIC)
      p = project_gaussian(m, q, s, cam, W, H, eps2d, near_plane, far_plane, radius_clip);
    else
      p = project_gaussian(m, q, s, cam, W, H, eps2d, near_plane, far_plane, radius_clip, RULE, opacities != nullptr,
                           opacities ? opacities[g] : 1.f, opac_out != nullptr);
    // radii / means2d / conics (and feats below) are null in an inference frame: the raster reads the packed
    // records, the seeded binning reads bin_info + depths -- 36 MB of stores per 1 M Gaussians nobody would read
    if (radii) radii[g] = p.radius;
    if (RULE != MGS_RADIUS_CLASSIC && radii_y) radii_y[g] = p.radius_y;
    if (means2d) reinterpret_cast<float2*>(means2d)[g] = make_float2(p.mean2d[0], p.mean2d[1]);
    depths[g] = p.depth;
    if (conics) {
      conics[3 * (size_t)g + 0] = p.conic[0];
      conics[3 * (size_t)g + 1] = p.conic[1];
      conics[3 * (size_t)g + 2] = p.conic[2];
    }
    if (opac_out) opac_out[g] = opacities[g] * p.compensation;
  }
  bool active = p.radius > 0;
  // seed of the binning (mgs_isect_tiles seed_info / seed_sums): this Gaussian's tile rectangle and
  // tile count, and the count sum of the wave's 64 Gaussians -- saves the binning its own pass over
  // means2d / radii / conics / opacities and one launch
  if (bin_info) {
    uint2 info = make_uint2(kEmptyTileRect, 0u);
    if (active) {
      TileRect r = RULE == MGS_RADIUS_CLASSIC
                       ? tile_rect(p.mean2d[0], p.mean2d[1], p.radius, (float)MGS_TILE_SIZE, tile_w, tile_h)
                       : tile_rect(p.mean2d[0], p.mean2d[1], p.radius, p.radius_y, (float)MGS_TILE_SIZE, tile_w, tile_h);
      if (bin_tight)
        r = tighten_rect(r, p.mean2d[0], p.mean2d[1], p.conic[0], p.conic[1], p.conic[2],
                         opacities[g] * (opac_out ? p.compensation : 1.f), (float)MGS_TILE_SIZE);
      info = pack_tile_rect(r);
    }
    if (g < n) bin_info[g] = info;
    uint32_t c = info.y;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & (kWave - 1)) == 0 && g < n) bin_sums[g / kWave] = c;
  }
  unsigned long long wave_mask = ballot(active);
  int g0 = blockIdx.x * kBlock + (threadIdx.x & ~(kWave - 1));
  float c[sh_reg_floats(DEG)];
  fetch_sh_row<DEG, STAGED>(coeffs, stride_f, g, g0, n, active, wave_mask,
                            lds + (threadIdx.x / kWave) * kWave * kShPitchF4, c);
  if (g >= n) return;
  float rgb[3] = {0.f, 0.f, 0.f};
  if (active) {
    float campos[3], d[3];
    camera_position(cam, campos);
    d[0] = m[0] - campos[0]; d[1] = m[1] - campos[1]; d[2] = m[2] - campos[2];
    sh_dot<DEG>(d, c, rgb);
    rgb[0] = fmaxf(rgb[0] + 0.5f, 0.f);
    rgb[1] = fmaxf(rgb[1] + 0.5f, 0.f);
    rgb[2] = fmaxf(rgb[2] + 0.5f, 0.f);
  }
  // one 48-byte record per visible Gaussian for the raster kernels' list gathers (culled
  // Gaussians never enter a tile list: their records are left unwritten)
  if (splats && active) {
    float op = opacities ? opacities[g] * (opac_out ? p.compensation : 1.f) : 0.f;
    splats[3 * (size_t)g + 0] = make_float4(p.mean2d[0], p.mean2d[1], p.conic[0], p.conic[1]);
    splats[3 * (size_t)g + 1] = make_float4(p.conic[2], op, rgb[0], rgb[1]);
    splats[3 * (size_t)g + 2] = make_float4(rgb[2], feat_stride == 4 ? p.depth : 0.f, 0.f, 0.f);
  }
  if (!feats) return;
  if (feat_stride == 4) {
    reinterpret_cast<float4*>(feats)[g] = make_float4(rgb[0], rgb[1], rgb[2], p.depth);
  } else {
    feats[3 * (size_t)g + 0] = rgb[0];
    feats[3 * (size_t)g + 1] = rgb[1];
    feats[3 * (size_t)g + 2] = rgb[2];
  }
}

}  // namespace
}  // namespace mgs

using namespace mgs;

extern "C" int mgs_projection_fwd(int n, const float* means, const float* quats,
                                  const float* scales, const float* viewmat, const float* K,
                                  int width, int height, float eps2d, float near_plane,
                                  float far_plane, float radius_clip, int32_t* radii,
                                  float* means2d, float* depths, float* conics,
                                  float* compensations, const float* opacities, int radius_rule,
                                  int32_t* radii_y, mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && width > 0 && height > 0, "projection_fwd: bad sizes n=%d %dx%d", n, width, height);
  if (n == 0) return MGS_OK;
  MGS_REQUIRE(means && quats && scales && viewmat && K && radii && means2d && depths && conics,
              "projection_fwd: null pointer");
  MGS_REQUIRE(radius_rule == MGS_RADIUS_CLASSIC || radius_rule == MGS_RADIUS_OPACITY_AWARE,
              "projection_fwd: radius_rule %d is neither MGS_RADIUS_CLASSIC nor MGS_RADIUS_OPACITY_AWARE", radius_rule);
  MGS_REQUIRE(radius_rule == MGS_RADIUS_CLASSIC || radii_y, "projection_fwd: the per-axis radius rule needs radii_y");
  hipLaunchKernelGGL(projection_fwd_kernel, dim3(div_up(n, kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, n, means, quats, scales, viewmat, K, (float)width,
                     (float)height, eps2d, near_plane, far_plane, radius_clip, radii, means2d,
                     depths, conics, compensations, opacities, radius_rule, radii_y);
  return check_launch("projection_fwd");
}

extern "C" int mgs_sh_fwd(int n, int degree, int coeff_stride, const float* dirs,
                          const float* coeffs, const uint8_t* masks, float* colors,
                          mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && degree >= 0 && degree <= 3, "sh_fwd: degree %d not in 0..3", degree);
  MGS_REQUIRE(coeff_stride >= (degree + 1) * (degree + 1), "sh_fwd: %d coefficients < (degree+1)^2", coeff_stride);
  if (n == 0) return MGS_OK;
  MGS_REQUIRE(dirs && coeffs && colors, "sh_fwd: null pointer");
  dim3 grid(div_up(n, kBlock)), block(kBlock);
  hipStream_t s = (hipStream_t)stream;
  int sf = coeff_stride * 3;
  bool staged = coeff_stride == 16 && degree >= 2;
#define MGS_SH_LAUNCH(D, S) \
  hipLaunchKernelGGL((sh_fwd_kernel<D, S>), grid, block, 0, s, n, sf, dirs, coeffs, masks, colors)
  switch (degree) {
    case 0: MGS_SH_LAUNCH(0, false); break;
    case 1: MGS_SH_LAUNCH(1, false); break;
    case 2: if (staged) MGS_SH_LAUNCH(2, true); else MGS_SH_LAUNCH(2, false); break;
    default: if (staged) MGS_SH_LAUNCH(3, true); else MGS_SH_LAUNCH(3, false); break;
  }
#undef MGS_SH_LAUNCH
  return check_launch("sh_fwd");
}

extern "C" int mgs_project_color_fwd(int n, const float* means, const float* quats,
                                     const float* scales, const float* opacities,
                                     int sh_degree, int coeff_stride, const float* sh_coeffs,
                                     const float* viewmat, const float* K, int width,
                                     int height, float eps2d, float near_plane,
                                     float far_plane, float radius_clip, int32_t* radii,
                                     float* means2d, float* depths, float* conics,
                                     float* opac_out, int feat_stride, float* feats,
                                     float* splats, int bin_flags, uint32_t* bin_info,
                                     uint32_t* bin_sums, int32_t* radii_y, mgs_stream_t stream) {
  const int bin_tight = bin_flags & MGS_BIN_TIGHT;
  const bool per_axis = (bin_flags & MGS_BIN_RADIUS_OPACITY_AWARE) != 0;
  MGS_REQUIRE(n >= 0 && width > 0 && height > 0, "project_color_fwd: bad sizes");
  MGS_REQUIRE(sh_degree >= 0 && sh_degree <= 3, "project_color_fwd: sh_degree %d not in 0..3", sh_degree);
  MGS_REQUIRE(coeff_stride >= (sh_degree + 1) * (sh_degree + 1), "project_color_fwd: coeff_stride too small");
  MGS_REQUIRE(feat_stride == 3 || feat_stride == 4, "project_color_fwd: feat_stride must be 3 or 4");
  if (n == 0) return MGS_OK;
  MGS_REQUIRE(means && quats && scales && sh_coeffs && viewmat && K && depths, "project_color_fwd: null pointer");
  // the per-Gaussian arrays may be dropped only when their consumers have a substitute: the raster reads
  // `splats`, the binning is seeded with `bin_info`
  MGS_REQUIRE((radii && means2d && conics && feats) || (splats && bin_info),
              "project_color_fwd: radii / means2d / conics / feats may be NULL only when splats and bin_info are given");
  MGS_REQUIRE(!opac_out || opacities, "project_color_fwd: opac_out needs opacities");
  MGS_REQUIRE(!splats || opacities, "project_color_fwd: splats needs opacities");
  MGS_REQUIRE((bin_info == nullptr) == (bin_sums == nullptr), "project_color_fwd: bin_info and bin_sums come together");
  MGS_REQUIRE(!(bin_info && bin_tight) || opacities, "project_color_fwd: tight tile bounds need opacities");
  MGS_REQUIRE(!per_axis || !radii || radii_y, "project_color_fwd: the per-axis radius rule writes radii_y beside radii");
  const int tile_w = (width + MGS_TILE_SIZE - 1) / MGS_TILE_SIZE, tile_h = (height + MGS_TILE_SIZE - 1) / MGS_TILE_SIZE;
  MGS_REQUIRE(!bin_info || (tile_w <= 1023 && tile_h <= 1023), "project_color_fwd: tile grid exceeds 1023x1023");
  dim3 grid(div_up(n, kBlock)), block(kBlock);
  hipStream_t s = (hipStream_t)stream;
  int sf = coeff_stride * 3;
  bool staged = coeff_stride == 16 && sh_degree >= 2;
#define MGS_PC_LAUNCH(D, S)                                                                   \
  if (per_axis) MGS_PC_LAUNCH_R(D, S, MGS_RADIUS_OPACITY_AWARE); else MGS_PC_LAUNCH_R(D, S, MGS_RADIUS_CLASSIC)
#define MGS_PC_LAUNCH_R(D, S, R)                                                              \
  hipLaunchKernelGGL((project_color_fwd_kernel<D, S, R>), grid, block, 0, s, n, means, quats, \
                     scales, opacities, sf, sh_coeffs, viewmat, K, (float)width,              \
                     (float)height, eps2d, near_plane, far_plane, radius_clip, radii,         \
                     means2d, depths, conics, opac_out, feat_stride, feats,                    \
                     reinterpret_cast<float4*>(splats), reinterpret_cast<uint2*>(bin_info),     \
                     bin_sums, bin_tight, tile_w, tile_h, radii_y)
  switch (sh_degree) {
    case 0: MGS_PC_LAUNCH(0, false); break;
    case 1: MGS_PC_LAUNCH(1, false); break;
    case 2: if (staged) { MGS_PC_LAUNCH(2, true); } else { MGS_PC_LAUNCH(2, false); } break;
    default: if (staged) { MGS_PC_LAUNCH(3, true); } else { MGS_PC_LAUNCH(3, false); } break;
  }
#undef MGS_PC_LAUNCH
#undef MGS_PC_LAUNCH_R
  return check_launch("project_color_fwd");
}
