// backward.hip -- per-Gaussian backward of the projection / colour stage for gfx950:
//   * mgs_projection_bwd      VJP of the EWA projection (A.2 steps 1-5)
//   * mgs_sh_bwd              VJP of the SH colour (A.2 step 6)
//   * mgs_project_color_bwd   both fused, the mirror of mgs_project_color_fwd
// One Gaussian per lane; HBM-bound streaming (the 192-byte SH coefficient row is read and its
// gradient row written through the same wave-cooperative LDS staging as the forward).
// Math: mgs_math.h (checked on the host against autograd of the oracle).
#include "mgs_common.h"
#include "mgs_math.h"
#include "raster_common.h"
#include "sh_staging.h"

namespace mgs {
namespace {

constexpr int kBlock = 256;
constexpr int kWave = kShWave;

__device__ __forceinline__ void load3(const float* p, float v[3]) {
  v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
}

// accumulate a per-lane 12-vector (v_R[9], v_t[3]) into v_viewmat[16] with one atomic per wave
__device__ __forceinline__ void reduce_viewmat(const ProjectedGrad& g, bool active,
                                               float* __restrict__ v_viewmat) {
  const unsigned lane = threadIdx.x & 63;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = active ? (c < 3 ? g.v_R[r * 3 + c] : g.v_t[r]) : 0.f;
      v = wave_reduce_to_lane63(v);
      if (lane == 63 && v != 0.f) unsafeAtomicAdd(&v_viewmat[r * 4 + c], v);
    }
  }
}

__global__ __launch_bounds__(kBlock) void projection_bwd_kernel(
    int n, const float* __restrict__ means, const float* __restrict__ quats,
    const float* __restrict__ scales, const float* __restrict__ viewmat,
    const float* __restrict__ Kmat, float W, float H, float eps2d,
    const int32_t* __restrict__ radii, const float* __restrict__ conics,
    const float* __restrict__ compensations, const float* __restrict__ v_means2d,
    const float* __restrict__ v_depths, const float* __restrict__ v_conics,
    const float* __restrict__ v_compensations, float* __restrict__ v_means,
    float* __restrict__ v_quats, float* __restrict__ v_scales, float* __restrict__ v_viewmat) {
  int g = blockIdx.x * kBlock + threadIdx.x;
  bool active = g < n && radii[g] > 0;
  ProjectedGrad r;
  if (active) {
    CameraParams cam = load_camera(viewmat, Kmat);
    float m[3], s[3], q[4], con[3], vm2[2], vcon[3];
    load3(means + 3 * (size_t)g, m);
    load3(scales + 3 * (size_t)g, s);
    load3(conics + 3 * (size_t)g, con);
    load3(v_conics + 3 * (size_t)g, vcon);
    float4 qq = reinterpret_cast<const float4*>(quats)[g];
    q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
    float2 v2 = reinterpret_cast<const float2*>(v_means2d)[g];
    vm2[0] = v2.x; vm2[1] = v2.y;
    float comp = compensations ? compensations[g] : 0.f;
    float vcomp = (compensations && v_compensations) ? v_compensations[g] : 0.f;
    r = project_gaussian_vjp(m, q, s, cam, W, H, eps2d, con, comp, vm2, v_depths ? v_depths[g] : 0.f,
                             vcon, vcomp);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      v_means[3 * (size_t)g + k] += r.v_mean[k];
      v_scales[3 * (size_t)g + k] += r.v_scale[k];
    }
    float4* vq = reinterpret_cast<float4*>(v_quats) + g;
    float4 o = *vq;
    o.x += r.v_quat[0]; o.y += r.v_quat[1]; o.z += r.v_quat[2]; o.w += r.v_quat[3];
    *vq = o;
  }
  if (v_viewmat) reduce_viewmat(r, active, v_viewmat);
}

// Per-lane SH backward.  Coefficients come from `crow` (registers or global row pointer
// semantics hidden by the caller); v_coeff rows go to `vrow` (LDS row or global row).
template <int DEG, typename CoeffAt, typename StoreV>
__device__ __forceinline__ void sh_bwd_lane(const float dir[3], const float v_rgb[3],
                                            CoeffAt coeff_at, StoreV store_v, float v_dir[3]) {
  constexpr int KC = (DEG + 1) * (DEG + 1);
  float n2 = dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2];
  float inv = n2 > 0.f ? 1.0f / sqrtf(n2) : 0.f;
  float x = dir[0] * inv, y = dir[1] * inv, z = dir[2] * inv;
  float Y[KC], Yx[KC], Yy[KC], Yz[KC];
  sh_basis(DEG, x, y, z, Y);
  sh_basis_grad(DEG, x, y, z, Yx, Yy, Yz);
  float vx = 0.f, vy = 0.f, vz = 0.f;
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    // read the coefficients before storing: in the staged path both live in the same LDS row
    float s = coeff_at(3 * k + 0) * v_rgb[0] + coeff_at(3 * k + 1) * v_rgb[1] +
              coeff_at(3 * k + 2) * v_rgb[2];
    store_v(3 * k + 0, Y[k] * v_rgb[0]);
    store_v(3 * k + 1, Y[k] * v_rgb[1]);
    store_v(3 * k + 2, Y[k] * v_rgb[2]);
    vx += Yx[k] * s; vy += Yy[k] * s; vz += Yz[k] * s;
  }
  float d = vx * x + vy * y + vz * z;
  v_dir[0] = (vx - d * x) * inv;
  v_dir[1] = (vy - d * y) * inv;
  v_dir[2] = (vz - d * z) * inv;
}

// Shared body: given an active flag, direction and colour cotangent per lane, produce the
// v_coeffs row (written / accumulated) and v_dir.  STAGED <=> rows are 48 floats.
template <int DEG, bool STAGED, bool ACCUM>
__device__ __forceinline__ void sh_bwd_rows(int n, int stride_f, int g, bool active,
                                            const float dir[3], const float v_rgb[3],
                                            const float* __restrict__ coeffs,
                                            float* __restrict__ v_coeffs, float4* lds_wave,
                                            float v_dir[3]) {
  constexpr int KC = (DEG + 1) * (DEG + 1);
  v_dir[0] = v_dir[1] = v_dir[2] = 0.f;
  if constexpr (STAGED) {
    const unsigned lane = threadIdx.x & (kWave - 1);
    const int g0 = g - (int)lane;
    unsigned long long wave_mask = ballot(active);
    sh_rows_to_lds(coeffs, g0, n, wave_mask, lds_wave);
    __syncthreads();
    float* row = reinterpret_cast<float*>(lds_wave + lane * kShPitchF4);
    if (active) {
      sh_bwd_lane<DEG>(dir, v_rgb, [&](int i) { return row[i]; },
                       [&](int i, float v) { row[i] = v; }, v_dir);
#pragma unroll
      for (int i = KC * 3; i < 48; ++i) row[i] = 0.f;
    } else {
#pragma unroll
      for (int i = 0; i < 48; ++i) row[i] = 0.f;
    }
    __syncthreads();
    sh_rows_from_lds<ACCUM>(v_coeffs, g0, n, lds_wave);
  } else {
    if (g < n) {
      const float* crow = coeffs + (size_t)g * stride_f;
      float* vrow = v_coeffs + (size_t)g * stride_f;
      if (active) {
        sh_bwd_lane<DEG>(dir, v_rgb, [&](int i) { return crow[i]; },
                         [&](int i, float v) { vrow[i] = ACCUM ? vrow[i] + v : v; }, v_dir);
        if (!ACCUM)
          for (int i = KC * 3; i < stride_f; ++i) vrow[i] = 0.f;
      } else if (!ACCUM) {
        for (int i = 0; i < stride_f; ++i) vrow[i] = 0.f;
      }
    }
  }
}

template <int DEG, bool STAGED>
__global__ __launch_bounds__(kBlock) void sh_bwd_kernel(
    int n, int stride_f, const float* __restrict__ dirs, const float* __restrict__ coeffs,
    const uint8_t* __restrict__ masks, const float* __restrict__ v_colors,
    float* __restrict__ v_coeffs, float* __restrict__ v_dirs) {
  __shared__ float4 lds[STAGED ? (kBlock / kWave) * kWave * kShPitchF4 : 1];
  int g = blockIdx.x * kBlock + threadIdx.x;
  bool active = g < n && (masks == nullptr || masks[g] != 0);
  float d[3] = {0.f, 0.f, 1.f}, vc[3] = {0.f, 0.f, 0.f}, vd[3];
  if (active) {
    load3(dirs + 3 * (size_t)g, d);
    load3(v_colors + 3 * (size_t)g, vc);
  }
  sh_bwd_rows<DEG, STAGED, false>(n, stride_f, g, active, d, vc, coeffs, v_coeffs,
                                  lds + (threadIdx.x / kWave) * kWave * kShPitchF4, vd);
  if (g < n && v_dirs) {
    v_dirs[3 * (size_t)g + 0] = vd[0];
    v_dirs[3 * (size_t)g + 1] = vd[1];
    v_dirs[3 * (size_t)g + 2] = vd[2];
  }
}

template <int DEG, bool STAGED, bool ACCUM, bool VIEWGRAD>
__global__ __launch_bounds__(kBlock) void project_color_bwd_kernel(
    int n, const float* __restrict__ means, const float* __restrict__ quats,
    const float* __restrict__ scales, const float* __restrict__ opacities, int stride_f,
    const float* __restrict__ coeffs, const float* __restrict__ viewmat,
    const float* __restrict__ Kmat, float W, float H, float eps2d,
    const int32_t* __restrict__ radii, const float* __restrict__ conics, int antialiased,
    int feat_stride, const float* __restrict__ feats, const float* __restrict__ v_feats,
    const float* __restrict__ v_means2d, const float* __restrict__ v_conics,
    const float* __restrict__ v_depths, const float* __restrict__ v_opac_out,
    float* __restrict__ v_means, float* __restrict__ v_quats, float* __restrict__ v_scales,
    float* __restrict__ v_coeffs, float* __restrict__ v_opacities, float* __restrict__ v_viewmat) {
  __shared__ float4 lds[STAGED ? (kBlock / kWave) * kWave * kShPitchF4 : 1];
  int g = blockIdx.x * kBlock + threadIdx.x;
  bool active = g < n && radii[g] > 0;
  CameraParams cam = load_camera(viewmat, Kmat);
  float m[3] = {0.f, 0.f, 0.f}, dir[3] = {0.f, 0.f, 1.f}, v_rgb[3] = {0.f, 0.f, 0.f};
  float v_depth = 0.f;
  if (active) {
    load3(means + 3 * (size_t)g, m);
    float campos[3];
    camera_position(cam, campos);
    dir[0] = m[0] - campos[0]; dir[1] = m[1] - campos[1]; dir[2] = m[2] - campos[2];
    const float* f = feats + (size_t)g * feat_stride;
    const float* vf = v_feats + (size_t)g * feat_stride;
#pragma unroll
    for (int c = 0; c < 3; ++c) v_rgb[c] = f[c] > 0.f ? vf[c] : 0.f;   // clamp_min(x + 0.5, 0)
    if (feat_stride == 4) v_depth = vf[3];
    if (v_depths) v_depth += v_depths[g];
  }
  float v_dir[3];
  sh_bwd_rows<DEG, STAGED, ACCUM>(n, stride_f, g, active, dir, v_rgb, coeffs, v_coeffs,
                                  lds + (threadIdx.x / kWave) * kWave * kShPitchF4, v_dir);
  if (g >= n && !VIEWGRAD) return;
  float om[3] = {0.f, 0.f, 0.f}, os[3] = {0.f, 0.f, 0.f}, oq[4] = {0.f, 0.f, 0.f, 0.f};
  float v_opac = 0.f;
  ProjectedGrad r;
  if (active) {
    float s[3], q[4], con[3], vcon[3], vm2[2];
    load3(scales + 3 * (size_t)g, s);
    load3(conics + 3 * (size_t)g, con);
    load3(v_conics + 3 * (size_t)g, vcon);
    float4 qq = reinterpret_cast<const float4*>(quats)[g];
    q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
    float2 v2 = reinterpret_cast<const float2*>(v_means2d)[g];
    vm2[0] = v2.x; vm2[1] = v2.y;
    float comp = 0.f, v_comp = 0.f;
    if (antialiased && v_opac_out) {
      // compensation from the blurred conic: det(C)/det(C + eps I)
      float det_conic = con[0] * con[2] - con[1] * con[1];
      float inv_dc = 1.0f / det_conic;
      float a00 = con[2] * inv_dc - eps2d, a11 = con[0] * inv_dc - eps2d, a01 = -con[1] * inv_dc;
      comp = sqrtf(fmaxf(0.f, (a00 * a11 - a01 * a01) * det_conic));
      float vo = v_opac_out[g];
      v_comp = vo * opacities[g];
      v_opac = vo * comp;
    }
    r = project_gaussian_vjp(m, q, s, cam, W, H, eps2d, con, comp, vm2, v_depth, vcon, v_comp);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      om[k] = r.v_mean[k] + v_dir[k];
      os[k] = r.v_scale[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) oq[k] = r.v_quat[k];
  }
  if constexpr (VIEWGRAD) {
    // the view direction depends on the camera too: dir = mean - campos = mean + R^T t
    if (active) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          r.v_R[j * 3 + i] += v_dir[i] * cam.t[j];
          r.v_t[j] += cam.R[j * 3 + i] * v_dir[i];
        }
      }
    }
    reduce_viewmat(r, active, v_viewmat);
    if (g >= n) return;
  }
  float4* vq = reinterpret_cast<float4*>(v_quats) + g;
  if (ACCUM) {
    if (active) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        v_means[3 * (size_t)g + k] += om[k];
        v_scales[3 * (size_t)g + k] += os[k];
      }
      float4 o = *vq;
      o.x += oq[0]; o.y += oq[1]; o.z += oq[2]; o.w += oq[3];
      *vq = o;
      if (v_opacities && antialiased) v_opacities[g] += v_opac;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      v_means[3 * (size_t)g + k] = om[k];
      v_scales[3 * (size_t)g + k] = os[k];
    }
    *vq = make_float4(oq[0], oq[1], oq[2], oq[3]);
    if (v_opacities && antialiased) v_opacities[g] = v_opac;
  }
}

}  // namespace
}  // namespace mgs

using namespace mgs;

extern "C" int mgs_projection_bwd(int n, const float* means, const float* quats,
                                  const float* scales, const float* viewmat, const float* K,
                                  int width, int height, float eps2d, const int32_t* radii,
                                  const float* conics, const float* compensations,
                                  const float* v_means2d, const float* v_depths,
                                  const float* v_conics, const float* v_compensations,
                                  float* v_means, float* v_quats, float* v_scales,
                                  float* v_viewmat, mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && width > 0 && height > 0, "projection_bwd: bad sizes");
  if (n == 0) return MGS_OK;
  MGS_REQUIRE(means && quats && scales && viewmat && K && radii && conics && v_means2d &&
                  v_conics && v_means && v_quats && v_scales, "projection_bwd: null pointer");
  hipLaunchKernelGGL(projection_bwd_kernel, dim3(div_up(n, kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, n, means, quats, scales, viewmat, K, (float)width,
                     (float)height, eps2d, radii, conics, compensations, v_means2d, v_depths,
                     v_conics, v_compensations, v_means, v_quats, v_scales, v_viewmat);
  return check_launch("projection_bwd");
}

extern "C" int mgs_sh_bwd(int n, int degree, int coeff_stride, const float* dirs,
                          const float* coeffs, const uint8_t* masks, const float* v_colors,
                          float* v_coeffs, float* v_dirs, mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && degree >= 0 && degree <= 3, "sh_bwd: degree %d not in 0..3", degree);
  MGS_REQUIRE(coeff_stride >= (degree + 1) * (degree + 1), "sh_bwd: too few coefficients");
  if (n == 0) return MGS_OK;
  MGS_REQUIRE(dirs && coeffs && v_colors && v_coeffs, "sh_bwd: null pointer");
  dim3 grid(div_up(n, kBlock)), block(kBlock);
  hipStream_t s = (hipStream_t)stream;
  int sf = coeff_stride * 3;
  bool staged = coeff_stride == 16;
#define MGS_SHB(D, S) \
  hipLaunchKernelGGL((sh_bwd_kernel<D, S>), grid, block, 0, s, n, sf, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs)
  switch (degree) {
    case 0: if (staged) MGS_SHB(0, true); else MGS_SHB(0, false); break;
    case 1: if (staged) MGS_SHB(1, true); else MGS_SHB(1, false); break;
    case 2: if (staged) MGS_SHB(2, true); else MGS_SHB(2, false); break;
    default: if (staged) MGS_SHB(3, true); else MGS_SHB(3, false); break;
  }
#undef MGS_SHB
  return check_launch("sh_bwd");
}

extern "C" int mgs_project_color_bwd(int n, const float* means, const float* quats,
                                     const float* scales, const float* opacities, int sh_degree,
                                     int coeff_stride, const float* sh_coeffs,
                                     const float* viewmat, const float* K, int width, int height,
                                     float eps2d, const int32_t* radii, const float* conics,
                                     int antialiased, int feat_stride, const float* feats,
                                     const float* v_feats, const float* v_means2d,
                                     const float* v_conics, const float* v_depths,
                                     const float* v_opac_out, float* v_means, float* v_quats,
                                     float* v_scales, float* v_sh_coeffs, float* v_opacities,
                                     float* v_viewmat, int accumulate, mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && width > 0 && height > 0, "project_color_bwd: bad sizes");
  MGS_REQUIRE(sh_degree >= 0 && sh_degree <= 3, "project_color_bwd: sh_degree %d not in 0..3", sh_degree);
  MGS_REQUIRE(coeff_stride >= (sh_degree + 1) * (sh_degree + 1), "project_color_bwd: coeff_stride too small");
  MGS_REQUIRE(feat_stride == 3 || feat_stride == 4, "project_color_bwd: feat_stride must be 3 or 4");
  if (n == 0) return MGS_OK;
  MGS_REQUIRE(means && quats && scales && sh_coeffs && viewmat && K && radii && conics && feats &&
                  v_feats && v_means2d && v_conics && v_means && v_quats && v_scales && v_sh_coeffs,
              "project_color_bwd: null pointer");
  MGS_REQUIRE(!antialiased || (opacities && v_opac_out && v_opacities),
              "project_color_bwd: antialiased needs opacities, v_opac_out, v_opacities");
  dim3 grid(div_up(n, kBlock)), block(kBlock);
  hipStream_t s = (hipStream_t)stream;
  int sf = coeff_stride * 3;
  bool staged = coeff_stride == 16;
#define MGS_PCB(D, S, A)                                                                        \
  if (v_viewmat) MGS_PCB_V(D, S, A, true); else MGS_PCB_V(D, S, A, false)
#define MGS_PCB_V(D, S, A, V)                                                                   \
  hipLaunchKernelGGL((project_color_bwd_kernel<D, S, A, V>), grid, block, 0, s, n, means, quats, \
                     scales, opacities, sf, sh_coeffs, viewmat, K, (float)width, (float)height, \
                     eps2d, radii, conics, antialiased, feat_stride, feats, v_feats, v_means2d, \
                     v_conics, v_depths, v_opac_out, v_means, v_quats, v_scales, v_sh_coeffs,   \
                     v_opacities, v_viewmat)
#define MGS_PCB_D(D)                                                  \
  if (staged) { if (accumulate) { MGS_PCB(D, true, true); } else { MGS_PCB(D, true, false); } } \
  else { if (accumulate) { MGS_PCB(D, false, true); } else { MGS_PCB(D, false, false); } }
  switch (sh_degree) {
    case 0: MGS_PCB_D(0) break;
    case 1: MGS_PCB_D(1) break;
    case 2: MGS_PCB_D(2) break;
    default: MGS_PCB_D(3) break;
  }
#undef MGS_PCB_D
#undef MGS_PCB
#undef MGS_PCB_V
  return check_launch("project_color_bwd");
}
