// transform.hip -- row (f3) on the device: similarity transforms of Gaussian groups, for gfx950.
// "World Coordinate Frame Alignment" (/root/reference/README.md:54-55) once per scene and, more
// to the point for a robot simulator, rigid motion of the Gaussians attached to articulated parts
// every frame: each Gaussian carries a group id, each group a transform x -> s R x + t.
//   means'  = (s R) p + t        scales' = s * scales        quats' = q_R (x) q   (Hamilton, wxyz)
//   SH: the coefficients of degree l are multiplied by the (2l+1)x(2l+1) real-SH rotation matrix
//   of R (computed on the host, gaussians.py:sh_rotation_matrices), so that the view-dependent
//   colour turns with the part.
// HBM-bound streaming: 44 B in / 44 B out per Gaussian, plus 192 B in / out for the SH rows of
// MOVING groups only, staged through LDS as in projection.hip (sh_staging.h) when rows hold 16
// coefficients (shorter rows are read per lane).  Group id -1 (or an
// identity group) passes a Gaussian through unchanged.
#include "mgs_common.h"
#include "sh_staging.h"

namespace mgs {
namespace {

constexpr int kBlock = 64;            // one wave per workgroup (see projection.hip)
constexpr int kXformFloats = 20;      // M[9] (= s R, row-major), t[3], q_R[4] (wxyz), s, pad[3]
constexpr int kShRotFloats = 84;      // 3x3 + 5x5 + 7x7 (+1 pad), row-major, degree 1..3

template <bool SH, bool STAGED>
__global__ __launch_bounds__(kBlock) void transform_kernel(
    int n, const float* __restrict__ means, const float* __restrict__ quats,
    const float* __restrict__ scales, int sh_degree, int stride_f, const float* __restrict__ sh_in,
    const int32_t* __restrict__ group_ids, int n_groups, const float* __restrict__ xforms,
    const float* __restrict__ sh_rot, float* __restrict__ out_means, float* __restrict__ out_quats,
    float* __restrict__ out_scales, float* __restrict__ sh_out) {
  __shared__ float4 lds[(SH && STAGED) ? kShWaveSlots : 1];
  const int g = blockIdx.x * kBlock + threadIdx.x;
  const unsigned lane = threadIdx.x;
  int gid = -1;
  if (g < n) gid = group_ids ? group_ids[g] : 0;
  const bool moving = gid >= 0 && gid < n_groups;
  if (g < n) {
    float p[3] = {means[3 * (size_t)g], means[3 * (size_t)g + 1], means[3 * (size_t)g + 2]};
    float4 q = reinterpret_cast<const float4*>(quats)[g];
    float s3[3] = {scales[3 * (size_t)g], scales[3 * (size_t)g + 1], scales[3 * (size_t)g + 2]};
    if (moving) {
      const float* X = xforms + (size_t)gid * kXformFloats;
      float o[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) o[r] = X[3 * r] * p[0] + X[3 * r + 1] * p[1] + X[3 * r + 2] * p[2] + X[9 + r];
      const float aw = X[12], ax = X[13], ay = X[14], az = X[15], s = X[16];
      const float4 b = q;   // (w, x, y, z)
      q = make_float4(aw * b.x - ax * b.y - ay * b.z - az * b.w, aw * b.y + ax * b.x + ay * b.w - az * b.z,
                      aw * b.z - ax * b.w + ay * b.x + az * b.y, aw * b.w + ax * b.z - ay * b.y + az * b.x);
#pragma unroll
      for (int r = 0; r < 3; ++r) { p[r] = o[r]; s3[r] *= s; }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      out_means[3 * (size_t)g + r] = p[r];
      out_scales[3 * (size_t)g + r] = s3[r];
    }
    reinterpret_cast<float4*>(out_quats)[g] = q;
  }
  if constexpr (SH) {
    // rows of moving Gaussians are fetched, rotated and written; when the output is a different
    // buffer the other rows are copied through (wave_mask = all rows)
    const bool copy_all = sh_out != sh_in;
    const unsigned long long need = copy_all ? ballot(g < n) : ballot(moving);
    if (need == 0ull) return;
    const int g0 = blockIdx.x * kBlock;
    const int kc = (sh_degree + 1) * (sh_degree + 1) - 1;      // coefficients above the DC term
    float* row;                                                  // this lane's 16 x rgb (or shorter) row
    if constexpr (STAGED) {
      sh_rows_to_lds(sh_in, g0, n, need, lds);
      __syncthreads();
      row = reinterpret_cast<float*>(lds + lane * kShPitchF4);
    } else {
      row = const_cast<float*>(sh_in) + (size_t)(g < n ? g : 0) * stride_f;
    }
    float c[15 * 3], o[15 * 3];
#pragma unroll
    for (int i = 0; i < 45; ++i) c[i] = (g < n && i < kc * 3) ? row[3 + i] : 0.f;
#pragma unroll
    for (int i = 0; i < 45; ++i) o[i] = c[i];
    if (moving && sh_degree >= 1) {
      const float* M = sh_rot + (size_t)gid * kShRotFloats;
      // degree 1: coefficients 1..3 (c[0..8]), degree 2: 4..8 (c[9..23]), degree 3: 9..15 (c[24..44])
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
          o[3 * k + ch] = M[3 * k] * c[ch] + M[3 * k + 1] * c[3 + ch] + M[3 * k + 2] * c[6 + ch];
      if (sh_degree >= 2) {
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 5; ++j) acc = fmaf(M[9 + 5 * k + j], c[9 + 3 * j + ch], acc);
            o[9 + 3 * k + ch] = acc;
          }
      }
      if (sh_degree >= 3) {
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 7; ++j) acc = fmaf(M[34 + 7 * k + j], c[24 + 3 * j + ch], acc);
            o[24 + 3 * k + ch] = acc;
          }
      }
    }
    if constexpr (STAGED) {
      if (moving) {
#pragma unroll
        for (int i = 0; i < 45; ++i)
          if (i < kc * 3) row[3 + i] = o[i];
      }
      __syncthreads();
      // lane-linear write-back; rows that were not fetched (in place, static Gaussian) are skipped
      float4* dst = reinterpret_cast<float4*>(sh_out + (size_t)g0 * 48);
#pragma unroll
      for (int m = 0; m < kShRowF4; ++m) {
        unsigned f = m * kShWave + lane, owner = f / kShRowF4;
        if (((need >> owner) & 1ull) && g0 + (int)owner < n)
          dst[f] = lds[owner * kShPitchF4 + (f - owner * kShRowF4)];
      }
    } else if (g < n && (copy_all || moving)) {
      float* dst = sh_out + (size_t)g * stride_f;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) dst[ch] = row[ch];          // DC term
#pragma unroll
      for (int i = 0; i < 45; ++i)
        if (i < kc * 3) dst[3 + i] = o[i];
      for (int i = 3 + kc * 3; i < stride_f; ++i) dst[i] = row[i];   // coefficients above the active degree
    }
  }
}

}  // namespace
}  // namespace mgs

using namespace mgs;

extern "C" int mgs_transform_gaussians(int n, const float* means, const float* quats,
                                       const float* scales, int sh_degree, int coeff_stride,
                                       const float* sh_coeffs, const int32_t* group_ids,
                                       int n_groups, const float* xforms, const float* sh_rot,
                                       float* out_means, float* out_quats, float* out_scales,
                                       float* out_sh, mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && n_groups >= 1, "transform_gaussians: bad sizes");
  if (n == 0) return MGS_OK;
  MGS_REQUIRE(means && quats && scales && xforms && out_means && out_quats && out_scales,
              "transform_gaussians: null pointer");
  MGS_REQUIRE((sh_coeffs == nullptr) == (out_sh == nullptr), "transform_gaussians: sh_coeffs and out_sh go together");
  MGS_REQUIRE(!sh_coeffs || (sh_degree >= 0 && sh_degree <= 3 && coeff_stride >= (sh_degree + 1) * (sh_degree + 1)),
              "transform_gaussians: SH rows [N,coeff_stride,3] must hold (degree+1)^2 coefficients, degree 0..3");
  MGS_REQUIRE(!sh_coeffs || sh_degree == 0 || sh_rot, "transform_gaussians: degree >= 1 needs sh_rot");
  dim3 grid(div_up(n, kBlock)), block(kBlock);
  hipStream_t s = (hipStream_t)stream;
  const int stride_f = coeff_stride * 3;
  if (sh_coeffs && coeff_stride == 16)
    hipLaunchKernelGGL((transform_kernel<true, true>), grid, block, 0, s, n, means, quats, scales, sh_degree,
                       stride_f, sh_coeffs, group_ids, n_groups, xforms, sh_rot, out_means, out_quats, out_scales, out_sh);
  else if (sh_coeffs)
    hipLaunchKernelGGL((transform_kernel<true, false>), grid, block, 0, s, n, means, quats, scales, sh_degree,
                       stride_f, sh_coeffs, group_ids, n_groups, xforms, sh_rot, out_means, out_quats, out_scales, out_sh);
  else
    hipLaunchKernelGGL((transform_kernel<false, false>), grid, block, 0, s, n, means, quats, scales, 0, 0,
                       (const float*)nullptr, group_ids, n_groups, xforms, (const float*)nullptr,
                       out_means, out_quats, out_scales, (float*)nullptr);
  return check_launch("transform_gaussians");
}
