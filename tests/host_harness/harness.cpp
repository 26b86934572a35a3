// Host build (g++) of the per-Gaussian device math in robosimgs_amd/csrc/mgs_math.h, so its
// logic can be checked against the oracle without a GPU.  Test-only: never linked into
// libmgs.so and never reachable from the product package.
#include "../../robosimgs_amd/csrc/mgs_math.h"

using namespace mgs;

extern "C" void hh_project(int n, const float* means, const float* quats, const float* scales,
                           const float* viewmat, const float* K, int W, int H, float eps2d,
                           float near_plane, float far_plane, float radius_clip, int* radii,
                           float* means2d, float* depths, float* conics, float* comps) {
  CameraParams cam = load_camera(viewmat, K);
  for (int g = 0; g < n; ++g) {
    Projected p = project_gaussian(means + 3 * g, quats + 4 * g, scales + 3 * g, cam, (float)W,
                                   (float)H, eps2d, near_plane, far_plane, radius_clip);
    radii[g] = p.radius;
    means2d[2 * g] = p.mean2d[0]; means2d[2 * g + 1] = p.mean2d[1];
    depths[g] = p.depth;
    for (int k = 0; k < 3; ++k) conics[3 * g + k] = p.conic[k];
    comps[g] = p.compensation;
  }
}

// the per-axis, opacity-aware radius rule (mgs_math.h MGS_RADIUS_OPACITY_AWARE); opacities nullable
extern "C" void hh_project_rule(int n, const float* means, const float* quats, const float* scales,
                                const float* viewmat, const float* K, int W, int H, float eps2d,
                                float near_plane, float far_plane, float radius_clip, const float* opacities,
                                int antialiased, int* radii_x, int* radii_y, float* means2d) {
  CameraParams cam = load_camera(viewmat, K);
  for (int g = 0; g < n; ++g) {
    Projected p = project_gaussian(means + 3 * g, quats + 4 * g, scales + 3 * g, cam, (float)W, (float)H, eps2d,
                                   near_plane, far_plane, radius_clip, MGS_RADIUS_OPACITY_AWARE, opacities != nullptr,
                                   opacities ? opacities[g] : 1.f, antialiased != 0);
    radii_x[g] = p.radius;
    radii_y[g] = p.radius_y;
    means2d[2 * g] = p.mean2d[0]; means2d[2 * g + 1] = p.mean2d[1];
  }
}

extern "C" void hh_project_vjp(int n, const float* means, const float* quats, const float* scales,
                               const float* viewmat, const float* K, int W, int H, float eps2d,
                               const int* radii, const float* conics, const float* comps,
                               const float* v_means2d, const float* v_depths,
                               const float* v_conics, const float* v_comps, float* v_means,
                               float* v_quats, float* v_scales, float* v_R, float* v_t) {
  CameraParams cam = load_camera(viewmat, K);
  for (int k = 0; k < 9; ++k) v_R[k] = 0.f;
  for (int k = 0; k < 3; ++k) v_t[k] = 0.f;
  for (int g = 0; g < n; ++g) {
    for (int k = 0; k < 3; ++k) v_means[3 * g + k] = v_scales[3 * g + k] = 0.f;
    for (int k = 0; k < 4; ++k) v_quats[4 * g + k] = 0.f;
    if (radii[g] <= 0) continue;
    ProjectedGrad r = project_gaussian_vjp(means + 3 * g, quats + 4 * g, scales + 3 * g, cam,
                                           (float)W, (float)H, eps2d, conics + 3 * g, comps[g],
                                           v_means2d + 2 * g, v_depths[g], v_conics + 3 * g,
                                           v_comps ? v_comps[g] : 0.f);
    for (int k = 0; k < 3; ++k) { v_means[3 * g + k] = r.v_mean[k]; v_scales[3 * g + k] = r.v_scale[k]; }
    for (int k = 0; k < 4; ++k) v_quats[4 * g + k] = r.v_quat[k];
    for (int k = 0; k < 9; ++k) v_R[k] += r.v_R[k];
    for (int k = 0; k < 3; ++k) v_t[k] += r.v_t[k];
  }
}

template <int DEG>
static void sh_all(int n, int stride, const float* dirs, const float* coeffs, const float* v_rgb,
                   float* colors, float* v_coeffs, float* v_dirs) {
  constexpr int KC = (DEG + 1) * (DEG + 1);
  for (int g = 0; g < n; ++g) {
    const float* d = dirs + 3 * g;
    float n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    float inv = 1.0f / sqrtf(n2);
    float Y[KC];
    sh_basis(DEG, d[0] * inv, d[1] * inv, d[2] * inv, Y);
    for (int c = 0; c < 3; ++c) {
      float s = 0.f;
      for (int k = 0; k < KC; ++k) s += Y[k] * coeffs[(size_t)g * stride * 3 + 3 * k + c];
      colors[3 * g + c] = s;
    }
    float vc[KC * 3];
    sh_vjp<DEG>(d, coeffs + (size_t)g * stride * 3, v_rgb + 3 * g, vc, v_dirs + 3 * g);
    for (int k = 0; k < stride * 3; ++k) v_coeffs[(size_t)g * stride * 3 + k] = k < KC * 3 ? vc[k] : 0.f;
  }
}

extern "C" void hh_sh(int n, int degree, int stride, const float* dirs, const float* coeffs,
                      const float* v_rgb, float* colors, float* v_coeffs, float* v_dirs) {
  switch (degree) {
    case 0: sh_all<0>(n, stride, dirs, coeffs, v_rgb, colors, v_coeffs, v_dirs); break;
    case 1: sh_all<1>(n, stride, dirs, coeffs, v_rgb, colors, v_coeffs, v_dirs); break;
    case 2: sh_all<2>(n, stride, dirs, coeffs, v_rgb, colors, v_coeffs, v_dirs); break;
    default: sh_all<3>(n, stride, dirs, coeffs, v_rgb, colors, v_coeffs, v_dirs); break;
  }
}
