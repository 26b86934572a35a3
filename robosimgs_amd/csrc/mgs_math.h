// mgs_math.h -- per-Gaussian math of the projection / colour stage, forward and backward.
//
// Plain scalar functions, usable from HIP device code and (for the logic tests under
// tests/host_harness) from a host compiler: MGS_HD expands to `__host__ __device__` under
// hipcc and to nothing under g++.  Semantics: SURVEY.md Appendix A.2 steps 1-6 (the
// gsplat 1.x "classic" projection the reference's Nerfstudio dependency uses; the
// reference itself holds no implementation, README.md:75).
#ifndef MGS_MATH_H_
#define MGS_MATH_H_

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MGS_HD __host__ __device__ __forceinline__
#else
#define MGS_HD inline
#endif

namespace mgs {

struct CameraParams {  // one camera, row-major viewmat (OpenCV world-to-camera)
  float R[9];
  float t[3];
  float fx, fy, cx, cy;
};

MGS_HD CameraParams load_camera(const float* viewmat, const float* K) {
  CameraParams c;
  c.R[0] = viewmat[0]; c.R[1] = viewmat[1]; c.R[2] = viewmat[2];  c.t[0] = viewmat[3];
  c.R[3] = viewmat[4]; c.R[4] = viewmat[5]; c.R[5] = viewmat[6];  c.t[1] = viewmat[7];
  c.R[6] = viewmat[8]; c.R[7] = viewmat[9]; c.R[8] = viewmat[10]; c.t[2] = viewmat[11];
  c.fx = K[0]; c.cx = K[2]; c.fy = K[4]; c.cy = K[5];
  return c;
}

// camera centre in world space: -R^T t
MGS_HD void camera_position(const CameraParams& c, float pos[3]) {
  pos[0] = -(c.R[0] * c.t[0] + c.R[3] * c.t[1] + c.R[6] * c.t[2]);
  pos[1] = -(c.R[1] * c.t[0] + c.R[4] * c.t[1] + c.R[7] * c.t[2]);
  pos[2] = -(c.R[2] * c.t[0] + c.R[5] * c.t[1] + c.R[8] * c.t[2]);
}

#ifndef MGS_PROJ_FACTORED
// project_gaussian: 0 = cov2d in the textbook order of A.2 steps 2-4 (the reference's, and what every parity statement is made
// on); 1 = from the 2 x 3 factor J R Rq S (measurement, MGS_EXTRA_FLAGS=-DMGS_PROJ_FACTORED=1: profiles/r6/00_experiments.md 9)
#define MGS_PROJ_FACTORED 0
#endif

// a b - c d with one rounding's worth of error (Kahan): the product c d is rounded, its error recovered by an FMA
MGS_HD float diff_of_products(float a, float b, float c, float d) {
  const float w = c * d;
  const float e = fmaf(-c, d, w);     // w - c d exactly
  const float f = fmaf(a, b, -w);
  return f + e;
}

// C = A * B, all 3x3 row-major
MGS_HD void mat3_mul(const float* A, const float* B, float* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      C[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] +
                     A[i * 3 + 2] * B[2 * 3 + j];
}
// C = A * B^T
MGS_HD void mat3_mul_bt(const float* A, const float* B, float* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      C[i * 3 + j] = A[i * 3 + 0] * B[j * 3 + 0] + A[i * 3 + 1] * B[j * 3 + 1] +
                     A[i * 3 + 2] * B[j * 3 + 2];
}
// C = A^T * B
MGS_HD void mat3_mul_at(const float* A, const float* B, float* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      C[i * 3 + j] = A[0 * 3 + i] * B[0 * 3 + j] + A[1 * 3 + i] * B[1 * 3 + j] +
                     A[2 * 3 + i] * B[2 * 3 + j];
}

// A.2 step 1: normalised wxyz quaternion -> rotation
MGS_HD void quat_to_rotmat(const float q[4], float R[9]) {
  float inv = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
  float x2 = x * x, y2 = y * y, z2 = z * z;
  float xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  R[0] = 1.f - 2.f * (y2 + z2); R[1] = 2.f * (xy - wz);       R[2] = 2.f * (xz + wy);
  R[3] = 2.f * (xy + wz);       R[4] = 1.f - 2.f * (x2 + z2); R[5] = 2.f * (yz - wx);
  R[6] = 2.f * (xz - wy);       R[7] = 2.f * (yz + wx);       R[8] = 1.f - 2.f * (x2 + y2);
}

struct Projected {
  float mean2d[2];
  float depth;
  float conic[3];
  float compensation;
  int radius;    // 0 = culled; the extent along x under the per-axis rule
  int radius_y;  // == radius under the classic rule
};

// The radius rule is a compile-time policy: every kernel passes it as a template constant.
//   MGS_RADIUS_CLASSIC        gsplat 1.4 (A.2 step 5): one radius ceil(3 sqrt(lambda_1)), the square mean +- radius
//   MGS_RADIUS_OPACITY_AWARE  gsplat >= 1.5: per-axis extents ceil(e sqrt(Sigma_xx)), ceil(e sqrt(Sigma_yy)) with
//                             e = min(3.33, sqrt(2 ln(255 opacity))) -- the bounding box of the alpha >= 1/255 ellipse,
//                             capped at 3.33 sigma; Gaussians of opacity < 1/255 are culled; opacity is multiplied by
//                             the compensation in "antialiased" mode; without opacities e = 3.33
#define MGS_RADIUS_CLASSIC 0
#define MGS_RADIUS_OPACITY_AWARE 1

// A.2 steps 1-5.  Returns radius == 0 for culled Gaussians (all other fields zeroed).
MGS_HD Projected project_gaussian(const float mean[3], const float quat[4],
                                  const float scale[3], const CameraParams& cam, float W,
                                  float H, float eps2d, float near_plane, float far_plane,
                                  float radius_clip, int radius_rule = MGS_RADIUS_CLASSIC,
                                  bool has_opacity = false, float opacity = 1.f, bool antialiased = false) {
  Projected out;
  out.mean2d[0] = out.mean2d[1] = out.depth = 0.f;
  out.conic[0] = out.conic[1] = out.conic[2] = 0.f;
  out.compensation = 0.f;
  out.radius = 0;
  out.radius_y = 0;

  const float* R = cam.R;
  float x = R[0] * mean[0] + R[1] * mean[1] + R[2] * mean[2] + cam.t[0];
  float y = R[3] * mean[0] + R[4] * mean[1] + R[5] * mean[2] + cam.t[1];
  float z = R[6] * mean[0] + R[7] * mean[1] + R[8] * mean[2] + cam.t[2];
  if (!(z >= near_plane) || !(z <= far_plane)) return out;

  float rz = 1.0f / z, rz2 = rz * rz;
  float tanx = 0.5f * W / cam.fx, tany = 0.5f * H / cam.fy;
  float lim_xp = (W - cam.cx) / cam.fx + 0.3f * tanx, lim_xn = cam.cx / cam.fx + 0.3f * tanx;
  float lim_yp = (H - cam.cy) / cam.fy + 0.3f * tany, lim_yn = cam.cy / cam.fy + 0.3f * tany;
  float tx = z * fminf(lim_xp, fmaxf(-lim_xn, x * rz));
  float ty = z * fminf(lim_yp, fmaxf(-lim_yn, y * rz));
  float j00 = cam.fx * rz, j02 = -cam.fx * tx * rz2;
  float j11 = cam.fy * rz, j12 = -cam.fy * ty * rz2;
  float Rq[9];
  quat_to_rotmat(quat, Rq);
#if MGS_PROJ_FACTORED
  // cov2d = M2 M2^T with M2 = J R Rq diag(scale), a 2 x 3 matrix -- the same matrix as J (R (Rq S S Rq^T) R^T) J^T of A.2 steps
  // 2-4 without ever forming a covariance: a, c are sums of squares and det(cov2d) is the sum of M2's squared 2 x 2 minors
  // (Cauchy-Binet).  The textbook order's det = a c - b^2 cancels for a needle-like Gaussian (relative conic error up to 5e-3
  // at 300 : 1, this order's 2e-6) -- but that error scales the WHOLE conic, which the blend forgives; what sigma = 1/2 d^T conic d
  // does not forgive is independent noise in the three entries, and there the two orders differ by a factor of two
  // (profiles/r6/00_experiments.md section 9: half the pixels over tolerance over 59 clustered scenes, worse on 10 of them).
  float RR[9];
  mat3_mul(R, Rq, RR);
  float m0[3], m1[3];
  for (int k = 0; k < 3; ++k) {
    m0[k] = (j00 * RR[k] + j02 * RR[6 + k]) * scale[k];
    m1[k] = (j11 * RR[3 + k] + j12 * RR[6 + k]) * scale[k];
  }
  float a = m0[0] * m0[0] + m0[1] * m0[1] + m0[2] * m0[2];
  float b = m0[0] * m1[0] + m0[1] * m1[1] + m0[2] * m1[2];
  float c = m1[0] * m1[0] + m1[1] * m1[1] + m1[2] * m1[2];
  const float d01 = diff_of_products(m0[0], m1[1], m0[1], m1[0]);
  const float d02 = diff_of_products(m0[0], m1[2], m0[2], m1[0]);
  const float d12 = diff_of_products(m0[1], m1[2], m0[2], m1[1]);
  float det0 = d01 * d01 + d02 * d02 + d12 * d12;
  float det = det0 + eps2d * (a + c) + eps2d * eps2d;       // det(cov2d + eps2d I): positive terms only
  a += eps2d;
  c += eps2d;
#else
  float M[9], cov[9], tmp[9], covc[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[i * 3 + j] = Rq[i * 3 + j] * scale[j];
  mat3_mul_bt(M, M, cov);       // Sigma = M M^T
  mat3_mul(R, cov, tmp);        // R Sigma
  mat3_mul_bt(tmp, R, covc);    // R Sigma R^T
  // cov2d = J covc J^T with J = [[j00,0,j02],[0,j11,j12]]
  float a = j00 * (j00 * covc[0] + j02 * covc[6]) + j02 * (j00 * covc[2] + j02 * covc[8]);
  float b = j00 * (j11 * covc[1] + j12 * covc[2]) + j02 * (j11 * covc[7] + j12 * covc[8]);
  float c = j11 * (j11 * covc[4] + j12 * covc[7]) + j12 * (j11 * covc[5] + j12 * covc[8]);
  float det0 = a * c - b * b;
  a += eps2d;
  c += eps2d;
  float det = a * c - b * b;
#endif
  float mx = cam.fx * x * rz + cam.cx, my = cam.fy * y * rz + cam.cy;

  if (!(det > 0.f)) return out;
  float inv_det = 1.0f / det;

  float radius, radius_y;
  if (radius_rule == MGS_RADIUS_OPACITY_AWARE) {
    float extent = 3.33f;
    if (has_opacity) {
      const float op = antialiased ? opacity * sqrtf(fmaxf(0.f, det0 * inv_det)) : opacity;
      if (!(op >= 1.0f / 255.0f)) return out;
      extent = fminf(extent, sqrtf(2.f * logf(op * 255.0f)));
    }
    radius = ceilf(extent * sqrtf(a));
    radius_y = ceilf(extent * sqrtf(c));
    if (!(radius > radius_clip) && !(radius_y > radius_clip)) return out;
    if (!(radius > 0.f) || !(radius_y > 0.f)) return out;            // an extent of exactly zero reaches no pixel
  } else {
    float m = 0.5f * (a + c);
    float lam = m + sqrtf(fmaxf(0.01f, m * m - det));
    radius = ceilf(3.f * sqrtf(lam));
    radius_y = radius;
    if (!(radius > radius_clip)) return out;
  }
  if (mx + radius <= 0.f || mx - radius >= W || my + radius_y <= 0.f || my - radius_y >= H)
    return out;

  out.mean2d[0] = mx;
  out.mean2d[1] = my;
  out.depth = z;
  out.conic[0] = c * inv_det;
  out.conic[1] = -b * inv_det;
  out.conic[2] = a * inv_det;
  out.compensation = sqrtf(fmaxf(0.f, det0 * inv_det));
  out.radius = (int)radius;
  out.radius_y = (int)radius_y;
  return out;
}

// ---- A.2 step 6: real SH basis (gsplat ordering / signs) --------------------------------
#define MGS_SH_C0 0.2820947917738781f
#define MGS_SH_C1 0.48860251190292f

// Y[0..(deg+1)^2) for a UNIT direction.
MGS_HD void sh_basis(int deg, float x, float y, float z, float* Y) {
  Y[0] = MGS_SH_C0;
  if (deg < 1) return;
  Y[1] = -MGS_SH_C1 * y;
  Y[2] = MGS_SH_C1 * z;
  Y[3] = -MGS_SH_C1 * x;
  if (deg < 2) return;
  float z2 = z * z, fC1 = x * x - y * y, fS1 = 2.f * x * y;
  float t = -1.092548430592079f * z;
  Y[4] = 0.5462742152960395f * fS1;
  Y[5] = t * y;
  Y[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
  Y[7] = t * x;
  Y[8] = 0.5462742152960395f * fC1;
  if (deg < 3) return;
  float u = -2.285228997322329f * z2 + 0.4570457994644658f;
  float w = 1.445305721320277f * z;
  float fC2 = x * fC1 - y * fS1, fS2 = x * fS1 + y * fC1;
  Y[9] = -0.5900435899266435f * fS2;
  Y[10] = w * fS1;
  Y[11] = u * y;
  Y[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
  Y[13] = u * x;
  Y[14] = w * fC1;
  Y[15] = -0.5900435899266435f * fC2;
}

// dY/dx, dY/dy, dY/dz of the polynomial basis above (treating x,y,z as independent).
MGS_HD void sh_basis_grad(int deg, float x, float y, float z, float* Yx, float* Yy, float* Yz) {
  Yx[0] = Yy[0] = Yz[0] = 0.f;
  if (deg < 1) return;
  Yx[1] = 0.f;        Yy[1] = -MGS_SH_C1; Yz[1] = 0.f;
  Yx[2] = 0.f;        Yy[2] = 0.f;        Yz[2] = MGS_SH_C1;
  Yx[3] = -MGS_SH_C1; Yy[3] = 0.f;        Yz[3] = 0.f;
  if (deg < 2) return;
  const float c2a = 0.5462742152960395f, c2b = -1.092548430592079f, c2c = 0.9461746957575601f;
  float z2 = z * z, fC1 = x * x - y * y, fS1 = 2.f * x * y;
  // fS1_x = 2y, fS1_y = 2x; fC1_x = 2x, fC1_y = -2y
  Yx[4] = c2a * 2.f * y; Yy[4] = c2a * 2.f * x; Yz[4] = 0.f;
  Yx[5] = 0.f;           Yy[5] = c2b * z;       Yz[5] = c2b * y;
  Yx[6] = 0.f;           Yy[6] = 0.f;           Yz[6] = c2c * 2.f * z;
  Yx[7] = c2b * z;       Yy[7] = 0.f;           Yz[7] = c2b * x;
  Yx[8] = c2a * 2.f * x; Yy[8] = -c2a * 2.f * y; Yz[8] = 0.f;
  if (deg < 3) return;
  const float c3a = -0.5900435899266435f, c3w = 1.445305721320277f;
  const float c3u1 = -2.285228997322329f, c3u0 = 0.4570457994644658f;
  const float c3z1 = 1.865881662950577f, c3z0 = 1.119528997770346f;
  float u = c3u1 * z2 + c3u0, w = c3w * z;
  float u_z = c3u1 * 2.f * z;
  // fC2 = x fC1 - y fS1 ; fS2 = x fS1 + y fC1
  float fC2_x = fC1 + x * 2.f * x - y * 2.f * y;   // = 3x^2 - 3y^2
  float fC2_y = x * (-2.f * y) - fS1 - y * 2.f * x; // = -6xy
  float fS2_x = fS1 + x * 2.f * y + y * 2.f * x;   // = 6xy
  float fS2_y = x * 2.f * x + fC1 + y * (-2.f * y); // = 3x^2 - 3y^2
  Yx[9] = c3a * fS2_x;   Yy[9] = c3a * fS2_y;   Yz[9] = 0.f;
  Yx[10] = w * 2.f * y;  Yy[10] = w * 2.f * x;  Yz[10] = c3w * fS1;
  Yx[11] = 0.f;          Yy[11] = u;            Yz[11] = u_z * y;
  Yx[12] = 0.f;          Yy[12] = 0.f;          Yz[12] = (c3z1 * z2 - c3z0) + z * c3z1 * 2.f * z;
  Yx[13] = u;            Yy[13] = 0.f;          Yz[13] = u_z * x;
  Yx[14] = w * 2.f * x;  Yy[14] = -w * 2.f * y; Yz[14] = c3w * fC1;
  Yx[15] = c3a * fC2_x;  Yy[15] = c3a * fC2_y;  Yz[15] = 0.f;
}


// =========================================================================================
// Backward (vector-Jacobian products) of the functions above
// =========================================================================================

// v_q (un-normalised wxyz) from the cotangent G (row-major 3x3) of R(q/|q|)
MGS_HD void quat_to_rotmat_vjp(const float q[4], const float G[9], float v_q[4]) {
  float inv = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
  float vn[4];
  vn[0] = 2.f * (x * (G[7] - G[5]) + y * (G[2] - G[6]) + z * (G[3] - G[1]));
  vn[1] = 2.f * (y * (G[1] + G[3]) + z * (G[2] + G[6]) + w * (G[7] - G[5]) - 2.f * x * (G[4] + G[8]));
  vn[2] = 2.f * (x * (G[1] + G[3]) + z * (G[5] + G[7]) + w * (G[2] - G[6]) - 2.f * y * (G[0] + G[8]));
  vn[3] = 2.f * (x * (G[2] + G[6]) + y * (G[5] + G[7]) + w * (G[3] - G[1]) - 2.f * z * (G[0] + G[4]));
  float d = vn[0] * w + vn[1] * x + vn[2] * y + vn[3] * z;
  v_q[0] = (vn[0] - d * w) * inv;
  v_q[1] = (vn[1] - d * x) * inv;
  v_q[2] = (vn[2] - d * y) * inv;
  v_q[3] = (vn[3] - d * z) * inv;
}

struct ProjectedGrad {
  float v_mean[3];
  float v_quat[4];
  float v_scale[3];
  float v_R[9];   // d/d viewmat rotation (row-major)
  float v_t[3];   // d/d viewmat translation
};

// VJP of project_gaussian for a VISIBLE Gaussian (radius > 0).  `conic` is the forward
// output; v_comp is the cotangent of the compensation factor (0 unless antialiased).
MGS_HD ProjectedGrad project_gaussian_vjp(const float mean[3], const float quat[4],
                                          const float scale[3], const CameraParams& cam, float W,
                                          float H, float eps2d, const float conic[3],
                                          float compensation, const float v_mean2d[2],
                                          float v_depth, const float v_conic[3], float v_comp) {
  ProjectedGrad g;
  const float* R = cam.R;
  float x = R[0] * mean[0] + R[1] * mean[1] + R[2] * mean[2] + cam.t[0];
  float y = R[3] * mean[0] + R[4] * mean[1] + R[5] * mean[2] + cam.t[1];
  float z = R[6] * mean[0] + R[7] * mean[1] + R[8] * mean[2] + cam.t[2];

  // recompute Sigma_c and J
  float Rq[9], M[9], cov[9], tmp[9], covc[9];
  quat_to_rotmat(quat, Rq);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[i * 3 + j] = Rq[i * 3 + j] * scale[j];
  mat3_mul_bt(M, M, cov);
  mat3_mul(R, cov, tmp);
  mat3_mul_bt(tmp, R, covc);
  float tanx = 0.5f * W / cam.fx, tany = 0.5f * H / cam.fy;
  float lim_xp = (W - cam.cx) / cam.fx + 0.3f * tanx, lim_xn = cam.cx / cam.fx + 0.3f * tanx;
  float lim_yp = (H - cam.cy) / cam.fy + 0.3f * tany, lim_yn = cam.cy / cam.fy + 0.3f * tany;
  float rz = 1.0f / z, rz2 = rz * rz, rz3 = rz2 * rz;
  float xr = x * rz, yr = y * rz;
  bool x_in = xr <= lim_xp && xr >= -lim_xn, y_in = yr <= lim_yp && yr >= -lim_yn;
  float tx = z * fminf(lim_xp, fmaxf(-lim_xn, xr));
  float ty = z * fminf(lim_yp, fmaxf(-lim_yn, yr));
  float j00 = cam.fx * rz, j02 = -cam.fx * tx * rz2, j11 = cam.fy * rz, j12 = -cam.fy * ty * rz2;

  // conic = inv(cov2d + eps I):  G2 = -conic * Vc * conic, Vc = [[va, vb/2],[vb/2, vc]]
  float ca = conic[0], cb = conic[1], cc = conic[2];
  float va = v_conic[0], vb = 0.5f * v_conic[1], vc = v_conic[2];
  // P = conic * Vc
  float p00 = ca * va + cb * vb, p01 = ca * vb + cb * vc;
  float p10 = cb * va + cc * vb, p11 = cb * vb + cc * vc;
  float g00 = -(p00 * ca + p01 * cb), g01 = -(p00 * cb + p01 * cc);
  float g10 = -(p10 * ca + p11 * cb), g11 = -(p10 * cb + p11 * cc);
  if (v_comp != 0.f) {   // compensation = sqrt(max(0, det(C)/det(C+eps I)))
    float det_conic = ca * cc - cb * cb;
    float v_sqr = v_comp * 0.5f / (compensation + 1e-6f);
    float om = 1.f - compensation * compensation;
    g00 += v_sqr * (om * ca - eps2d * det_conic);
    g01 += v_sqr * (om * cb);
    g10 += v_sqr * (om * cb);
    g11 += v_sqr * (om * cc - eps2d * det_conic);
  }
  // cov2d = J covc J^T ;  v_covc = J^T G2 J ;  v_J = (G2 + G2^T) J covc
  float J[6] = {j00, 0.f, j02, 0.f, j11, j12};
  float G2[4] = {g00, g01, g10, g11};
  float GJ[6];    // G2 * J (2x3)
  for (int c = 0; c < 3; ++c) {
    GJ[c] = G2[0] * J[c] + G2[1] * J[3 + c];
    GJ[3 + c] = G2[2] * J[c] + G2[3] * J[3 + c];
  }
  float v_covc[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) v_covc[r * 3 + c] = J[r] * GJ[c] + J[3 + r] * GJ[3 + c];
  float Gs[4] = {2.f * g00, g01 + g10, g01 + g10, 2.f * g11};
  float GsJ[6];
  for (int c = 0; c < 3; ++c) {
    GsJ[c] = Gs[0] * J[c] + Gs[1] * J[3 + c];
    GsJ[3 + c] = Gs[2] * J[c] + Gs[3] * J[3 + c];
  }
  float vJ00 = 0.f, vJ02 = 0.f, vJ11 = 0.f, vJ12 = 0.f;   // only the non-constant entries of J
  for (int k = 0; k < 3; ++k) {
    vJ00 += GsJ[k] * covc[k * 3 + 0];
    vJ02 += GsJ[k] * covc[k * 3 + 2];
    vJ11 += GsJ[3 + k] * covc[k * 3 + 1];
    vJ12 += GsJ[3 + k] * covc[k * 3 + 2];
  }
  // camera-space mean
  float vx = cam.fx * rz * v_mean2d[0];
  float vy = cam.fy * rz * v_mean2d[1];
  float vz = -(cam.fx * x * v_mean2d[0] + cam.fy * y * v_mean2d[1]) * rz2 + v_depth;
  if (x_in) vx += -cam.fx * rz2 * vJ02; else vz += -cam.fx * rz3 * vJ02 * tx;
  if (y_in) vy += -cam.fy * rz2 * vJ12; else vz += -cam.fy * rz3 * vJ12 * ty;
  vz += -cam.fx * rz2 * vJ00 - cam.fy * rz2 * vJ11 + 2.f * cam.fx * tx * rz3 * vJ02 +
        2.f * cam.fy * ty * rz3 * vJ12;
  // world mean, view matrix
  g.v_mean[0] = R[0] * vx + R[3] * vy + R[6] * vz;
  g.v_mean[1] = R[1] * vx + R[4] * vy + R[7] * vz;
  g.v_mean[2] = R[2] * vx + R[5] * vy + R[8] * vz;
  g.v_t[0] = vx; g.v_t[1] = vy; g.v_t[2] = vz;
  float vcs[9];   // symmetrised v_covc
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) vcs[r * 3 + c] = v_covc[r * 3 + c] + v_covc[c * 3 + r];
  // covc = R cov R^T : v_R = (v_covc + v_covc^T) R cov ; v_cov = R^T v_covc R
  float Rcov[9];
  mat3_mul(R, cov, Rcov);
  mat3_mul(vcs, Rcov, g.v_R);
  g.v_R[0] += vx * mean[0]; g.v_R[1] += vx * mean[1]; g.v_R[2] += vx * mean[2];
  g.v_R[3] += vy * mean[0]; g.v_R[4] += vy * mean[1]; g.v_R[5] += vy * mean[2];
  g.v_R[6] += vz * mean[0]; g.v_R[7] += vz * mean[1]; g.v_R[8] += vz * mean[2];
  float t2[9], v_cov[9];
  mat3_mul_at(R, v_covc, t2);
  mat3_mul(t2, R, v_cov);
  // cov = M M^T : v_M = (v_cov + v_cov^T) M ; M = Rq diag(s)
  float vs[9], v_M[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) vs[r * 3 + c] = v_cov[r * 3 + c] + v_cov[c * 3 + r];
  mat3_mul(vs, M, v_M);
  float v_Rq[9];
  for (int j = 0; j < 3; ++j) {
    g.v_scale[j] = Rq[0 * 3 + j] * v_M[0 * 3 + j] + Rq[1 * 3 + j] * v_M[1 * 3 + j] +
                   Rq[2 * 3 + j] * v_M[2 * 3 + j];
    for (int i = 0; i < 3; ++i) v_Rq[i * 3 + j] = v_M[i * 3 + j] * scale[j];
  }
  quat_to_rotmat_vjp(quat, v_Rq, g.v_quat);
  return g;
}

// VJP of colour = sum_k Y_k(normalize(dir)) coeff_k.  Writes v_coeff[KC*3] and v_dir[3].
template <int DEG>
MGS_HD void sh_vjp(const float dir[3], const float* coeff, const float v_rgb[3], float* v_coeff,
                   float v_dir[3]) {
  constexpr int KC = (DEG + 1) * (DEG + 1);
  float n2 = dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2];
  float inv = n2 > 0.f ? 1.0f / sqrtf(n2) : 0.f;
  float x = dir[0] * inv, y = dir[1] * inv, z = dir[2] * inv;
  float Y[KC], Yx[KC], Yy[KC], Yz[KC];
  sh_basis(DEG, x, y, z, Y);
  sh_basis_grad(DEG, x, y, z, Yx, Yy, Yz);
  float vx = 0.f, vy = 0.f, vz = 0.f;
  for (int k = 0; k < KC; ++k) {
    v_coeff[3 * k + 0] = Y[k] * v_rgb[0];
    v_coeff[3 * k + 1] = Y[k] * v_rgb[1];
    v_coeff[3 * k + 2] = Y[k] * v_rgb[2];
    float s = coeff[3 * k + 0] * v_rgb[0] + coeff[3 * k + 1] * v_rgb[1] + coeff[3 * k + 2] * v_rgb[2];
    vx += Yx[k] * s; vy += Yy[k] * s; vz += Yz[k] * s;
  }
  float d = vx * x + vy * y + vz * z;
  v_dir[0] = (vx - d * x) * inv;
  v_dir[1] = (vy - d * y) * inv;
  v_dir[2] = (vz - d * z) * inv;
}

}  // namespace mgs
#endif  // MGS_MATH_H_
