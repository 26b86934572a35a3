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
  int radius;  // 0 = culled
};

// A.2 steps 1-5.  Returns radius == 0 for culled Gaussians (all other fields zeroed).
MGS_HD Projected project_gaussian(const float mean[3], const float quat[4],
                                  const float scale[3], const CameraParams& cam, float W,
                                  float H, float eps2d, float near_plane, float far_plane,
                                  float radius_clip) {
  Projected out;
  out.mean2d[0] = out.mean2d[1] = out.depth = 0.f;
  out.conic[0] = out.conic[1] = out.conic[2] = 0.f;
  out.compensation = 0.f;
  out.radius = 0;

  const float* R = cam.R;
  float x = R[0] * mean[0] + R[1] * mean[1] + R[2] * mean[2] + cam.t[0];
  float y = R[3] * mean[0] + R[4] * mean[1] + R[5] * mean[2] + cam.t[1];
  float z = R[6] * mean[0] + R[7] * mean[1] + R[8] * mean[2] + cam.t[2];
  if (!(z >= near_plane) || !(z <= far_plane)) return out;

  float Rq[9], M[9], cov[9], tmp[9], covc[9];
  quat_to_rotmat(quat, Rq);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[i * 3 + j] = Rq[i * 3 + j] * scale[j];
  mat3_mul_bt(M, M, cov);       // Sigma = M M^T
  mat3_mul(R, cov, tmp);        // R Sigma
  mat3_mul_bt(tmp, R, covc);    // R Sigma R^T

  float tanx = 0.5f * W / cam.fx, tany = 0.5f * H / cam.fy;
  float lim_xp = (W - cam.cx) / cam.fx + 0.3f * tanx, lim_xn = cam.cx / cam.fx + 0.3f * tanx;
  float lim_yp = (H - cam.cy) / cam.fy + 0.3f * tany, lim_yn = cam.cy / cam.fy + 0.3f * tany;
  float rz = 1.0f / z, rz2 = rz * rz;
  float tx = z * fminf(lim_xp, fmaxf(-lim_xn, x * rz));
  float ty = z * fminf(lim_yp, fmaxf(-lim_yn, y * rz));
  float j00 = cam.fx * rz, j02 = -cam.fx * tx * rz2;
  float j11 = cam.fy * rz, j12 = -cam.fy * ty * rz2;
  // cov2d = J covc J^T with J = [[j00,0,j02],[0,j11,j12]]
  float a = j00 * (j00 * covc[0] + j02 * covc[6]) + j02 * (j00 * covc[2] + j02 * covc[8]);
  float b = j00 * (j11 * covc[1] + j12 * covc[2]) + j02 * (j11 * covc[7] + j12 * covc[8]);
  float c = j11 * (j11 * covc[4] + j12 * covc[7]) + j12 * (j11 * covc[5] + j12 * covc[8]);
  float mx = cam.fx * x * rz + cam.cx, my = cam.fy * y * rz + cam.cy;

  float det0 = a * c - b * b;
  a += eps2d;
  c += eps2d;
  float det = a * c - b * b;
  if (!(det > 0.f)) return out;
  float inv_det = 1.0f / det;

  float m = 0.5f * (a + c);
  float lam = m + sqrtf(fmaxf(0.01f, m * m - det));
  float radius = ceilf(3.f * sqrtf(lam));
  if (!(radius > radius_clip)) return out;
  if (mx + radius <= 0.f || mx - radius >= W || my + radius <= 0.f || my - radius >= H)
    return out;

  out.mean2d[0] = mx;
  out.mean2d[1] = my;
  out.depth = z;
  out.conic[0] = c * inv_det;
  out.conic[1] = -b * inv_det;
  out.conic[2] = a * inv_det;
  out.compensation = sqrtf(fmaxf(0.f, det0 * inv_det));
  out.radius = (int)radius;
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

}  // namespace mgs
#endif  // MGS_MATH_H_
