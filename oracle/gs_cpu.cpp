// gs_cpu.cpp -- fp32 C++/OpenMP restatement of the forward render path (TEST INFRASTRUCTURE
// ONLY: the checker for large scenes and the timed `cpu_baseline` of bench.py).
//
// PARITY UNPINNED: the reference (Maxwell-Zhao/RoboSimGS) contains no renderer and no CPU
// fallback (README.md:75 delegates 3DGS to Nerfstudio; README.md:29 lists the render stage as
// unreleased), so this is a "port" of the published gsplat 1.x algorithm as written down in
// SURVEY.md Appendix A.2 steps 1-9, in the textbook formulation: one 64-bit key
// (tile << 32 | depth bits) per (Gaussian, tile) pair, one global stable sort, one sequential
// blend loop per pixel.  It shares no code with robosimgs_amd/csrc.  Validated against the
// fp64 NumPy oracle in tests/test_oracle_cpu.py.
//
// Build: g++ -O3 -march=native -fopenmp -shared -fPIC gs_cpu.cpp -o _build/libgs_cpu.so
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct Splat {
  float mx, my, depth, ca, cb, cc, opac;
  int radius;
  int x0, y0, x1, y1;
};

inline void mat3mul(const float* A, const float* B, float* C, bool bt) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = 0.f;
      for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * (bt ? B[j * 3 + k] : B[k * 3 + j]);
      C[i * 3 + j] = s;
    }
}

// A.2 steps 1-5
bool project_one(const float* mean, const float* quat, const float* scale, const float* V,
                 const float* K, float W, float H, float eps2d, float near_p, float far_p,
                 float radius_clip, Splat& s, float& comp) {
  const float R[9] = {V[0], V[1], V[2], V[4], V[5], V[6], V[8], V[9], V[10]};
  const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  float x = R[0] * mean[0] + R[1] * mean[1] + R[2] * mean[2] + V[3];
  float y = R[3] * mean[0] + R[4] * mean[1] + R[5] * mean[2] + V[7];
  float z = R[6] * mean[0] + R[7] * mean[1] + R[8] * mean[2] + V[11];
  if (z < near_p || z > far_p) return false;
  float qn = std::sqrt(quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2] + quat[3] * quat[3]);
  float w = quat[0] / qn, qx = quat[1] / qn, qy = quat[2] / qn, qz = quat[3] / qn;
  float Rq[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - w * qz), 2 * (qx * qz + w * qy),
                 2 * (qx * qy + w * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - w * qx),
                 2 * (qx * qz - w * qy), 2 * (qy * qz + w * qx), 1 - 2 * (qx * qx + qy * qy)};
  float M[9], cov[9], t[9], cc3[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[i * 3 + j] = Rq[i * 3 + j] * scale[j];
  mat3mul(M, M, cov, true);
  mat3mul(R, cov, t, false);
  mat3mul(t, R, cc3, true);
  float tanx = 0.5f * W / fx, tany = 0.5f * H / fy;
  float lxp = (W - cx) / fx + 0.3f * tanx, lxn = cx / fx + 0.3f * tanx;
  float lyp = (H - cy) / fy + 0.3f * tany, lyn = cy / fy + 0.3f * tany;
  float rz = 1.f / z;
  float tx = z * std::min(lxp, std::max(-lxn, x * rz)), ty = z * std::min(lyp, std::max(-lyn, y * rz));
  float J[6] = {fx * rz, 0.f, -fx * tx * rz * rz, 0.f, fy * rz, -fy * ty * rz * rz};
  float JC[6];
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 3; ++c)
      JC[r * 3 + c] = J[r * 3] * cc3[c] + J[r * 3 + 1] * cc3[3 + c] + J[r * 3 + 2] * cc3[6 + c];
  float a = JC[0] * J[0] + JC[1] * J[1] + JC[2] * J[2];
  float b = JC[0] * J[3] + JC[1] * J[4] + JC[2] * J[5];
  float c = JC[3] * J[3] + JC[4] * J[4] + JC[5] * J[5];
  float det0 = a * c - b * b;
  a += eps2d; c += eps2d;
  float det = a * c - b * b;
  if (det <= 0.f) return false;
  float mid = 0.5f * (a + c);
  float lam = mid + std::sqrt(std::max(0.01f, mid * mid - det));
  float radius = std::ceil(3.f * std::sqrt(lam));
  if (radius <= radius_clip) return false;
  float mx = fx * x * rz + cx, my = fy * y * rz + cy;
  if (mx + radius <= 0 || mx - radius >= W || my + radius <= 0 || my - radius >= H) return false;
  s.mx = mx; s.my = my; s.depth = z;
  s.ca = c / det; s.cb = -b / det; s.cc = a / det;
  s.radius = (int)radius;
  comp = std::sqrt(std::max(0.f, det0 / det));
  return true;
}

// A.2 step 6
void sh_color(int deg, const float* mean, const float* campos, const float* coef, float* rgb) {
  float dx = mean[0] - campos[0], dy = mean[1] - campos[1], dz = mean[2] - campos[2];
  float inv = 1.f / std::sqrt(dx * dx + dy * dy + dz * dz);
  float x = dx * inv, y = dy * inv, z = dz * inv;
  float Y[16];
  Y[0] = 0.2820947917738781f;
  if (deg >= 1) { Y[1] = -0.48860251190292f * y; Y[2] = 0.48860251190292f * z; Y[3] = -0.48860251190292f * x; }
  float z2 = z * z, fC1 = x * x - y * y, fS1 = 2.f * x * y;
  if (deg >= 2) {
    float t = -1.092548430592079f * z;
    Y[4] = 0.5462742152960395f * fS1; Y[5] = t * y; Y[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
    Y[7] = t * x; Y[8] = 0.5462742152960395f * fC1;
  }
  if (deg >= 3) {
    float u = -2.285228997322329f * z2 + 0.4570457994644658f, w = 1.445305721320277f * z;
    float fC2 = x * fC1 - y * fS1, fS2 = x * fS1 + y * fC1;
    Y[9] = -0.5900435899266435f * fS2; Y[10] = w * fS1; Y[11] = u * y;
    Y[12] = z * (1.865881662950577f * z2 - 1.119528997770346f); Y[13] = u * x; Y[14] = w * fC1;
    Y[15] = -0.5900435899266435f * fC2;
  }
  int KC = (deg + 1) * (deg + 1);
  for (int c = 0; c < 3; ++c) {
    float s = 0.f;
    for (int k = 0; k < KC; ++k) s += Y[k] * coef[3 * k + c];
    rgb[c] = std::max(0.f, s + 0.5f);
  }
}

}  // namespace

// Whole forward frame for one camera.  feat layout: channels = 3 (rgb) or 4 (rgb + depth).
// Returns the number of tile intersections; counters[0] = visible Gaussians,
// counters[1] = pixel-Gaussian pair evaluations (for throughput reporting).
extern "C" long long gs_cpu_render(int n, const float* means, const float* quats,
                                   const float* scales, const float* opacities, int sh_degree,
                                   int coeff_stride, const float* sh, const float* viewmat,
                                   const float* K, int width, int height, float eps2d,
                                   float near_p, float far_p, float radius_clip, int channels,
                                   const float* background, int n_threads, float* render,
                                   float* alphas, long long* counters) {
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
  const int T = 16, tw = (width + T - 1) / T, th = (height + T - 1) / T;
  std::vector<Splat> sp(n);
  std::vector<float> feat((size_t)n * channels);
  std::vector<long long> cnt(n, 0);
  const float* V = viewmat;
  float campos[3];
  for (int i = 0; i < 3; ++i) campos[i] = -(V[0 + i] * V[3] + V[4 + i] * V[7] + V[8 + i] * V[11]);
  long long n_vis = 0;
#pragma omp parallel for schedule(static) reduction(+ : n_vis)
  for (int g = 0; g < n; ++g) {
    float comp;
    Splat s{};
    if (!project_one(means + 3 * g, quats + 4 * g, scales + 3 * g, V, K, (float)width,
                     (float)height, eps2d, near_p, far_p, radius_clip, s, comp)) {
      s.radius = 0;
      sp[g] = s;
      continue;
    }
    s.opac = opacities[g];
    float tr = (float)s.radius / T, tx = s.mx / T, ty = s.my / T;
    s.x0 = std::min(std::max(0, (int)std::floor(tx - tr)), tw);
    s.x1 = std::min(std::max(0, (int)std::ceil(tx + tr)), tw);
    s.y0 = std::min(std::max(0, (int)std::floor(ty - tr)), th);
    s.y1 = std::min(std::max(0, (int)std::ceil(ty + tr)), th);
    cnt[g] = (long long)(s.x1 - s.x0) * (s.y1 - s.y0);
    sh_color(sh_degree, means + 3 * g, campos, sh + (size_t)g * coeff_stride * 3, &feat[(size_t)g * channels]);
    if (channels == 4) feat[(size_t)g * 4 + 3] = s.depth;
    sp[g] = s;
    ++n_vis;
  }
  // A.2 step 7: keys in Gaussian-index order
  std::vector<long long> off(n + 1, 0);
  for (int g = 0; g < n; ++g) off[g + 1] = off[g] + cnt[g];
  const long long n_isect = off[n];
  std::vector<uint64_t> keys(n_isect);
  std::vector<int> ids(n_isect);
#pragma omp parallel for schedule(dynamic, 4096)
  for (int g = 0; g < n; ++g) {
    if (!cnt[g]) continue;
    const Splat& s = sp[g];
    uint32_t db;
    std::memcpy(&db, &s.depth, 4);
    long long o = off[g];
    for (int y = s.y0; y < s.y1; ++y)
      for (int x = s.x0; x < s.x1; ++x) {
        keys[o] = ((uint64_t)(y * tw + x) << 32) | db;
        ids[o++] = g;
      }
  }
  // A.2 step 8: stable sort by key (permutation sort), tile ranges
  std::vector<long long> perm(n_isect);
  std::iota(perm.begin(), perm.end(), 0LL);
  {
    // parallel: bucket by tile (counting sort, stable), then sort each tile's slice by depth
    std::vector<long long> tstart((size_t)tw * th + 1, 0);
    for (long long i = 0; i < n_isect; ++i) ++tstart[(keys[i] >> 32) + 1];
    for (size_t t = 0; t < (size_t)tw * th; ++t) tstart[t + 1] += tstart[t];
    std::vector<long long> cur(tstart.begin(), tstart.end() - 1);
    for (long long i = 0; i < n_isect; ++i) perm[cur[keys[i] >> 32]++] = i;
#pragma omp parallel for schedule(dynamic, 16)
    for (long long t = 0; t < (long long)tw * th; ++t)
      std::stable_sort(perm.begin() + tstart[t], perm.begin() + tstart[t + 1],
                       [&](long long a, long long b) { return keys[a] < keys[b]; });
    // A.2 step 9: blend
    long long evals = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : evals)
    for (long long t = 0; t < (long long)tw * th; ++t) {
      const int tx = (int)(t % tw), ty = (int)(t / tw);
      for (int py = ty * T; py < std::min((ty + 1) * T, height); ++py)
        for (int px = tx * T; px < std::min((tx + 1) * T, width); ++px) {
          float Tr = 1.f, C[4] = {0.f, 0.f, 0.f, 0.f};
          const float fx = px + 0.5f, fy = py + 0.5f;
          for (long long i = tstart[t]; i < tstart[t + 1]; ++i) {
            const int g = ids[perm[i]];
            const Splat& s = sp[g];
            ++evals;
            float dx = s.mx - fx, dy = s.my - fy;
            float sigma = 0.5f * (s.ca * dx * dx + s.cc * dy * dy) + s.cb * dx * dy;
            if (sigma < 0.f) continue;
            float alpha = std::min(0.999f, s.opac * std::exp(-sigma));
            if (alpha < 1.f / 255.f) continue;
            float nT = Tr * (1.f - alpha);
            if (nT <= 1e-4f) break;
            float wgt = alpha * Tr;
            for (int c = 0; c < channels; ++c) C[c] += wgt * feat[(size_t)g * channels + c];
            Tr = nT;
          }
          size_t p = (size_t)py * width + px;
          for (int c = 0; c < channels; ++c)
            render[p * channels + c] = C[c] + (background ? Tr * background[c] : 0.f);
          alphas[p] = 1.f - Tr;
        }
    }
    if (counters) { counters[0] = n_vis; counters[1] = evals; }
  }
  return n_isect;
}

extern "C" int gs_cpu_max_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
