// gs_cpu.cpp -- C++/OpenMP restatement of the render path (TEST INFRASTRUCTURE ONLY: the checker
// for large scenes and the timed `cpu_baseline` of bench.py).
//
// PARITY UNPINNED: the reference (Maxwell-Zhao/RoboSimGS) contains no renderer and no CPU
// fallback (README.md:75 delegates 3DGS to Nerfstudio; README.md:29 lists the render stage as
// unreleased), so this is a "port" of the published gsplat 1.x algorithm as written down in
// SURVEY.md Appendix A.2 steps 1-10, in the textbook formulation: one 64-bit key
// (tile << 32 | depth bits) per (Gaussian, tile) pair, one global stable sort, one sequential
// blend loop per pixel.  It shares no code with robosimgs_amd/csrc.  Validated against the
// fp64 NumPy oracle in tests/test_oracle_cpu.py.
//
// One template, two instantiations:
//   float   gs_cpu_render      the timed CPU baseline (device-like rounding)
//   double  gs_cpu_render_f64  the full-size reference answer.  It can also report, per pixel,
//           how close every branch of the blend came to flipping (the margins of
//           oracle/gs_oracle_np.py:rasterize) and which pixels a per-Gaussian knife edge of the
//           projection reaches (gaussian_edge_mask), so that tests demand ZERO unexplained pixels
//           over the 1e-4 tolerance at 1080p and 4K too; and it can run A.2 step 10 (the blend's
//           backward) with fp64 accumulation: gradients w.r.t. the projected quantities.
//
// Build: g++ -O3 -std=c++17 -fopenmp -shared -fPIC gs_cpu.cpp -o _build/libgs_cpu.so
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

template <typename F>
struct Splat {
  F mx, my, depth, ca, cb, cc, opac;
  float depth32;            // sort key: fp32 depth bits (A.2 step 7), also in the double build
  int radius;               // the x extent under the per-axis rule
  int radius_y;             // == radius under the classic rule
  int x0, y0, x1, y1;
  // knife-edge bookkeeping (margins only): outer / inner tile rectangles under perturbation
  int ox0, oy0, ox1, oy1, ix0, iy0, ix1, iy1;
  bool inner_on, unsure;
};

template <typename F>
inline void mat3mul(const F* A, const F* B, F* C, bool bt) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      F s = 0;
      for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * (bt ? B[j * 3 + k] : B[k * 3 + j]);
      C[i * 3 + j] = s;
    }
}

// A.2 steps 1-5.  Returns false when culled; `edge` (optional) receives the un-culled
// intermediates the knife-edge classification needs: {extent x, mx, my, z, det > 0, extent y, extent factor e}
// (extent = 3 sqrt(lambda) on both axes under the classic rule).
// rule 0: A.2 step 5 (gsplat 1.4).  rule 1: SURVEY.md A.4 (gsplat >= 1.5) -- per-axis extents e sqrt(Sigma_ii),
// e = min(3.33, sqrt(2 ln(255 opacity))), opacity < 1/255 culled, radius_clip culls only when both extents are under it.
template <typename F>
bool project_one(const float* mean, const float* quat, const float* scale, const float* Vf,
                 const float* Kf, F W, F H, F eps2d, F near_p, F far_p, F radius_clip,
                 Splat<F>& s, F& comp, F* edge, int rule = 0, F opacity = 1) {
  F V[12], K[9];
  for (int i = 0; i < 12; ++i) V[i] = Vf[i];
  for (int i = 0; i < 9; ++i) K[i] = Kf[i];
  const F R[9] = {V[0], V[1], V[2], V[4], V[5], V[6], V[8], V[9], V[10]};
  const F fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  const F m0 = mean[0], m1 = mean[1], m2 = mean[2];
  F x = R[0] * m0 + R[1] * m1 + R[2] * m2 + V[3];
  F y = R[3] * m0 + R[4] * m1 + R[5] * m2 + V[7];
  F z = R[6] * m0 + R[7] * m1 + R[8] * m2 + V[11];
  if (edge) { edge[0] = 0; edge[1] = 0; edge[2] = 0; edge[3] = z; edge[4] = 0; edge[5] = 0; edge[6] = 0; }
  const bool z_ok = !(z < near_p || z > far_p);
  if (!z_ok && !edge) return false;
  const F zs = z_ok ? z : (z > 0 ? z : F(1));
  const F q0 = quat[0], q1 = quat[1], q2 = quat[2], q3 = quat[3];
  F qn = std::sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  F w = q0 / qn, qx = q1 / qn, qy = q2 / qn, qz = q3 / qn;
  F Rq[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - w * qz), 2 * (qx * qz + w * qy),
             2 * (qx * qy + w * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - w * qx),
             2 * (qx * qz - w * qy), 2 * (qy * qz + w * qx), 1 - 2 * (qx * qx + qy * qy)};
  F M[9], cov[9], t[9], cc3[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[i * 3 + j] = Rq[i * 3 + j] * (F)scale[j];
  mat3mul(M, M, cov, true);
  mat3mul(R, cov, t, false);
  mat3mul(t, R, cc3, true);
  F tanx = F(0.5) * W / fx, tany = F(0.5) * H / fy;
  F lxp = (W - cx) / fx + F(0.3) * tanx, lxn = cx / fx + F(0.3) * tanx;
  F lyp = (H - cy) / fy + F(0.3) * tany, lyn = cy / fy + F(0.3) * tany;
  F rz = F(1) / zs;
  F tx = zs * std::min(lxp, std::max(-lxn, x * rz)), ty = zs * std::min(lyp, std::max(-lyn, y * rz));
  F J[6] = {fx * rz, 0, -fx * tx * rz * rz, 0, fy * rz, -fy * ty * rz * rz};
  F JC[6];
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 3; ++c)
      JC[r * 3 + c] = J[r * 3] * cc3[c] + J[r * 3 + 1] * cc3[3 + c] + J[r * 3 + 2] * cc3[6 + c];
  F a = JC[0] * J[0] + JC[1] * J[1] + JC[2] * J[2];
  F b = JC[0] * J[3] + JC[1] * J[4] + JC[2] * J[5];
  F c = JC[3] * J[3] + JC[4] * J[4] + JC[5] * J[5];
  F det0 = a * c - b * b;
  a += eps2d; c += eps2d;
  F det = a * c - b * b;
  const bool det_ok = det > 0;
  if (!det_ok) return false;
  F mid = F(0.5) * (a + c);
  F lam = mid + std::sqrt(std::max(F(0.01), mid * mid - det));
  F v3 = F(3) * std::sqrt(lam), v3y = v3, efac = 3;
  bool op_ok = true;
  if (rule == 1) {
    op_ok = opacity >= F(1) / F(255);
    efac = std::min(F(3.33), std::sqrt(F(2) * std::log((op_ok ? opacity : F(1) / F(255)) * F(255))));
    v3 = efac * std::sqrt(a);
    v3y = efac * std::sqrt(c);
  }
  F radius = std::ceil(v3), radius_y = std::ceil(v3y);
  F mx = fx * x * rz + cx, my = fy * y * rz + cy;
  s.mx = mx; s.my = my; s.depth = z; s.depth32 = (float)z;
  s.ca = c / det; s.cb = -b / det; s.cc = a / det;
  s.radius = (int)radius;
  s.radius_y = (int)radius_y;
  comp = std::sqrt(std::max(F(0), det0 / det));
  if (edge) { edge[0] = v3; edge[1] = mx; edge[2] = my; edge[3] = z; edge[4] = 1; edge[5] = v3y; edge[6] = efac; }
  if (!z_ok || !op_ok) return false;
  if (radius <= radius_clip && radius_y <= radius_clip) return false;
  if (!(radius > 0) || !(radius_y > 0)) return false;
  if (mx + radius <= 0 || mx - radius >= W || my + radius_y <= 0 || my - radius_y >= H) return false;
  return true;
}

// A.2 step 6
template <typename F>
void sh_color(int deg, const float* mean, const F* campos, const float* coef, F* rgb) {
  F dx = mean[0] - campos[0], dy = mean[1] - campos[1], dz = mean[2] - campos[2];
  F inv = F(1) / std::sqrt(dx * dx + dy * dy + dz * dz);
  F x = dx * inv, y = dy * inv, z = dz * inv;
  F Y[16];
  Y[0] = F(0.2820947917738781);
  if (deg >= 1) { Y[1] = F(-0.48860251190292) * y; Y[2] = F(0.48860251190292) * z; Y[3] = F(-0.48860251190292) * x; }
  F z2 = z * z, fC1 = x * x - y * y, fS1 = 2 * x * y;
  if (deg >= 2) {
    F t = F(-1.092548430592079) * z;
    Y[4] = F(0.5462742152960395) * fS1; Y[5] = t * y; Y[6] = F(0.9461746957575601) * z2 - F(0.3153915652525201);
    Y[7] = t * x; Y[8] = F(0.5462742152960395) * fC1;
  }
  if (deg >= 3) {
    F u = F(-2.285228997322329) * z2 + F(0.4570457994644658), w = F(1.445305721320277) * z;
    F fC2 = x * fC1 - y * fS1, fS2 = x * fS1 + y * fC1;
    Y[9] = F(-0.5900435899266435) * fS2; Y[10] = w * fS1; Y[11] = u * y;
    Y[12] = z * (F(1.865881662950577) * z2 - F(1.119528997770346)); Y[13] = u * x; Y[14] = w * fC1;
    Y[15] = F(-0.5900435899266435) * fC2;
  }
  int KC = (deg + 1) * (deg + 1);
  for (int c = 0; c < 3; ++c) {
    F s = 0;
    for (int k = 0; k < KC; ++k) s += Y[k] * (F)coef[3 * k + c];
    rgb[c] = std::max(F(0), s + F(0.5));
  }
}

struct Extras {            // optional outputs of the double build (all may be null)
  float* margins = nullptr;        // [4,H,W]  alpha / T / sigma / depth-order margins (inf where nothing was decided)
  uint8_t* edge_mask = nullptr;    // [H,W]
  long long* n_edge = nullptr;     // Gaussians with an uncertain tile rectangle
  // flip weight [H,W] (needs margins): sum of what every decision within flip_eps = {alpha, T, sigma, depth}
  // of flipping is worth, as a blend weight (multiply by the feature range for a colour bound):
  //   alpha / sigma toggle of a Gaussian   alpha T        T threshold (stop here or go on)   T
  //   depth-order swap of two neighbours   alpha_i alpha_j T      uncertain tile membership   alpha
  // A toggle also moves every later T by the factor (1 - alpha): later T thresholds are tested against
  // eps_T + (sum of the toggled alphas so far).
  float* flip_weight = nullptr;
  const float* flip_eps = nullptr;
  // touched [N] (needs margins, edge_mask and flip_eps): 1 for every Gaussian that reaches alpha >= 0.5/255 at
  // some could-flip pixel (a margin below flip_eps, or an edge pixel).  A flip at a pixel changes T and the
  // colour behind for every Gaussian blended there: only these rows of the gradient may differ from the
  // oracle's by more than rounding.
  uint8_t* touched = nullptr;
  // budget [N,4] (needs touched's inputs and v_render / v_alpha): per Gaussian, how far its rows of the blend's
  // gradient {means2d, conics, feats, opacity} can move when the near-flip decisions at the could-flip pixels it
  // reaches go the other way.  At such a pixel p the decisions within flip_eps are worth flip_weight[p] as a blend
  // weight; toggling one of them changes, for every Gaussian g blended at p (alpha_g >= 0.5/255),
  //   T_g or the colour behind g, hence  d loss / d alpha_g  by at most  A = flip_weight[p] (sum_c 2 max|f_c| |v_c| + |v_alpha'|) / (1 - alpha_g)
  //   (v_alpha' carries the background term), and g's own weight alpha_g T_g by at most flip_weight[p];
  // g's OWN toggle (its alpha or sigma test within eps) removes its whole term: |d loss / d alpha_g| <= T_g (...) with
  // T_g <= flip_weight[p] / alpha_g.  Summed over the pixels:
  //   opacity  vis A  (own: A / opacity)      means2d  ov A max|conic d|  (own: A max|conic d|)
  //   conics   ov A max(dx^2 / 2, |dx dy|, dy^2 / 2)  (own: without ov)      feats  flip_weight[p] max_c |v_c| / (1 - alpha_g)
  // A test then demands |got - ref| <= rounding tolerance + 1.5 budget on EVERY row: a gradient that is wrong on the
  // Gaussians near a threshold by more than the threshold can explain no longer passes.
  double* budget = nullptr;
  // thresholds {alpha >= thr[0] / 255, stop at T' <= thr[1] * 1e-4}; null = {1, 1}.  Moving a threshold by less than the
  // gate's eps flips exactly the decisions the margins call "could flip" and changes nothing else: how a test produces
  // REAL flips to hold the flip weight and the gradient budget against (tests/test_oracle_cpu.py).
  const float* thresholds = nullptr;
  // backward of the blend (A.2 step 10), fp64 accumulation
  const float* v_render = nullptr; // [H,W,ch]
  const float* v_alpha = nullptr;  // [H,W]
  double* g_means2d = nullptr;     // [N,2]
  double* g_conics = nullptr;      // [N,3]
  double* g_feats = nullptr;       // [N,ch]
  double* g_opac = nullptr;        // [N]
  // projected quantities as the oracle computed them (for the projection's autograd oracle)
  double* o_means2d = nullptr;     // [N,2]
  double* o_conics = nullptr;      // [N,3]
  double* o_feats = nullptr;       // [N,ch]
  int32_t* o_radii = nullptr;      // [N]  (the x extents under the per-axis rule)
  int radius_rule = 0;             // 0: A.2 step 5; 1: SURVEY.md A.4 (per axis, opacity-aware; classic rasterize mode)
  // STAGE-ISOLATED BLEND (gs_cpu_blend_f64): the projected quantities and the depth-ordered tile lists are GIVEN (e.g. the
  // GPU's own fp32 means2d / conics / opacities / feats and its lists, read back): A.2 steps 1-8 are skipped and steps
  // 9-10 run in F on exactly those inputs -- the answer a blend kernel must reproduce whatever the projection's rounding
  // did upstream (on an ill-conditioned scene the whole path is only gated relative to an fp32 restatement; this is not).
  const float* in_means2d = nullptr;       // [N,2]
  const float* in_conics = nullptr;        // [N,3]
  const float* in_opac = nullptr;          // [N]
  const float* in_feats = nullptr;         // [N,ch]
  const float* in_depths = nullptr;        // [N] (depth-tie margins only; null: none)
  const int32_t* in_flatten_ids = nullptr; // [n_isect]
  const int32_t* in_tile_offsets = nullptr;// [tw*th + 1]
  // noise_weight [H,W] (with margins): sum over the pixel's contributors of SIGMA_ABS S_k alpha_k T_k / (1 - alpha_k) -- what
  // the rounding of sigma alone (d alpha / alpha = SIGMA_ABS S, see the conditioned margins) can move the pixel by, as a
  // blend weight (x 2 max|feature|): the SMOOTH part of an fp32 blend's error, next to the flips.  ~1e-6 on a
  // well-conditioned scene; up to 1e-3 where needle-like Gaussians are seen hundreds of pixels from their means.
  float* noise_weight = nullptr;
};

template <typename F>
long long render_impl(int n, const float* means, const float* quats, const float* scales,
                      const float* opacities, int sh_degree, int coeff_stride, const float* sh,
                      const float* viewmat, const float* K, int width, int height, float eps2d,
                      float near_p, float far_p, float radius_clip, int channels,
                      const float* background, int n_threads, float* render, float* alphas,
                      long long* counters, const Extras& ex) {
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
  const int T = 16, tw = (width + T - 1) / T, th = (height + T - 1) / T;
  const bool want_edges = ex.edge_mask != nullptr;
  std::vector<Splat<F>> sp(n);
  std::vector<F> feat((size_t)n * channels);
  std::vector<long long> cnt(n, 0);
  const float* V = viewmat;
  F campos[3] = {0, 0, 0};
  for (int i = 0; i < 3 && V; ++i)             // (no camera in the stage-isolated blend)
    campos[i] = -((F)V[0 + i] * (F)V[3] + (F)V[4 + i] * (F)V[7] + (F)V[8 + i] * (F)V[11]);
  long long n_vis = 0, n_edge = 0;
  const bool given = ex.in_flatten_ids != nullptr;
  if (given) {
#pragma omp parallel for schedule(static)
    for (int g = 0; g < n; ++g) {
      Splat<F> s{};
      s.mx = ex.in_means2d[2 * g]; s.my = ex.in_means2d[2 * g + 1];
      s.ca = ex.in_conics[3 * g]; s.cb = ex.in_conics[3 * g + 1]; s.cc = ex.in_conics[3 * g + 2];
      s.opac = ex.in_opac[g];
      s.depth = ex.in_depths ? (F)ex.in_depths[g] : F(1);
      s.depth32 = ex.in_depths ? ex.in_depths[g] : 1.f;
      s.radius = s.radius_y = 1;                    // (listed or not is the lists' business)
      for (int c = 0; c < channels; ++c) feat[(size_t)g * channels + c] = ex.in_feats[(size_t)g * channels + c];
      sp[g] = s;
    }
  }
#pragma omp parallel for schedule(static) reduction(+ : n_vis, n_edge)
  for (int g = 0; g < (given ? 0 : n); ++g) {
    F comp;
    F edge[7];
    Splat<F> s{};
    const bool vis = project_one<F>(means + 3 * g, quats + 4 * g, scales + 3 * g, V, K, (F)width,
                                    (F)height, (F)eps2d, (F)near_p, (F)far_p, (F)radius_clip, s,
                                    comp, want_edges ? edge : nullptr, ex.radius_rule, (F)opacities[g]);
    s.opac = opacities[g];
    if (want_edges && edge[4] != 0) {
      // gaussian_edge_mask of oracle/gs_oracle_np.py: rectangles under the perturbations an fp32
      // projection can apply (radius +-1 when 3 sqrt(lambda) is within 3e-5 relative of an integer,
      // mean +-1e-3 px, depth 1e-5 relative around the near / far plane)
      const F mx = edge[1], my = edge[2], z = edge[3];
      F r_lo[2], r_hi[2];
      for (int ax = 0; ax < 2; ++ax) {
        const F v3 = edge[ax ? 5 : 0];
        const F r = std::ceil(v3), fr = v3 - std::floor(v3);
        F tol = F(3e-5) * std::max(v3, F(1)) + F(1e-6);
        if (ex.radius_rule == 1)       // d(extent factor) = d(ln) / e with d(ln) ~ 2e-7 in fp32 (gs_oracle_np.gaussian_edge_mask)
          tol += v3 * (F(2e-7) / std::max(edge[6] * edge[6], F(1e-12)));
        r_lo[ax] = (fr > 0 && fr < tol) ? r - 1 : r;
        r_hi[ax] = (1 - fr < tol || fr == 0) ? r + 1 : r;
      }
      bool zin = z >= near_p * (1 + 1e-5) && z <= far_p * (1 - 1e-5);
      bool zout = z >= near_p * (1 - 1e-5) && z <= far_p * (1 + 1e-5);
      if (ex.radius_rule == 1) {       // culled below 1/255: present or not is uncertain within 1e-5 of it
        zin = zin && (F)opacities[g] >= F(1 + 1e-5) / F(255);
        zout = zout && (F)opacities[g] >= F(1 - 1e-5) / F(255);
      }
      auto rect = [&](const F* rr, F d, int& x0, int& y0, int& x1, int& y1) {
        x0 = std::min(std::max(0, (int)std::floor((mx - rr[0] - d) / T)), tw);
        x1 = std::min(std::max(0, (int)std::ceil((mx + rr[0] + d) / T)), tw);
        y0 = std::min(std::max(0, (int)std::floor((my - rr[1] - d) / T)), th);
        y1 = std::min(std::max(0, (int)std::ceil((my + rr[1] + d) / T)), th);
        return !(mx + rr[0] + d <= 0 || mx - rr[0] - d >= width || my + rr[1] + d <= 0 || my - rr[1] - d >= height) &&
               (rr[0] > radius_clip || rr[1] > radius_clip) && rr[0] > 0 && rr[1] > 0;
      };
      const F dmu = F(1e-3);
      bool ion = rect(r_lo, -dmu, s.ix0, s.iy0, s.ix1, s.iy1) && zin;
      bool oon = rect(r_hi, dmu, s.ox0, s.oy0, s.ox1, s.oy1) && zout;
      if (!ion) s.ix0 = s.ix1 = s.iy0 = s.iy1 = 0;
      s.inner_on = ion;
      s.unsure = oon && (!ion || s.ox0 != s.ix0 || s.ox1 != s.ix1 || s.oy0 != s.iy0 || s.oy1 != s.iy1);
      if (s.unsure) ++n_edge;
    }
    if (!vis) {
      s.radius = s.radius_y = 0;
      sp[g] = s;
      continue;
    }
    F tr = (F)s.radius / T, try_ = (F)s.radius_y / T, tx = s.mx / T, ty = s.my / T;
    s.x0 = std::min(std::max(0, (int)std::floor(tx - tr)), tw);
    s.x1 = std::min(std::max(0, (int)std::ceil(tx + tr)), tw);
    s.y0 = std::min(std::max(0, (int)std::floor(ty - try_)), th);
    s.y1 = std::min(std::max(0, (int)std::ceil(ty + try_)), th);
    cnt[g] = (long long)(s.x1 - s.x0) * (s.y1 - s.y0);
    sh_color<F>(sh_degree, means + 3 * g, campos, sh + (size_t)g * coeff_stride * 3, &feat[(size_t)g * channels]);
    if (channels == 4) feat[(size_t)g * 4 + 3] = s.depth;
    sp[g] = s;
    ++n_vis;
  }
  if (ex.o_radii)
    for (int g = 0; g < n; ++g) {
      ex.o_radii[g] = sp[g].radius;
      const bool v = sp[g].radius > 0;
      if (ex.o_means2d) { ex.o_means2d[2 * g] = v ? sp[g].mx : 0; ex.o_means2d[2 * g + 1] = v ? sp[g].my : 0; }
      if (ex.o_conics) { ex.o_conics[3 * g] = v ? sp[g].ca : 0; ex.o_conics[3 * g + 1] = v ? sp[g].cb : 0; ex.o_conics[3 * g + 2] = v ? sp[g].cc : 0; }
      if (ex.o_feats) for (int c = 0; c < channels; ++c) ex.o_feats[(size_t)g * channels + c] = v ? feat[(size_t)g * channels + c] : 0;
    }
  // A.2 steps 7-8 (or the given lists): ids[perm[i]] is the Gaussian of list entry i, tstart the tiles' ranges
  const size_t n_tiles_sz = (size_t)tw * th;
  std::vector<long long> tstart(n_tiles_sz + 1, 0);
  long long n_isect = 0;
  std::unique_ptr<uint64_t[]> keys_buf;
  std::unique_ptr<int[]> ids_buf;
  std::unique_ptr<long long[]> perm_buf;
  if (given) {
    for (size_t t = 0; t <= n_tiles_sz; ++t) tstart[t] = ex.in_tile_offsets[t];
    n_isect = tstart[n_tiles_sz];
    ids_buf.reset(new int[(size_t)n_isect + 1]);
    perm_buf.reset(new long long[(size_t)n_isect + 1]);
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < n_isect; ++i) { ids_buf[i] = ex.in_flatten_ids[i]; perm_buf[i] = i; }
  } else {
  // A.2 step 7: keys in Gaussian-index order
  std::vector<long long> off(n + 1, 0);
  for (int g = 0; g < n; ++g) off[g + 1] = off[g] + cnt[g];
  n_isect = off[n];
  // (uninitialised: a std::vector would zero 100 MB on one thread before the parallel loops fill every entry)
  keys_buf.reset(new uint64_t[(size_t)n_isect + 1]);
  ids_buf.reset(new int[(size_t)n_isect + 1]);
  uint64_t* keys = keys_buf.get();
  int* ids = ids_buf.get();
#pragma omp parallel for schedule(dynamic, 4096)
  for (int g = 0; g < n; ++g) {
    if (!cnt[g]) continue;
    const Splat<F>& s = sp[g];
    uint32_t db;
    std::memcpy(&db, &s.depth32, 4);
    long long o = off[g];
    for (int y = s.y0; y < s.y1; ++y)
      for (int x = s.x0; x < s.x1; ++x) {
        keys[o] = ((uint64_t)(y * tw + x) << 32) | db;
        ids[o++] = g;
      }
  }
  // A.2 step 8: stable sort by key (permutation sort), tile ranges
  perm_buf.reset(new long long[(size_t)n_isect + 1]);
  long long* perm = perm_buf.get();
  // parallel: bucket by tile (counting sort, STABLE: every thread owns a contiguous run of the keys, counts it, takes its
  // place behind the lower threads' shares of each tile and drops its keys there in order), then sort each tile's slice by
  // depth.  (Serial, this pass and the histogram before it were a third of the frame on a 128-thread host.)
  {
    int n_thr = 1;
#ifdef _OPENMP
    n_thr = omp_get_max_threads();
#endif
    std::vector<long long> hist((size_t)n_thr * n_tiles_sz, 0);
    // The keys are cut into n_thr SLICES and the slices are dealt to whatever team the runtime really starts
    // (`omp for schedule(static, 1)` over the slices): with fewer threads than omp_get_max_threads() promised
    // (OMP_THREAD_LIMIT, OMP_DYNAMIC, a nested region) a thread takes several slices and none is left uncounted.
    long long scattered = 0;
#pragma omp parallel num_threads(n_thr) reduction(+ : scattered)
    {
#pragma omp for schedule(static, 1)
      for (int k = 0; k < n_thr; ++k) {
        const long long i0 = n_isect * k / n_thr, i1 = n_isect * (k + 1) / n_thr;
        long long* h = hist.data() + (size_t)k * n_tiles_sz;
        for (long long i = i0; i < i1; ++i) ++h[keys[i] >> 32];
      }                                                                   // (implicit barrier)
#pragma omp for schedule(static)
      for (long long t = 0; t < (long long)n_tiles_sz; ++t) {          // per tile: total, and every slice's share turned into its offset
        long long run = 0;
        for (int kk = 0; kk < n_thr; ++kk) {
          const long long c = hist[(size_t)kk * n_tiles_sz + t];
          hist[(size_t)kk * n_tiles_sz + t] = run;
          run += c;
        }
        tstart[t + 1] = run;
      }
#pragma omp single
      for (size_t t = 0; t < n_tiles_sz; ++t) tstart[t + 1] += tstart[t];
#pragma omp for schedule(static, 1)
      for (int k = 0; k < n_thr; ++k) {
        const long long i0 = n_isect * k / n_thr, i1 = n_isect * (k + 1) / n_thr;
        long long* h = hist.data() + (size_t)k * n_tiles_sz;
        for (long long i = i0; i < i1; ++i) {
          const size_t t = (size_t)(keys[i] >> 32);
          perm[tstart[t] + h[t]++] = i;
          ++scattered;
        }
      }
    }
    if (scattered != n_isect || tstart[n_tiles_sz] != n_isect) {        // every entry of `perm` (uninitialised storage) was written
      std::fprintf(stderr, "gs_cpu: counting sort placed %lld of %lld keys\n", scattered, n_isect);
      std::abort();
    }
  }
#pragma omp parallel for schedule(dynamic, 16)
  for (long long t = 0; t < (long long)tw * th; ++t)
    std::stable_sort(perm + tstart[t], perm + tstart[t + 1],
                     [&](long long a, long long b) { return keys[a] < keys[b]; });

  }
  const int* ids = ids_buf.get();
  const long long* perm = perm_buf.get();
  const bool want_margins = ex.margins != nullptr;
  const bool want_bwd = ex.v_render != nullptr;
  const double thr_alpha = (ex.thresholds ? (double)ex.thresholds[0] : 1.0) / 255.0, thr_T = (ex.thresholds ? (double)ex.thresholds[1] : 1.0) * 1e-4;
  const size_t n_px = (size_t)width * height;
  const float inf = std::numeric_limits<float>::infinity();
  if (want_margins) std::fill(ex.margins, ex.margins + 4 * n_px, inf);
  const bool want_fw = want_margins && ex.flip_weight != nullptr && ex.flip_eps != nullptr;
  std::vector<float> t_at_min;         // T at the pair that came closest to the T threshold (for the edge pixels)
  if (want_fw) { std::fill(ex.flip_weight, ex.flip_weight + n_px, 0.f); t_at_min.assign(n_px, 0.f); }
  const double fe_a = want_fw ? ex.flip_eps[0] : 0, fe_t = want_fw ? ex.flip_eps[1] : 0,
               fe_s = want_fw ? ex.flip_eps[2] : 0, fe_z = want_fw ? ex.flip_eps[3] : 0;
  std::vector<long long> last_idx;     // backward: list position of the last blended Gaussian, -1 if none
  std::vector<double> t_final;         // backward: final transmittance in full precision
  if (want_bwd) { last_idx.assign(n_px, -1); t_final.assign(n_px, 1.0); }
  // A.2 step 9: blend
  long long evals = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : evals)
  for (long long t = 0; t < (long long)tw * th; ++t) {
    const int tx = (int)(t % tw), ty = (int)(t / tw);
    for (int py = ty * T; py < std::min((ty + 1) * T, height); ++py)
      for (int px = tx * T; px < std::min((tx + 1) * T, width); ++px) {
        F Tr = 1, C[4] = {0, 0, 0, 0};
        const F fx = px + F(0.5), fy = py + F(0.5);
        F m_a = inf, m_t = inf, m_s = inf, m_z = inf, z_prev = -1;
        double fw = 0, loose = 0, wt_prev = 0, t_min = 0, noise = 0;
        F r_acc = 0, sr_acc = 0;      // over the contributors so far: sum of r^2, r = alpha / (1 - alpha), and of SIGMA_ABS S r (T's own relative error)
        long long last = -1;
        for (long long i = tstart[t]; i < tstart[t + 1]; ++i) {
          const int g = ids[perm[i]];
          const Splat<F>& s = sp[g];
          ++evals;
          F dx = s.mx - fx, dy = s.my - fy;
          F sigma = F(0.5) * (s.ca * dx * dx + s.cc * dy * dy) + s.cb * dx * dy;
          F alpha = std::min(F(0.999), s.opac * std::exp(-sigma));
          F nT = Tr * (1 - alpha);
          if (want_margins) {
            // conditioned margins (oracle/gs_oracle_np.py:rasterize): alpha's and T''s distance from their thresholds less
            // what any fp32 evaluation of the pair can be off by -- SIGMA_ABS S on sigma, times alpha / (1 - alpha) on T'
            const F S = F(0.5) * (std::abs(s.ca) * dx * dx + std::abs(s.cc) * dy * dy) + std::abs(s.cb * dx * dy);
            // (a clamped alpha is 0.999 in every implementation: no error of its own to amplify)
            const F cS = F(1e-6) * S, r_amp = (s.opac * std::exp(-sigma) >= F(0.999)) ? F(0) : alpha / std::max(1 - alpha, F(1e-3));
            const F ma = std::max(std::abs(alpha * 255 - 1) - cS, F(0));
            m_a = std::min(m_a, ma);
            bool toggle = want_fw && (double)ma < fe_a;
            if (alpha >= F(0.5 / 255.0)) {
              // (T = prod (1 - alpha_j) carries every earlier contributor's error too: sum_j r_j d alpha_j / alpha_j)
              // -- independent roundings: their eps shares add in quadrature, the conditioned shares (rare, large) plainly
              const F mt = std::max(std::abs(nT / F(1e-4) - 1) - (cS * r_amp + sr_acc), F(0)) / std::sqrt((1 + r_amp) * (1 + r_amp) + r_acc);
              if (S > 0) {
                const F ms = std::abs(sigma) / S;
                m_s = std::min(m_s, ms);
                if (want_fw && (double)ms < fe_s) toggle = true;
              }
              if (toggle) { fw += (double)(alpha * Tr); loose += (double)alpha; toggle = false; }
              if (mt < m_t) { m_t = mt; t_min = (double)Tr; }
              if (want_fw && (double)mt < fe_t + loose) fw += (double)Tr;
            }
            if (toggle) { fw += (double)(alpha * Tr); loose += (double)alpha; }
          }
          if (sigma < 0) continue;
          if (alpha < F(thr_alpha)) continue;
          if (want_margins) {      // consecutive contributors whose depths are within rounding of a tie may swap
            if (z_prev > 0) {
              const F mz = (s.depth - z_prev) / s.depth;
              m_z = std::min(m_z, mz);
              if (want_fw && (double)mz < fe_z) fw += wt_prev * (double)alpha;
            }
            z_prev = s.depth;
            wt_prev = (double)(alpha * Tr);
          }
          if (nT <= F(thr_T)) break;
          F wgt = alpha * Tr;
          if (want_margins) {
            const F Sk = F(0.5) * (std::abs(s.ca) * dx * dx + std::abs(s.cc) * dy * dy) + std::abs(s.cb * dx * dy);
            const F rk = (s.opac * std::exp(-sigma) >= F(0.999)) ? F(0) : alpha / std::max(1 - alpha, F(1e-3));
            r_acc += rk * rk; sr_acc += F(1e-6) * Sk * rk;
          }
          if (want_margins && ex.noise_weight) {
            const double S = 0.5 * (std::abs((double)s.ca) * dx * dx + std::abs((double)s.cc) * dy * dy) + std::abs((double)(s.cb * dx * dy));
            if ((double)(s.opac * std::exp(-sigma)) < 0.999) noise += 1e-6 * S * (double)wgt / std::max(1.0 - (double)alpha, 1e-3);
          }
          for (int c = 0; c < channels; ++c) C[c] += wgt * feat[(size_t)g * channels + c];
          Tr = nT;
          last = i;
        }
        size_t p = (size_t)py * width + px;
        for (int c = 0; c < channels; ++c)
          render[p * channels + c] = (float)(C[c] + (background ? Tr * background[c] : 0));
        alphas[p] = (float)(1 - Tr);
        if (want_margins) {
          ex.margins[p] = (float)m_a;
          ex.margins[n_px + p] = (float)m_t;
          ex.margins[2 * n_px + p] = (float)m_s;
          ex.margins[3 * n_px + p] = (float)m_z;
          if (want_fw) { ex.flip_weight[p] = (float)fw; t_at_min[p] = (float)t_min; }
          if (ex.noise_weight) ex.noise_weight[p] = (float)noise;
        }
        if (want_bwd) { last_idx[p] = last; t_final[p] = (double)Tr; }
      }
  }
  if (counters) { counters[0] = n_vis; counters[1] = evals; }

  // pixels a per-Gaussian knife edge can reach (gaussian_edge_mask)
  if (want_edges) {
    std::memset(ex.edge_mask, 0, n_px);
    for (int g = 0; g < n; ++g) {
      const Splat<F>& s = sp[g];
      if (!s.unsure) continue;
      for (int ty = s.oy0; ty < s.oy1; ++ty)
        for (int tx = s.ox0; tx < s.ox1; ++tx) {
          if (s.inner_on && tx >= s.ix0 && tx < s.ix1 && ty >= s.iy0 && ty < s.iy1) continue;
          for (int py = ty * T; py < std::min((ty + 1) * T, height); ++py)
            for (int px = tx * T; px < std::min((tx + 1) * T, width); ++px) {
              F dx = s.mx - (px + F(0.5)), dy = s.my - (py + F(0.5));
              F sigma = F(0.5) * (s.ca * dx * dx + s.cc * dy * dy) + s.cb * dx * dy;
              F alpha = std::min(F(0.999), s.opac * std::exp(-sigma));
              if (alpha >= F((1.0 - 1e-3) / 255.0)) {
                const size_t p = (size_t)py * width + px;
                ex.edge_mask[p] = 1;
                if (want_fw) {     // the Gaussian may or may not be in this tile's list: worth alpha (T <= 1), and the
                                   // pixel's closest T threshold moves by the factor (1 - alpha)
                  ex.flip_weight[p] += (float)alpha;
                  if ((double)ex.margins[n_px + p] < fe_t + (double)alpha) ex.flip_weight[p] += t_at_min[p];
                }
              }
            }
        }
    }
    if (ex.n_edge) *ex.n_edge = n_edge;
  }
  if (want_margins && want_edges && ex.touched && ex.flip_eps) {
    std::memset(ex.touched, 0, (size_t)n);
    const bool want_budget = ex.budget != nullptr && ex.v_render != nullptr && ex.flip_weight != nullptr;
    double fmax[4] = {0, 0, 0, 0};
    if (want_budget) {
      std::fill(ex.budget, ex.budget + 4 * (size_t)n, 0.0);
      for (int g = 0; g < n; ++g)
        if (sp[g].radius > 0)
          for (int c = 0; c < channels; ++c) fmax[c] = std::max(fmax[c], std::abs((double)feat[(size_t)g * channels + c]));
    }
#pragma omp parallel for schedule(dynamic, 16)
    for (long long t = 0; t < (long long)tw * th; ++t) {
      const int tx = (int)(t % tw), ty = (int)(t / tw);
      for (int py = ty * T; py < std::min((ty + 1) * T, height); ++py)
        for (int px = tx * T; px < std::min((tx + 1) * T, width); ++px) {
          const size_t p = (size_t)py * width + px;
          const bool could_flip = ex.edge_mask[p] || ex.margins[p] < ex.flip_eps[0] || ex.margins[n_px + p] < ex.flip_eps[1] ||
                                  ex.margins[2 * n_px + p] < ex.flip_eps[2] || ex.margins[3 * n_px + p] < ex.flip_eps[3];
          if (!could_flip) continue;
          const F fx = px + F(0.5), fy = py + F(0.5);
          double vabs = 0, vmax = 0, fwp = 0;
          if (want_budget) {
            fwp = (double)ex.flip_weight[p];
            double va = ex.v_alpha ? std::abs((double)ex.v_alpha[p]) : 0.0;
            for (int c = 0; c < channels; ++c) {
              const double vc = std::abs((double)ex.v_render[p * channels + c]);
              vabs += 2.0 * fmax[c] * vc;
              vmax = std::max(vmax, vc);
              if (background) va += std::abs((double)background[c]) * vc;
            }
            vabs += va;
          }
          for (long long i = tstart[t]; i < tstart[t + 1]; ++i) {
            const int g = ids[perm[i]];
            const Splat<F>& s = sp[g];
            F dx = s.mx - fx, dy = s.my - fy;
            F sigma = F(0.5) * (s.ca * dx * dx + s.cc * dy * dy) + s.cb * dx * dy;
            const F vis = std::exp(-sigma), ov = s.opac * vis;
            if (ov >= F(0.5 / 255.0)) {
              ex.touched[g] = 1;     // (benign race: all writers store 1)
              if (want_budget && fwp > 0) {
                // g's OWN decision at p within eps of flipping (alpha or sigma test): its whole term at this pixel comes or
                // goes -- |d loss / d alpha_g| <= T_g vabs with T_g <= flip_weight / alpha_g (its alpha_g T_g is in the
                // weight), so the factors are flip_weight vabs x {geometry, 1 / opacity}, not ov times that
                const double S_abs = 0.5 * (std::abs((double)s.ca) * dx * dx + std::abs((double)s.cc) * dy * dy) + std::abs((double)(s.cb * dx * dy));
                const bool self = std::max(std::abs((double)ov * 255.0 - 1.0) - 1e-6 * S_abs, 0.0) < (double)ex.flip_eps[0] ||
                                  (S_abs > 0 && std::abs((double)sigma) / S_abs < (double)ex.flip_eps[2]);
                const double ra = 1.0 / (1.0 - std::min(0.999, (double)ov)), A = fwp * vabs * ra;
                const double bs = (self ? 1.0 : (double)ov) * A, bo = self ? A / std::max(1e-30, (double)s.opac) : (double)vis * A;
                const double gm = std::max(std::abs((double)(s.ca * dx + s.cb * dy)), std::abs((double)(s.cb * dx + s.cc * dy)));
                const double gc = std::max(std::max(0.5 * (double)(dx * dx), std::abs((double)(dx * dy))), 0.5 * (double)(dy * dy));
                double* b = ex.budget + 4 * (size_t)g;
#pragma omp atomic
                b[0] += bs * gm;
#pragma omp atomic
                b[1] += bs * gc;
#pragma omp atomic
                b[2] += fwp * vmax * ra;
#pragma omp atomic
                b[3] += bo;
              }
            }
          }
        }
    }
  }

  // A.2 step 10: backward of the blend, Gaussian-outer per tile, fp64 sums
  if (want_bwd) {
    std::fill(ex.g_means2d, ex.g_means2d + 2 * (size_t)n, 0.0);
    std::fill(ex.g_conics, ex.g_conics + 3 * (size_t)n, 0.0);
    std::fill(ex.g_feats, ex.g_feats + (size_t)channels * n, 0.0);
    std::fill(ex.g_opac, ex.g_opac + (size_t)n, 0.0);
#pragma omp parallel for schedule(dynamic, 4)
    for (long long t = 0; t < (long long)tw * th; ++t) {
      const int tx = (int)(t % tw), ty = (int)(t / tw);
      const int x_lo = tx * T, x_hi = std::min((tx + 1) * T, width), y_lo = ty * T, y_hi = std::min((ty + 1) * T, height);
      double Tcur[256], Tfin[256], Sbuf[256][4];
      long long hi = -1;
      for (int py = y_lo; py < y_hi; ++py)
        for (int px = x_lo; px < x_hi; ++px) {
          const int k = (py - y_lo) * T + (px - x_lo);
          const size_t p = (size_t)py * width + px;
          Tfin[k] = Tcur[k] = t_final[p];
          for (int c = 0; c < 4; ++c) Sbuf[k][c] = 0.0;
          hi = std::max(hi, last_idx[p]);
        }
      for (long long i = hi; i >= tstart[t]; --i) {
        const int g = ids[perm[i]];
        const Splat<F>& s = sp[g];
        double gx = 0, gy = 0, gca = 0, gcb = 0, gcc = 0, gop = 0, gf[4] = {0, 0, 0, 0};
        bool any = false;
        for (int py = y_lo; py < y_hi; ++py)
          for (int px = x_lo; px < x_hi; ++px) {
            const size_t p = (size_t)py * width + px;
            if (i > last_idx[p]) continue;
            const int k = (py - y_lo) * T + (px - x_lo);
            const double dx = (double)s.mx - (px + 0.5), dy = (double)s.my - (py + 0.5);
            const double sigma = 0.5 * ((double)s.ca * dx * dx + (double)s.cc * dy * dy) + (double)s.cb * dx * dy;
            const double vis = std::exp(-sigma), ov = (double)s.opac * vis;
            const double alpha = std::min(0.999, ov);
            if (sigma < 0 || alpha < thr_alpha) continue;
            any = true;
            const double ra = 1.0 / (1.0 - alpha);
            Tcur[k] *= ra;
            const double Tk = Tcur[k], fac = alpha * Tk;
            double v_alpha = 0;
            for (int c = 0; c < channels; ++c) {
              const double vc = ex.v_render[p * channels + c], fc = (double)feat[(size_t)g * channels + c];
              gf[c] += fac * vc;
              v_alpha += (fc * Tk - Sbuf[k][c] * ra) * vc;
              Sbuf[k][c] += fc * fac;
            }
            double va = ex.v_alpha[p];
            if (background)
              for (int c = 0; c < channels; ++c) va -= (double)background[c] * ex.v_render[p * channels + c];
            v_alpha += Tfin[k] * ra * va;
            if (ov <= 0.999) {
              const double v_sigma = -ov * v_alpha;
              gca += 0.5 * v_sigma * dx * dx;
              gcb += v_sigma * dx * dy;
              gcc += 0.5 * v_sigma * dy * dy;
              gx += v_sigma * ((double)s.ca * dx + (double)s.cb * dy);
              gy += v_sigma * ((double)s.cb * dx + (double)s.cc * dy);
              gop += vis * v_alpha;
            }
          }
        if (!any) continue;
#pragma omp atomic
        ex.g_means2d[2 * (size_t)g] += gx;
#pragma omp atomic
        ex.g_means2d[2 * (size_t)g + 1] += gy;
#pragma omp atomic
        ex.g_conics[3 * (size_t)g] += gca;
#pragma omp atomic
        ex.g_conics[3 * (size_t)g + 1] += gcb;
#pragma omp atomic
        ex.g_conics[3 * (size_t)g + 2] += gcc;
#pragma omp atomic
        ex.g_opac[g] += gop;
        for (int c = 0; c < channels; ++c) {
#pragma omp atomic
          ex.g_feats[(size_t)g * channels + c] += gf[c];
        }
      }
    }
  }
  return n_isect;
}

}  // namespace

// Whole forward frame for one camera, fp32 arithmetic (the timed CPU baseline).  feat layout:
// channels = 3 (rgb) or 4 (rgb + depth).  Returns the number of tile intersections;
// counters[0] = visible Gaussians, counters[1] = pixel-Gaussian pair evaluations.
extern "C" long long gs_cpu_render(int n, const float* means, const float* quats,
                                   const float* scales, const float* opacities, int sh_degree,
                                   int coeff_stride, const float* sh, const float* viewmat,
                                   const float* K, int width, int height, float eps2d,
                                   float near_p, float far_p, float radius_clip, int channels,
                                   const float* background, int n_threads, float* render,
                                   float* alphas, long long* counters) {
  return render_impl<float>(n, means, quats, scales, opacities, sh_degree, coeff_stride, sh, viewmat,
                            K, width, height, eps2d, near_p, far_p, radius_clip, channels,
                            background, n_threads, render, alphas, counters, Extras{});
}

// The same frame in fp64 (inputs are the fp32 arrays the GPU gets).  Optional outputs (null to skip):
// margins [4,H,W] + edge_mask [H,W] + n_edge (+ flip_weight [H,W] for the thresholds flip_eps[4], touched [N] and, with the
// cotangents, budget [N,4]: see Extras); blend backward given v_render [H,W,ch] / v_alpha [H,W]
// into g_means2d [N,2], g_conics [N,3], g_feats [N,ch], g_opac [N]; the projected quantities
// o_means2d / o_conics / o_feats / o_radii.
extern "C" long long gs_cpu_render_f64(int n, const float* means, const float* quats,
                                       const float* scales, const float* opacities, int sh_degree,
                                       int coeff_stride, const float* sh, const float* viewmat,
                                       const float* K, int width, int height, float eps2d,
                                       float near_p, float far_p, float radius_clip, int channels,
                                       const float* background, int n_threads, float* render,
                                       float* alphas, long long* counters, float* margins,
                                       uint8_t* edge_mask, long long* n_edge, const float* v_render,
                                       const float* v_alpha, double* g_means2d, double* g_conics,
                                       double* g_feats, double* g_opac, double* o_means2d,
                                       double* o_conics, double* o_feats, int32_t* o_radii,
                                       float* flip_weight, const float* flip_eps, uint8_t* touched,
                                       double* budget, const float* thresholds, int radius_rule) {
  Extras ex;
  ex.radius_rule = radius_rule;
  ex.thresholds = thresholds;
  ex.touched = touched;
  ex.budget = budget;
  ex.margins = margins; ex.edge_mask = edge_mask; ex.n_edge = n_edge;
  ex.flip_weight = flip_weight; ex.flip_eps = flip_eps;
  ex.v_render = v_render; ex.v_alpha = v_alpha;
  ex.g_means2d = g_means2d; ex.g_conics = g_conics; ex.g_feats = g_feats; ex.g_opac = g_opac;
  ex.o_means2d = o_means2d; ex.o_conics = o_conics; ex.o_feats = o_feats; ex.o_radii = o_radii;
  return render_impl<double>(n, means, quats, scales, opacities, sh_degree, coeff_stride, sh, viewmat,
                             K, width, height, eps2d, near_p, far_p, radius_clip, channels,
                             background, n_threads, render, alphas, counters, ex);
}

// A.2 steps 9-10 alone, in fp64, on GIVEN projected quantities and depth-ordered tile lists (Extras: stage-isolated blend):
// means2d [N,2], conics [N,3], opacities [N], feats [N,ch], depths [N] (nullable), flatten_ids [tile_offsets[tiles]],
// tile_offsets [ceil(W/16) ceil(H/16) + 1].  Optional outputs as gs_cpu_render_f64 (no edge mask: list membership is given).
extern "C" long long gs_cpu_blend_f64(int n, const float* means2d, const float* conics, const float* opacities,
                                      const float* feats, const float* depths, const int32_t* flatten_ids,
                                      const int32_t* tile_offsets, int width, int height, int channels,
                                      const float* background, int n_threads, float* render, float* alphas,
                                      long long* counters, float* margins, const float* v_render, const float* v_alpha,
                                      double* g_means2d, double* g_conics, double* g_feats, double* g_opac,
                                      float* flip_weight, const float* flip_eps, uint8_t* edge_mask, uint8_t* touched,
                                      double* budget, const float* thresholds, float* noise_weight) {
  Extras ex;
  ex.noise_weight = noise_weight;
  ex.in_means2d = means2d; ex.in_conics = conics; ex.in_opac = opacities; ex.in_feats = feats; ex.in_depths = depths;
  ex.in_flatten_ids = flatten_ids; ex.in_tile_offsets = tile_offsets;
  ex.thresholds = thresholds;
  ex.margins = margins; ex.flip_weight = flip_weight; ex.flip_eps = flip_eps;
  ex.edge_mask = edge_mask; ex.touched = touched; ex.budget = budget;
  ex.v_render = v_render; ex.v_alpha = v_alpha;
  ex.g_means2d = g_means2d; ex.g_conics = g_conics; ex.g_feats = g_feats; ex.g_opac = g_opac;
  return render_impl<double>(n, nullptr, nullptr, nullptr, opacities, 0, 0, nullptr, nullptr, nullptr, width, height, 0.f,
                             0.f, 0.f, 0.f, channels, background, n_threads, render, alphas, counters, ex);
}

// ... and the same blend in plain fp32 (the reference's formulas, float arithmetic) on the same given inputs: what "an fp32
// blend" makes of them, for telling a kernel's own rounding from what any fp32 evaluation does.
extern "C" long long gs_cpu_blend_f32(int n, const float* means2d, const float* conics, const float* opacities,
                                      const float* feats, const int32_t* flatten_ids, const int32_t* tile_offsets, int width,
                                      int height, int channels, const float* background, int n_threads, float* render,
                                      float* alphas) {
  Extras ex;
  ex.in_means2d = means2d; ex.in_conics = conics; ex.in_opac = opacities; ex.in_feats = feats;
  ex.in_flatten_ids = flatten_ids; ex.in_tile_offsets = tile_offsets;
  return render_impl<float>(n, nullptr, nullptr, nullptr, opacities, 0, 0, nullptr, nullptr, nullptr, width, height, 0.f,
                            0.f, 0.f, 0.f, channels, background, n_threads, render, alphas, nullptr, ex);
}

extern "C" int gs_cpu_max_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
