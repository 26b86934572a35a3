// tile_rect.h -- which tiles a projected Gaussian is binned into (shared by the binning and by the
// fused projection kernel, which seeds the binning with each Gaussian's rectangle and tile count).
#ifndef MGS_TILE_RECT_H_
#define MGS_TILE_RECT_H_

#include "mgs_common.h"

namespace mgs {

struct TileRect { int x0, y0, w, h; };

// A.2 step 7: axis-aligned tile rectangle of mean2d +- (radius_x, radius_y) -- the square mean2d +- radius under the
// classic radius rule, gsplat >= 1.5's per-axis box under MGS_RADIUS_OPACITY_AWARE (mgs_math.h)
__device__ __forceinline__ TileRect tile_rect(float mx, float my, int radius_x, int radius_y, float tile_size,
                                              int tile_w, int tile_h) {
  TileRect r;
  float trx = (float)radius_x / tile_size, try_ = (float)radius_y / tile_size;
  float tx = mx / tile_size, ty = my / tile_size;
  int x0 = min(max(0, (int)floorf(tx - trx)), tile_w);
  int x1 = min(max(0, (int)ceilf(tx + trx)), tile_w);
  int y0 = min(max(0, (int)floorf(ty - try_)), tile_h);
  int y1 = min(max(0, (int)ceilf(ty + try_)), tile_h);
  r.x0 = x0; r.y0 = y0; r.w = x1 - x0; r.h = y1 - y0;
  return r;
}
__device__ __forceinline__ TileRect tile_rect(float mx, float my, int radius, float tile_size,
                                              int tile_w, int tile_h) {
  return tile_rect(mx, my, radius, radius, tile_size, tile_w, tile_h);
}

// Tiles that hold a pixel centre the Gaussian can reach with alpha >= 1/255: the bounding box of
// the ellipse sigma <= ln(255 opacity) (half extents sqrt(2 lim Sigma_xx), sqrt(2 lim Sigma_yy)),
// intersected with the classic rectangle `r`.  Every dropped (tile, Gaussian) pair fails the
// raster's alpha test at all 256 pixels, so the image and the gradients do not change by a bit
// (tests/test_gpu_forward.py, test_gpu_backward.py) while the lists shrink by ~27 % at config 2.
// Sigma = conic^-1 is taken from the fp32 conic the blend itself evaluates; a*c - b*b cancels
// catastrophically for elongated footprints, so the determinant is Kahan-compensated (fma
// residuals: exact to an ulp of the true value) and the extents carry a 1e-4 relative margin.
// `lim` carries the same fp32 slack as the raster's own quadrant cull (raster_common.h).
__device__ __forceinline__ TileRect tighten_rect(TileRect r, float mx, float my, float a, float b,
                                                 float c, float opac, float tile_size) {
  if (!(opac >= 1.0f / 255.0f)) { r.w = 0; r.h = 0; return r; }   // alpha <= opacity < 1/255
  // det = a c - b b with Kahan's compensated 2x2 determinant: exact products via fma residuals
  const float w = b * b, e = fmaf(-b, b, w), f = fmaf(a, c, -w);
  const float det = f + e;
  if (!(det > 0.f) || !(a > 0.f) || !(c > 0.f)) return r;          // degenerate: keep classic
  const float inv_det = 1.0f / det;
  const float sxx = c * inv_det, syy = a * inv_det;
  const float thr = __logf(255.0f * opac);
  const float e2 = 2.0f * (thr + 0.05f) * (sxx + syy);
  const float slack = 0.05f + 8e-6f * (fabsf(a) + fabsf(c) + 2.f * fabsf(b)) * (e2 + 512.f);
  const float lim = 2.0f * (thr + slack);
  const float ex = sqrtf(lim * sxx) * 1.0001f + 0.01f;
  const float ey = sqrtf(lim * syy) * 1.0001f + 0.01f;
  // pixel centres are i + 0.5: first / last pixel column and row inside the box
  const float plx = ceilf(mx - ex - 0.5f), phx = floorf(mx + ex - 0.5f);
  const float ply = ceilf(my - ey - 0.5f), phy = floorf(my + ey - 0.5f);
  if (!(phx >= plx) || !(phy >= ply)) {
    if (phx < plx || phy < ply) { r.w = 0; r.h = 0; }               // no pixel centre inside
    return r;                                                      // NaN: keep classic
  }
  const int x0 = max(r.x0, (int)fmaxf(floorf(plx / tile_size), -1.f));
  const int y0 = max(r.y0, (int)fmaxf(floorf(ply / tile_size), -1.f));
  const int x1 = min(r.x0 + r.w, (int)fminf(floorf(phx / tile_size), 65535.f) + 1);
  const int y1 = min(r.y0 + r.h, (int)fminf(floorf(phy / tile_size), 65535.f) + 1);
  if (x1 <= x0 || y1 <= y0) { r.w = 0; r.h = 0; return r; }
  r.x0 = x0; r.y0 = y0; r.w = x1 - x0; r.h = y1 - y0;
  return r;
}

// (rectangle, count) record of one Gaussian: x0 | y0 << 10 | max(w, 1) << 20, and w * h
__device__ __forceinline__ uint2 pack_tile_rect(const TileRect& r) {
  return make_uint2((uint32_t)r.x0 | ((uint32_t)r.y0 << 10) | ((uint32_t)max(r.w, 1) << 20),
                    (uint32_t)(r.w * r.h));
}
constexpr uint32_t kEmptyTileRect = 1u << 20;     // culled Gaussian: width 1, count 0

}  // namespace mgs
#endif
