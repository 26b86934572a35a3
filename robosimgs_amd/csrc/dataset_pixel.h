// dataset_pixel.h -- one pixel of the dataset frame the reference's readers open (RGBA8 + ray distance; composite.hip:
// mgs_frame_to_dataset says which lines of /root/reference/Articulation/utils/nerf2physic_utils.py read it).  Shared by the
// standalone conversion kernel and by the raster forward's dataset epilogue, so that a frame converted on its way out of
// the raster is the very bytes the conversion kernel gives on the float frame.
#ifndef MGS_DATASET_PIXEL_H_
#define MGS_DATASET_PIXEL_H_

#include "mgs_common.h"

namespace mgs {

struct KInv { double m[9]; };            // K^-1, row-major

__device__ __forceinline__ uint32_t quant8(float v) {
  return (uint32_t)__float2int_rn(255.f * fminf(fmaxf(v, 0.f), 1.f));
}

// alpha > 0 must survive the quantisation: the reader's mask is A > 0
__device__ __forceinline__ uint32_t dataset_rgba(float r, float g, float b, float alpha) {
  const uint32_t A = alpha > 0.f ? max(1u, quant8(alpha)) : 0u;
  return quant8(r) | quant8(g) << 8 | quant8(b) << 16 | A << 24;
}

// ray distance z * || K^-1 (x, y, 1) || through INTEGER pixel coordinates, in fp64, no fused multiply-add (the test-side
// NumPy restatement has none, see oracle/dataset_np.py); the caller rounds once to its output type
__device__ __forceinline__ double dataset_distance(const KInv& ki, int px, int py, float depth) {
#pragma clang fp contract(off)
  const double x = (double)px, y = (double)py;
  const double rx = (x * ki.m[0] + y * ki.m[1]) + ki.m[2], ry = (x * ki.m[3] + y * ki.m[4]) + ki.m[5],
               rz = (x * ki.m[6] + y * ki.m[7]) + ki.m[8];
  const double norm = sqrt((rx * rx + ry * ry) + rz * rz);
  return (double)depth * norm;
}

// what the raster forward's dataset epilogue needs (mgs_rasterize_fwd ds_* arguments); rgba == nullptr: no dataset output
struct DatasetOut {
  uint32_t* rgba;      // [H,W] packed RGBA8
  void* dist;          // [H,W] f32 / f64 / f16 by `type` (nullable)
  int type;            // 0 f32, 1 f64, 2 f16
  KInv ki;
};

__device__ __forceinline__ void dataset_store(const DatasetOut& ds, size_t p, int px, int py, float r, float g, float b,
                                              float depth, float alpha) {
  ds.rgba[p] = dataset_rgba(r, g, b, alpha);
  if (ds.dist) {
    const double d = dataset_distance(ds.ki, px, py, depth);
    if (ds.type == 2) static_cast<_Float16*>(ds.dist)[p] = (_Float16)d;
    else if (ds.type == 1) static_cast<double*>(ds.dist)[p] = d;
    else static_cast<float*>(ds.dist)[p] = (float)d;
  }
}

}  // namespace mgs
#endif
