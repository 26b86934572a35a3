// composite.hip -- row (f2): depth-tested compositing of the 3DGS background with an opaque
// foreground layer (a simulator / mesh render), per pixel, for gfx950.  HBM-bound streaming:
// 37 B read + 16 B written per pixel, 16-byte accesses where the layout allows.
//
// The reference names this step ("Holistic Scene Augmentation -> Simulated Data",
// /root/reference/README.md:53-56, imgs/pipeline.png) but has not released it; the rule
// implemented here is the standard single-surface z-test against the splats' expected depth:
//   foreground present and (background empty or fg_depth <= bg_depth):  out = fg,            depth = fg_depth
//   foreground present but behind the splats:                            out = bg + (1-a) fg, depth = bg_depth
//   no foreground:                                                       out = bg + (1-a) backdrop,
//                                                                        depth = a > 0 ? bg_depth : +inf
// bg_rgb is the rasteriser's premultiplied accumulation, a its alpha, bg_depth its "ED" channel
// (z-depth, the convention of nerf2physic_utils.py:120-146).
#include "mgs_common.h"
#include "dataset_pixel.h"

namespace mgs {
namespace {

__global__ __launch_bounds__(256) void composite_kernel(
    int n_px, const float* __restrict__ bg_rgb, const float* __restrict__ bg_alpha,
    const float* __restrict__ bg_depth, const float* __restrict__ fg_rgb,
    const float* __restrict__ fg_depth, const uint8_t* __restrict__ fg_mask,
    const float* __restrict__ backdrop, float* __restrict__ out_rgb,
    float* __restrict__ out_depth) {
  const float bd0 = backdrop ? backdrop[0] : 0.f, bd1 = backdrop ? backdrop[1] : 0.f,
              bd2 = backdrop ? backdrop[2] : 0.f;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < n_px; p += gridDim.x * 256) {
    const float a = bg_alpha[p], zb = bg_depth[p], zf = fg_depth[p];
    const bool has_fg = fg_mask ? fg_mask[p] != 0 : (zf > 0.f && zf < INFINITY);
    const float b0 = bg_rgb[3 * (size_t)p], b1 = bg_rgb[3 * (size_t)p + 1], b2 = bg_rgb[3 * (size_t)p + 2];
    float f0 = bd0, f1 = bd1, f2 = bd2;
    if (has_fg) {
      f0 = fg_rgb[3 * (size_t)p]; f1 = fg_rgb[3 * (size_t)p + 1]; f2 = fg_rgb[3 * (size_t)p + 2];
    }
    const bool front = has_fg && (!(a > 0.f) || zf <= zb);
    const float t = front ? 0.f : 1.f;          // weight of the splat layer
    const float w = front ? 1.f : 1.f - a;      // weight of the foreground / backdrop
    out_rgb[3 * (size_t)p] = t * b0 + w * f0;
    out_rgb[3 * (size_t)p + 1] = t * b1 + w * f1;
    out_rgb[3 * (size_t)p + 2] = t * b2 + w * f2;
    out_depth[p] = front ? zf : (a > 0.f ? zb : (has_fg ? zf : INFINITY));
  }
}

}  // namespace
}  // namespace mgs

using namespace mgs;

extern "C" int mgs_composite_over(int n_px, const float* bg_rgb, const float* bg_alpha,
                                  const float* bg_depth, const float* fg_rgb,
                                  const float* fg_depth, const uint8_t* fg_mask,
                                  const float* backdrop, float* out_rgb, float* out_depth,
                                  mgs_stream_t stream) {
  MGS_REQUIRE(n_px >= 0, "composite_over: negative pixel count");
  if (n_px == 0) return MGS_OK;
  MGS_REQUIRE(bg_rgb && bg_alpha && bg_depth && fg_rgb && fg_depth && out_rgb && out_depth,
              "composite_over: null pointer");
  unsigned grid = div_up((unsigned)n_px, 256u);
  if (grid > 2048u) grid = 2048u;               // grid-stride beyond 256 CUs x 8 workgroups
  hipLaunchKernelGGL(composite_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n_px, bg_rgb,
                     bg_alpha, bg_depth, fg_rgb, fg_depth, fg_mask, backdrop, out_rgb, out_depth);
  return check_launch("composite_over");
}

// ---- 8-bit frames for the dataset writer / the multi-GPU gather -----------------------------
// out = round(255 * clamp(rgb + (1 - alpha) * background, 0, 1)): splatfacto's post-processing
//  followed by the quantisation every image file format applies anyway; a quarter
// of the bytes to gather over xGMI or to copy to the host.  HBM-bound: 16 B read, 3 B written.
namespace mgs {
namespace {
// Four pixels per thread when rgb is packed [P,3]: three 16-byte loads, one 16-byte alpha load,
// three 4-byte stores (12 output bytes); the generic path (strided rgb, tail) goes pixel by pixel.
__global__ __launch_bounds__(256) void frame_to_u8_kernel(int n_px, const float* __restrict__ rgb,
                                                          int rgb_stride, const float* __restrict__ alpha,
                                                          const float* __restrict__ background,
                                                          uint8_t* __restrict__ out, int quads) {
  const float b0 = background ? background[0] : 0.f, b1 = background ? background[1] : 0.f,
              b2 = background ? background[2] : 0.f;
  const int n_quads = quads ? n_px / 4 : 0;
  for (int q = blockIdx.x * 256 + threadIdx.x; q < n_quads; q += gridDim.x * 256) {
    const float4* c = reinterpret_cast<const float4*>(rgb) + 3 * (size_t)q;
    const float4 c0 = c[0], c1 = c[1], c2 = c[2];
    const float4 a = reinterpret_cast<const float4*>(alpha)[q];
    const float w0 = 1.f - a.x, w1 = 1.f - a.y, w2 = 1.f - a.z, w3 = 1.f - a.w;
    // pixels: (c0.x c0.y c0.z) (c0.w c1.x c1.y) (c1.z c1.w c2.x) (c2.y c2.z c2.w)
    const uint32_t o0 = quant8(c0.x + w0 * b0) | quant8(c0.y + w0 * b1) << 8 | quant8(c0.z + w0 * b2) << 16 |
                        quant8(c0.w + w1 * b0) << 24;
    const uint32_t o1 = quant8(c1.x + w1 * b1) | quant8(c1.y + w1 * b2) << 8 | quant8(c1.z + w2 * b0) << 16 |
                        quant8(c1.w + w2 * b1) << 24;
    const uint32_t o2 = quant8(c2.x + w2 * b2) | quant8(c2.y + w3 * b0) << 8 | quant8(c2.z + w3 * b1) << 16 |
                        quant8(c2.w + w3 * b2) << 24;
    uint32_t* o = reinterpret_cast<uint32_t*>(out) + 3 * (size_t)q;
    o[0] = o0; o[1] = o1; o[2] = o2;
  }
  for (int p = n_quads * 4 + blockIdx.x * 256 + threadIdx.x; p < n_px; p += gridDim.x * 256) {
    const float w = 1.f - alpha[p];
    const float* c = rgb + (size_t)p * rgb_stride;
    out[3 * (size_t)p + 0] = (uint8_t)quant8(c[0] + w * b0);
    out[3 * (size_t)p + 1] = (uint8_t)quant8(c[1] + w * b1);
    out[3 * (size_t)p + 2] = (uint8_t)quant8(c[2] + w * b2);
  }
}
}  // namespace
}  // namespace mgs

extern "C" int mgs_frame_to_u8(int n_px, const float* rgb, int rgb_stride, const float* alpha,
                               const float* background, uint8_t* out, mgs_stream_t stream) {
  MGS_REQUIRE(n_px >= 0 && rgb_stride >= 3, "frame_to_u8: bad sizes");
  if (n_px == 0) return MGS_OK;
  MGS_REQUIRE(rgb && alpha && out, "frame_to_u8: null pointer");
  const bool quads = rgb_stride == 3 && ((uintptr_t)rgb & 15) == 0 && ((uintptr_t)alpha & 15) == 0 &&
                     ((uintptr_t)out & 3) == 0;
  unsigned grid = div_up((unsigned)(quads ? (n_px + 3) / 4 : n_px), 256u);
  if (grid > 4096u) grid = 4096u;
  hipLaunchKernelGGL(frame_to_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n_px, rgb,
                     rgb_stride, alpha, background, out, quads ? 1 : 0);
  return check_launch("frame_to_u8");
}


// ---- dataset frames in the layout the reference's readers expect ------------------------------------
// /root/reference/Articulation/utils/nerf2physic_utils.py: load_images (:84-101) opens RGBA images and takes
// alpha > 0 as the object mask; load_depths (:104-118) opens [H,W,1] .npy.gz RAY DISTANCES and converts them
// with distance_to_depth (:135-146), whose rays go through INTEGER pixel coordinates (np.arange, no + 0.5):
//     distance = z * || K^-1 (x, y, 1) ||.
// One pass over the frame: 20 B read, 4 + 4 (or 8) B written per pixel.
namespace mgs {
namespace {
template <typename DistT>
__global__ __launch_bounds__(256) void frame_to_dataset_kernel(int width, int height, const float* __restrict__ colors,
                                                               int stride, const float* __restrict__ alpha,
                                                               const float* __restrict__ background, KInv ki,
                                                               uint32_t* __restrict__ rgba, DistT* __restrict__ dist) {
#pragma clang fp contract(off)      // the test-side NumPy restatement has no fused multiply-add (see oracle/dataset_np.py)
  const float b0 = background ? background[0] : 0.f, b1 = background ? background[1] : 0.f,
              b2 = background ? background[2] : 0.f;
  const int n_px = width * height;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < n_px; p += gridDim.x * 256) {
    const float* c = colors + (size_t)p * stride;
    const float a = alpha[p], w = 1.f - a;
    if (rgba) {
      // alpha > 0 must survive the quantisation: the reader's mask is A > 0
      const uint32_t A = a > 0.f ? max(1u, quant8(a)) : 0u;
      rgba[p] = quant8(c[0] + w * b0) | quant8(c[1] + w * b1) << 8 | quant8(c[2] + w * b2) << 16 | A << 24;
    }
    if (dist)                                                   // one rounding (fp32 / fp16 output) or none (fp64)
      dist[p] = (DistT)dataset_distance(ki, p % width, p / width, c[stride - 1]);
  }
}
}  // namespace
}  // namespace mgs

extern "C" int mgs_frame_to_dataset(int width, int height, const float* colors, int color_stride,
                                    const float* alpha, const float* background, const double* Kinv_host,
                                    uint8_t* rgba, void* distance, int distance_f64, mgs_stream_t stream) {
  MGS_REQUIRE(width > 0 && height > 0 && color_stride >= 3, "frame_to_dataset: bad sizes");
  MGS_REQUIRE(colors && alpha && (rgba || distance), "frame_to_dataset: null pointer");
  MGS_REQUIRE(!distance || (Kinv_host && color_stride >= 4),
              "frame_to_dataset: the distance map needs K^-1 and a depth channel (the last of >= 4)");
  MGS_REQUIRE(((uintptr_t)rgba & 3) == 0, "frame_to_dataset: rgba must be 4-byte aligned");
  KInv ki;
  for (int i = 0; i < 9; ++i) ki.m[i] = Kinv_host ? Kinv_host[i] : 0.0;
  unsigned grid = div_up((unsigned)(width * height), 256u);
  if (grid > 4096u) grid = 4096u;
  hipStream_t s = (hipStream_t)stream;
  MGS_REQUIRE(distance_f64 >= 0 && distance_f64 <= 2, "frame_to_dataset: distance type %d not in {0: f32, 1: f64, 2: f16}", distance_f64);
  if (distance_f64 == 2)
    hipLaunchKernelGGL(frame_to_dataset_kernel<_Float16>, dim3(grid), dim3(256), 0, s, width, height, colors, color_stride,
                       alpha, background, ki, reinterpret_cast<uint32_t*>(rgba), static_cast<_Float16*>(distance));
  else if (distance_f64)
    hipLaunchKernelGGL(frame_to_dataset_kernel<double>, dim3(grid), dim3(256), 0, s, width, height, colors, color_stride,
                       alpha, background, ki, reinterpret_cast<uint32_t*>(rgba), static_cast<double*>(distance));
  else
    hipLaunchKernelGGL(frame_to_dataset_kernel<float>, dim3(grid), dim3(256), 0, s, width, height, colors, color_stride,
                       alpha, background, ki, reinterpret_cast<uint32_t*>(rgba), static_cast<float*>(distance));
  return check_launch("frame_to_dataset");
}
