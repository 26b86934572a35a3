// points.hip -- row (f4): the reference's point-cloud z-buffer helpers for gfx950
// (/root/reference/Articulation/utils/point_utils.py): pinhole projection of a point cloud
// (:13-26), scatter-min depth map with the winning point per cell (:44-73) and bilinear mask /
// depth lookup per point (:76-111).  All three are HBM-bound streaming / scatter kernels; the
// z-buffer is one 64-bit atomicMin per point on (order-preserving depth bits << 32 | point index),
// which yields torch_scatter's result (first minimal element wins) without a second pass.
#include "mgs_common.h"

namespace mgs {
namespace {

constexpr int kBlock = 256;

// float -> uint32 whose unsigned order is the float order (negative depths included)
__device__ __forceinline__ uint32_t sortable_bits(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_sortable(uint32_t s) {
  return __uint_as_float((s & 0x80000000u) ? (s & 0x7fffffffu) : ~s);
}

__global__ __launch_bounds__(kBlock) void points_project_kernel(
    int n, const float* __restrict__ pts, const float* __restrict__ Kmat,
    const float* __restrict__ c2w, float* __restrict__ uv, float* __restrict__ pnt_cam) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  // c2w row-major 4x4: R = c2w[:3,:3], t = c2w[:3,3];  pnt_cam = R^T (p - t)
  float d0 = pts[3 * (size_t)i] - c2w[3], d1 = pts[3 * (size_t)i + 1] - c2w[7],
        d2 = pts[3 * (size_t)i + 2] - c2w[11];
  float c[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) c[k] = d0 * c2w[k] + d1 * c2w[4 + k] + d2 * c2w[8 + k];
  float x = c[0] / c[2], y = c[1] / c[2], z = c[2] / c[2];     // the reference divides z too (NaN at z == 0)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    uv[3 * (size_t)i + k] = x * Kmat[3 * k] + y * Kmat[3 * k + 1] + z * Kmat[3 * k + 2];
    pnt_cam[3 * (size_t)i + k] = c[k];
  }
}

__global__ __launch_bounds__(kBlock) void zbuffer_init_kernel(int cells, float bg_depth,
                                                              unsigned long long* __restrict__ zbuf) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < cells) zbuf[i] = (unsigned long long)sortable_bits(bg_depth) << 32;
}

__global__ __launch_bounds__(kBlock) void zbuffer_splat_kernel(
    int n, const float* __restrict__ uv, int uv_stride, const float* __restrict__ depth,
    float scale, int cw, int ch, unsigned long long* __restrict__ zbuf) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float d = depth[i];
  if (d != d) return;                                             // NaN never wins a '<'
  // np.round(uv / scale) (half to even), cast to int32, clipped to the cell grid
  float fu = rintf(uv[(size_t)i * uv_stride] / scale), fv = rintf(uv[(size_t)i * uv_stride + 1] / scale);
  int u = fu != fu ? 0 : (int)fminf(fmaxf(fu, 0.f), (float)(cw - 1));
  int v = fv != fv ? 0 : (int)fminf(fmaxf(fv, 0.f), (float)(ch - 1));
  unsigned long long key = ((unsigned long long)sortable_bits(d) << 32) | (unsigned)i;
  atomicMin(&zbuf[(size_t)u * ch + v], key);                       // cell order u * _h + v (:62)
}

__global__ __launch_bounds__(kBlock) void zbuffer_index_kernel(
    int cells, int n, float bg_depth, const unsigned long long* __restrict__ zbuf,
    int64_t* __restrict__ index) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= cells) return;
  unsigned long long k = zbuf[i];
  // a cell still at the background depth was never strictly undercut
  index[i] = (uint32_t)(k >> 32) == sortable_bits(bg_depth) ? (int64_t)n : (int64_t)(uint32_t)k;
}

// nearest-neighbour upsample of the [_w,_h] column-major cells to [h,w] (cv2.resize INTER_NEAREST)
__global__ __launch_bounds__(kBlock) void zbuffer_resolve_kernel(
    int h, int w, int cw, int ch, const unsigned long long* __restrict__ zbuf,
    float* __restrict__ depth_map) {
  int p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= h * w) return;
  int y = p / w, x = p - y * w;
  int sx = min((int)(((long long)x * cw) / w), cw - 1), sy = min((int)(((long long)y * ch) / h), ch - 1);
  depth_map[p] = from_sortable((uint32_t)(zbuf[(size_t)sx * ch + sy] >> 32));
}

// F.grid_sample(img, (uv - [w/2,h/2]) / [w/2,h/2], bilinear, border, align_corners=True)
__device__ __forceinline__ float sample_bilinear(const float* __restrict__ img, int h, int w,
                                                 float u, float v) {
  float hw = 0.5f * (float)w, hh = 0.5f * (float)h;
  float gx = (u - hw) / hw, gy = (v - hh) / hh;
  float x = fminf(fmaxf((gx + 1.f) * 0.5f * (float)(w - 1), 0.f), (float)(w - 1));
  float y = fminf(fmaxf((gy + 1.f) * 0.5f * (float)(h - 1), 0.f), (float)(h - 1));
  float x0 = floorf(x), y0 = floorf(y);
  float fx = x - x0, fy = y - y0;
  int x0i = (int)x0, y0i = (int)y0;
  int x1i = min(x0i + 1, w - 1), y1i = min(y0i + 1, h - 1);
  return img[(size_t)y0i * w + x0i] * (1.f - fx) * (1.f - fy) + img[(size_t)y0i * w + x1i] * fx * (1.f - fy) +
         img[(size_t)y1i * w + x0i] * (1.f - fx) * fy + img[(size_t)y1i * w + x1i] * fx * fy;
}

__global__ __launch_bounds__(kBlock) void points_sample_mask_kernel(
    int n, const float* __restrict__ uv, int uv_stride, const float* __restrict__ mask, int h,
    int w, float thresh, const float* __restrict__ depth_map, const float* __restrict__ pnt_depth,
    float depth_thresh, uint8_t* __restrict__ out) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float u = uv[(size_t)i * uv_stride], v = uv[(size_t)i * uv_stride + 1];
  bool keep = false;
  if (u == u && v == v) {
    keep = sample_bilinear(mask, h, w, u, v) > thresh;
    if (depth_map && pnt_depth)
      keep = keep && fabsf(sample_bilinear(depth_map, h, w, u, v) - pnt_depth[i]) < depth_thresh;
  }
  out[i] = keep ? 1 : 0;
}

}  // namespace
}  // namespace mgs

using namespace mgs;

extern "C" int mgs_points_project(int n, const float* pts, const float* K, const float* c2w,
                                  float* uv, float* pnt_cam, mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0, "points_project: negative point count");
  if (n == 0) return MGS_OK;
  MGS_REQUIRE(pts && K && c2w && uv && pnt_cam, "points_project: null pointer");
  hipLaunchKernelGGL(points_project_kernel, dim3(div_up(n, kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, n, pts, K, c2w, uv, pnt_cam);
  return check_launch("points_project");
}

extern "C" int mgs_points_depth_map(int n, const float* uv, int uv_stride, const float* depth,
                                    int height, int width, int cells_h, int cells_w, float scale,
                                    float bg_depth, float* depth_map, int64_t* index,
                                    void* workspace, size_t* workspace_bytes,
                                    mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && height > 0 && width > 0 && cells_h > 0 && cells_w > 0 && scale > 0.f,
              "points_depth_map: bad sizes");
  MGS_REQUIRE(uv_stride >= 2, "points_depth_map: uv_stride %d < 2", uv_stride);
  MGS_REQUIRE(workspace_bytes, "points_depth_map: workspace_bytes is null");
  const int cells = cells_h * cells_w;
  const size_t need = (size_t)cells * sizeof(unsigned long long);
  if (!workspace) {
    *workspace_bytes = need;
    return MGS_OK;
  }
  if (*workspace_bytes < need)
    return set_error(MGS_ERR_WORKSPACE_TOO_SMALL, "points_depth_map: workspace %zu < %zu bytes",
                     *workspace_bytes, need);
  MGS_REQUIRE((n == 0 || (uv && depth)) && depth_map, "points_depth_map: null pointer");
  hipStream_t s = (hipStream_t)stream;
  unsigned long long* zbuf = static_cast<unsigned long long*>(workspace);
  hipLaunchKernelGGL(zbuffer_init_kernel, dim3(div_up(cells, kBlock)), dim3(kBlock), 0, s, cells,
                     bg_depth, zbuf);
  if (n > 0)
    hipLaunchKernelGGL(zbuffer_splat_kernel, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, n, uv,
                       uv_stride, depth, scale, cells_w, cells_h, zbuf);
  if (index)
    hipLaunchKernelGGL(zbuffer_index_kernel, dim3(div_up(cells, kBlock)), dim3(kBlock), 0, s,
                       cells, n, bg_depth, zbuf, index);
  hipLaunchKernelGGL(zbuffer_resolve_kernel, dim3(div_up(height * width, kBlock)), dim3(kBlock),
                     0, s, height, width, cells_w, cells_h, zbuf, depth_map);
  return check_launch("points_depth_map");
}

extern "C" int mgs_points_sample_mask(int n, const float* uv, int uv_stride, const float* mask,
                                      int height, int width, float thresh,
                                      const float* depth_map, const float* pnt_depth,
                                      float depth_thresh, uint8_t* out, mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && height > 0 && width > 0, "points_sample_mask: bad sizes");
  MGS_REQUIRE(uv_stride >= 2, "points_sample_mask: uv_stride %d < 2", uv_stride);
  if (n == 0) return MGS_OK;
  MGS_REQUIRE(uv && mask && out, "points_sample_mask: null pointer");
  MGS_REQUIRE((depth_map == nullptr) == (pnt_depth == nullptr),
              "points_sample_mask: depth test needs both depth_map and pnt_depth");
  hipLaunchKernelGGL(points_sample_mask_kernel, dim3(div_up(n, kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, n, uv, uv_stride, mask, height, width, thresh, depth_map,
                     pnt_depth, depth_thresh, out);
  return check_launch("points_sample_mask");
}
