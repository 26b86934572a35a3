// frame.hip -- mgs_render_frames: the whole inference path for a BATCH of cameras behind one C call.
//
// Per camera the three stage entry points in their inference-frame form (mgs_project_color_fwd writing only
// depths + packed records + binning seed, mgs_isect_tiles seeded and without tile ids, mgs_rasterize_fwd reading the
// records), enqueued back to back on the caller's stream.  The per-camera intermediates live in the caller's
// workspace and are reused from one camera to the next (stream order makes that safe), so a batch of any size needs
// one camera's worth of scratch; the frames land in render[C,H,W,channels] / alphas[C,H,W].  No host read-back, no
// allocation: a batch is capturable in a hipGraph like a single frame.  What a non-Python host calls to "render these
// cameras", and what rasterization(C > 1) uses instead of 3 C ctypes calls with their tensor bookkeeping.
#include "mgs_common.h"

namespace {
struct FrameWs {
  size_t total, depths, opac, splats, bin_info, bin_sums, flatten, offsets, order, isect;
  size_t isect_bytes;
  FrameWs(int n, uint32_t cap, int n_tiles, bool antialiased, size_t isect_ws) {
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += mgs::align_up(bytes ? bytes : 1, 256); return at; };
    const size_t nn = (size_t)(n > 0 ? n : 1);
    depths = take(nn * 4);
    opac = take(antialiased ? nn * 4 : 0);
    splats = take(nn * 48);
    bin_info = take(nn * 8);
    bin_sums = take(((nn + 63) / 64) * 4);
    flatten = take((size_t)cap * 4);
    offsets = take(((size_t)n_tiles + 1) * 4);
    order = take((((size_t)n_tiles + 3) / 4) * 4);
    isect = take(isect_ws);
    isect_bytes = isect_ws;
    total = o;
  }
};
}  // namespace

extern "C" int mgs_render_frames(int n, const float* means, const float* quats, const float* scales,
                                 const float* opacities, int sh_degree, int coeff_stride, const float* sh_coeffs,
                                 int n_cams, const float* viewmats, const float* Ks, int width, int height,
                                 float eps2d, float near_plane, float far_plane, float radius_clip,
                                 int antialiased, int channels, int flags, const float* backgrounds,
                                 uint32_t isect_capacity, float* render, float* alphas, uint32_t* n_isect,
                                 uint32_t* status, void* workspace, size_t* workspace_bytes, mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && n_cams >= 1 && width > 0 && height > 0, "render_frames: bad sizes");
  MGS_REQUIRE(channels == 3 || channels == 4, "render_frames: channels must be 3 (RGB) or 4 (RGB + depth), got %d", channels);
  MGS_REQUIRE(workspace_bytes, "render_frames: workspace_bytes is null");
  MGS_REQUIRE(isect_capacity > 0, "render_frames: zero capacity");
  const int tile_w = (width + MGS_TILE_SIZE - 1) / MGS_TILE_SIZE, tile_h = (height + MGS_TILE_SIZE - 1) / MGS_TILE_SIZE;
  const int n_tiles = tile_w * tile_h;
  size_t isect_ws = 0;
  int rc = mgs_isect_tiles(n, nullptr, nullptr, nullptr, nullptr, nullptr, MGS_TILE_SIZE, tile_w, tile_h, 0, 1,
                           isect_capacity, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, nullptr, nullptr, &isect_ws, stream);
  if (rc) return rc;
  const FrameWs ws(n, isect_capacity, n_tiles, antialiased != 0, isect_ws);
  if (!workspace) {
    *workspace_bytes = ws.total;
    return MGS_OK;
  }
  if (*workspace_bytes < ws.total)
    return mgs::set_error(MGS_ERR_WORKSPACE_TOO_SMALL, "render_frames: workspace %zu < %zu bytes", *workspace_bytes, ws.total);
  MGS_REQUIRE(viewmats && Ks && render && alphas && n_isect && status, "render_frames: null pointer");
  MGS_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "render_frames: workspace must be 256-byte aligned");
  char* w = static_cast<char*>(workspace);
  float* depths = reinterpret_cast<float*>(w + ws.depths);
  float* opac_aa = antialiased ? reinterpret_cast<float*>(w + ws.opac) : nullptr;
  float* splats = reinterpret_cast<float*>(w + ws.splats);
  uint32_t* bin_info = reinterpret_cast<uint32_t*>(w + ws.bin_info);
  uint32_t* bin_sums = reinterpret_cast<uint32_t*>(w + ws.bin_sums);
  int32_t* flatten = reinterpret_cast<int32_t*>(w + ws.flatten);
  int32_t* offsets = reinterpret_cast<int32_t*>(w + ws.offsets);
  int32_t* order = reinterpret_cast<int32_t*>(w + ws.order);
  const size_t n_px = (size_t)width * height;
  for (int c = 0; c < n_cams; ++c) {
    rc = mgs_project_color_fwd(n, means, quats, scales, opacities, sh_degree, coeff_stride, sh_coeffs,
                               viewmats + 16 * (size_t)c, Ks + 9 * (size_t)c, width, height, eps2d, near_plane, far_plane,
                               radius_clip, nullptr, nullptr, depths, nullptr, opac_aa, channels, nullptr, splats,
                               (flags & MGS_FRAMES_CLASSIC_BOUNDS) ? 0 : 1 /* tight tile bounds: same pixels, shorter lists */,
                               bin_info, bin_sums, stream);
    if (rc) return rc;
    size_t iw = ws.isect_bytes;
    rc = mgs_isect_tiles(n, nullptr, nullptr, depths, nullptr, nullptr, MGS_TILE_SIZE, tile_w, tile_h, 0, 1,
                         isect_capacity, nullptr, n_isect + c, nullptr, flatten, nullptr, offsets, nullptr, order,
                         status + c, bin_info, bin_sums, w + ws.isect, &iw, stream);
    if (rc) return rc;
    rc = mgs_rasterize_fwd(n, nullptr, nullptr, nullptr, nullptr, splats, backgrounds ? backgrounds + (size_t)channels * c : nullptr,
                           channels, width, height, tile_w, tile_h, offsets, flatten, order,
                           flags & (MGS_RASTER_EXPECTED_LAST | MGS_RASTER_LATENCY),
                           render + n_px * channels * c, alphas + n_px * c, nullptr, nullptr, 0, stream);
    if (rc) return rc;
  }
  return MGS_OK;
}
