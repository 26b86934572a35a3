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
                                 uint32_t* status, uint8_t* ds_rgba, void* ds_distance, int ds_distance_type,
                                 const double* ds_Kinv_host, void* workspace, size_t* workspace_bytes, mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && n_cams >= 1 && width > 0 && height > 0, "render_frames: bad sizes");
  MGS_REQUIRE(channels == 3 || channels == 4, "render_frames: channels must be 3 (RGB) or 4 (RGB + depth), got %d", channels);
  MGS_REQUIRE(workspace_bytes, "render_frames: workspace_bytes is null");
  MGS_REQUIRE(isect_capacity > 0, "render_frames: zero capacity");
  const int tile_w = (width + MGS_TILE_SIZE - 1) / MGS_TILE_SIZE, tile_h = (height + MGS_TILE_SIZE - 1) / MGS_TILE_SIZE;
  const int n_tiles = tile_w * tile_h;
  size_t isect_ws = 0;
  int rc = mgs_isect_tiles(n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, MGS_TILE_SIZE, tile_w, tile_h, 0, 1,
                           isect_capacity, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, nullptr, nullptr, nullptr, &isect_ws, stream);
  if (rc) return rc;
  const FrameWs ws(n, isect_capacity, n_tiles, antialiased != 0, isect_ws);
  if (!workspace) {
    *workspace_bytes = ws.total;
    return MGS_OK;
  }
  if (*workspace_bytes < ws.total)
    return mgs::set_error(MGS_ERR_WORKSPACE_TOO_SMALL, "render_frames: workspace %zu < %zu bytes", *workspace_bytes, ws.total);
  MGS_REQUIRE(viewmats && Ks && ((render && alphas) || (ds_rgba && !render && !alphas)) && n_isect && status,
              "render_frames: null pointer");
  MGS_REQUIRE(ds_distance_type >= 0 && ds_distance_type <= 2, "render_frames: distance type %d not in {0: f32, 1: f64, 2: f16}",
              ds_distance_type);
  const size_t ds_dist_bytes = ds_distance_type == 1 ? 8 : (ds_distance_type == 2 ? 2 : 4);
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
                               ((flags & MGS_FRAMES_CLASSIC_BOUNDS) ? 0 : MGS_BIN_TIGHT /* same pixels, shorter lists */) |
                                   ((flags & MGS_FRAMES_RADIUS_OPACITY_AWARE) ? MGS_BIN_RADIUS_OPACITY_AWARE : 0),
                               bin_info, bin_sums, nullptr, stream);
    if (rc) return rc;
    size_t iw = ws.isect_bytes;
    rc = mgs_isect_tiles(n, nullptr, nullptr, nullptr, depths, nullptr, nullptr, MGS_TILE_SIZE, tile_w, tile_h, 0, 1,
                         isect_capacity, nullptr, n_isect + c, nullptr, flatten, nullptr, offsets, nullptr, order,
                         status + c, bin_info, bin_sums, nullptr, w + ws.isect, &iw, stream);
    if (rc) return rc;
    rc = mgs_rasterize_fwd(n, nullptr, nullptr, nullptr, nullptr, splats, backgrounds ? backgrounds + (size_t)channels * c : nullptr,
                           channels, width, height, tile_w, tile_h, offsets, flatten, order,
                           flags & (MGS_RASTER_EXPECTED_LAST | MGS_RASTER_LATENCY),
                           render ? render + n_px * channels * c : nullptr, alphas ? alphas + n_px * c : nullptr, nullptr, nullptr, 0,
                           ds_rgba ? ds_rgba + n_px * 4 * c : nullptr,
                           ds_distance ? static_cast<char*>(ds_distance) + n_px * ds_dist_bytes * c : nullptr, ds_distance_type,
                           ds_Kinv_host, stream);
    if (rc) return rc;
  }
  return MGS_OK;
}

// ---- a batch of TRAINING frames behind two C calls ----------------------------------------------------------------
// mgs_render_frames_train: per camera the full-output projection (radii / means2d / depths / conics / feats / records /
// binning seed), the seeded binning with tiles_per_gauss, tile ids, pair_info and launch order, and the raster forward
// with last_ids and checkpoints -- everything the backward and gsplat's `meta` need, kept per camera in the caller's
// `state` (mgs_train_state_layout says where).  mgs_render_frames_backward: per camera the segmented raster backward
// and the projection / SH backward, the first camera overwriting the parameter gradients, the later ones adding.
// What rasterization() used to do with five ctypes calls and a dozen tensor allocations per camera.
namespace {
enum TrainField { TF_RADII, TF_MEANS2D, TF_DEPTHS, TF_CONICS, TF_OPAC, TF_FEATS, TF_SPLATS, TF_TILES_PER_GAUSS, TF_PAIR_INFO,
                  TF_TILE_IDS, TF_FLATTEN, TF_OFFSETS, TF_ORDER, TF_LAST_IDS, TF_CKPT, TF_COUNTS, TF_RADII_Y, TF_FIELDS };
static_assert(TF_FIELDS == MGS_TRAIN_FIELDS, "mgs.h: MGS_TRAIN_FIELDS");
struct TrainState {
  size_t at[TF_FIELDS], total;
  TrainState(int n, int width, int height, int channels, uint32_t cap, bool antialiased, int interval) {
    const int tile_w = (width + MGS_TILE_SIZE - 1) / MGS_TILE_SIZE, tile_h = (height + MGS_TILE_SIZE - 1) / MGS_TILE_SIZE;
    const size_t nn = (size_t)(n > 0 ? n : 1), n_tiles = (size_t)tile_w * tile_h, n_px = (size_t)width * height;
    size_t o = 0;
    auto take = [&](int f, size_t bytes) { at[f] = o; o += mgs::align_up(bytes ? bytes : 1, 256); };
    take(TF_RADII, nn * 4);
    take(TF_MEANS2D, nn * 8);
    take(TF_DEPTHS, nn * 4);
    take(TF_CONICS, nn * 12);
    take(TF_OPAC, antialiased ? nn * 4 : 0);
    take(TF_FEATS, nn * 4 * channels);
    take(TF_SPLATS, nn * 48);
    take(TF_TILES_PER_GAUSS, nn * 4);
    take(TF_PAIR_INFO, nn * 16);
    take(TF_TILE_IDS, (size_t)cap * 4);
    take(TF_FLATTEN, (size_t)cap * 4);
    take(TF_OFFSETS, (n_tiles + 1) * 4);
    take(TF_ORDER, ((n_tiles + 3) / 4) * 4);
    take(TF_LAST_IDS, n_px * 4);
    take(TF_CKPT, interval ? mgs_raster_checkpoint_floats(cap, tile_w, tile_h, channels, interval) * 4 : 0);
    take(TF_COUNTS, 8);                      // n_isect, status
    take(TF_RADII_Y, nn * 4);                // written under MGS_FRAMES_RADIUS_OPACITY_AWARE only
    total = o;
  }
};
struct TrainWs {     // shared by the cameras of a call
  size_t bin_info, bin_sums, isect, total;
  TrainWs(int n, size_t isect_ws) {
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t a = o; o += mgs::align_up(bytes ? bytes : 1, 256); return a; };
    const size_t nn = (size_t)(n > 0 ? n : 1);
    bin_info = take(nn * 8);
    bin_sums = take(((nn + 63) / 64) * 4);
    isect = take(isect_ws);
    total = o;
  }
};
struct BwdWs {
  size_t v_means2d, v_abs, v_conics, v_feats, v_opac, raster, total;
  BwdWs(int n, int channels, size_t raster_ws) {
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t a = o; o += mgs::align_up(bytes ? bytes : 1, 256); return a; };
    const size_t nn = (size_t)(n > 0 ? n : 1);
    v_means2d = take(nn * 8);
    v_abs = take(nn * 8);
    v_conics = take(nn * 12);
    v_feats = take(nn * 4 * channels);
    v_opac = take(nn * 4);
    raster = take(raster_ws);
    total = o;
  }
};
__global__ __launch_bounds__(256) void add_rows_kernel(size_t n, const float* __restrict__ src, float* __restrict__ dst, int first) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = first ? src[i] : dst[i] + src[i];
}
}  // namespace

extern "C" int mgs_train_state_layout(int n, int width, int height, int channels, uint32_t isect_capacity, int antialiased,
                                      int checkpoint_interval, size_t* offsets, size_t* bytes_per_camera) {
  MGS_REQUIRE(n >= 0 && width > 0 && height > 0 && (channels == 3 || channels == 4) && isect_capacity > 0 && bytes_per_camera,
              "train_state_layout: bad arguments");
  MGS_REQUIRE(checkpoint_interval == 0 || (checkpoint_interval >= 64 && (checkpoint_interval & (checkpoint_interval - 1)) == 0),
              "train_state_layout: checkpoint_interval %d is not 0 or a power of two >= 64", checkpoint_interval);
  const TrainState st(n, width, height, channels, isect_capacity, antialiased != 0, checkpoint_interval);
  if (offsets)
    for (int f = 0; f < TF_FIELDS; ++f) offsets[f] = st.at[f];
  *bytes_per_camera = st.total;
  return MGS_OK;
}

extern "C" int mgs_render_frames_train(int n, const float* means, const float* quats, const float* scales,
                                       const float* opacities, int sh_degree, int coeff_stride, const float* sh_coeffs,
                                       int n_cams, const float* viewmats, const float* Ks, int width, int height,
                                       float eps2d, float near_plane, float far_plane, float radius_clip, int antialiased,
                                       int channels, int flags, const float* backgrounds, uint32_t isect_capacity,
                                       int checkpoint_interval, float* render, float* alphas, void* state, void* workspace,
                                       size_t* workspace_bytes, mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && n_cams >= 1 && width > 0 && height > 0, "render_frames_train: bad sizes");
  MGS_REQUIRE(channels == 3 || channels == 4, "render_frames_train: channels must be 3 (RGB) or 4 (RGB + depth), got %d", channels);
  MGS_REQUIRE(workspace_bytes, "render_frames_train: workspace_bytes is null");
  MGS_REQUIRE(isect_capacity > 0, "render_frames_train: zero capacity");
  const int tile_w = (width + MGS_TILE_SIZE - 1) / MGS_TILE_SIZE, tile_h = (height + MGS_TILE_SIZE - 1) / MGS_TILE_SIZE;
  size_t isect_ws = 0;
  int rc = mgs_isect_tiles(n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, MGS_TILE_SIZE, tile_w, tile_h, 0, 1,
                           isect_capacity, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, nullptr, nullptr, nullptr, &isect_ws, stream);
  if (rc) return rc;
  const TrainWs ws(n, isect_ws);
  if (!workspace) {
    *workspace_bytes = ws.total;
    return MGS_OK;
  }
  if (*workspace_bytes < ws.total)
    return mgs::set_error(MGS_ERR_WORKSPACE_TOO_SMALL, "render_frames_train: workspace %zu < %zu bytes", *workspace_bytes, ws.total);
  MGS_REQUIRE(viewmats && Ks && render && alphas && state, "render_frames_train: null pointer");
  MGS_REQUIRE(((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(state)) & 255u) == 0,
              "render_frames_train: workspace and state must be 256-byte aligned");
  MGS_REQUIRE(checkpoint_interval == 0 || (checkpoint_interval >= 64 && (checkpoint_interval & (checkpoint_interval - 1)) == 0),
              "render_frames_train: checkpoint_interval %d is not 0 or a power of two >= 64", checkpoint_interval);
  const TrainState st(n, width, height, channels, isect_capacity, antialiased != 0, checkpoint_interval);
  char* w = static_cast<char*>(workspace);
  uint32_t* bin_info = reinterpret_cast<uint32_t*>(w + ws.bin_info);
  uint32_t* bin_sums = reinterpret_cast<uint32_t*>(w + ws.bin_sums);
  const size_t n_px = (size_t)width * height;
  const int tight = ((flags & MGS_FRAMES_CLASSIC_BOUNDS) ? 0 : MGS_BIN_TIGHT) |
                    ((flags & MGS_FRAMES_RADIUS_OPACITY_AWARE) ? MGS_BIN_RADIUS_OPACITY_AWARE : 0);
  for (int c = 0; c < n_cams; ++c) {
    char* s = static_cast<char*>(state) + st.total * (size_t)c;
    auto F = [&](int f) { return reinterpret_cast<float*>(s + st.at[f]); };
    auto I = [&](int f) { return reinterpret_cast<int32_t*>(s + st.at[f]); };
    auto U = [&](int f) { return reinterpret_cast<uint32_t*>(s + st.at[f]); };
    float* opac_aa = antialiased ? F(TF_OPAC) : nullptr;
    rc = mgs_project_color_fwd(n, means, quats, scales, opacities, sh_degree, coeff_stride, sh_coeffs,
                               viewmats + 16 * (size_t)c, Ks + 9 * (size_t)c, width, height, eps2d, near_plane, far_plane,
                               radius_clip, I(TF_RADII), F(TF_MEANS2D), F(TF_DEPTHS), F(TF_CONICS), opac_aa, channels,
                               F(TF_FEATS), F(TF_SPLATS), tight, bin_info, bin_sums, I(TF_RADII_Y), stream);
    if (rc) return rc;
    size_t iw = isect_ws;
    rc = mgs_isect_tiles(n, nullptr, nullptr, nullptr, F(TF_DEPTHS), nullptr, nullptr, MGS_TILE_SIZE, tile_w, tile_h, c, n_cams,
                         isect_capacity, I(TF_TILES_PER_GAUSS), U(TF_COUNTS), U(TF_TILE_IDS), I(TF_FLATTEN), nullptr,
                         I(TF_OFFSETS), I(TF_PAIR_INFO), I(TF_ORDER), U(TF_COUNTS) + 1, bin_info, bin_sums, F(TF_SPLATS),
                         w + ws.isect, &iw, stream);
    if (rc) return rc;
    rc = mgs_rasterize_fwd(n, nullptr, nullptr, nullptr, nullptr, F(TF_SPLATS),
                           backgrounds ? backgrounds + (size_t)channels * c : nullptr, channels, width, height, tile_w, tile_h,
                           I(TF_OFFSETS), I(TF_FLATTEN), I(TF_ORDER), flags & (MGS_RASTER_EXPECTED_LAST | MGS_RASTER_LATENCY),
                           render + n_px * channels * c, alphas + n_px * c, I(TF_LAST_IDS),
                           checkpoint_interval ? F(TF_CKPT) : nullptr, checkpoint_interval, nullptr, nullptr, 0, nullptr, stream);
    if (rc) return rc;
  }
  return MGS_OK;
}

extern "C" int mgs_render_frames_backward(int n, const float* means, const float* quats, const float* scales,
                                          const float* opacities, int sh_degree, int coeff_stride, const float* sh_coeffs,
                                          int n_cams, const float* viewmats, const float* Ks, int width, int height,
                                          float eps2d, int antialiased, int channels, int flags, const float* backgrounds,
                                          uint32_t isect_capacity, int checkpoint_interval, const float* render,
                                          const float* alphas, const float* v_render, const float* v_alphas,
                                          const void* state, float* v_means, float* v_quats, float* v_scales,
                                          float* v_sh_coeffs, float* v_opacities, float* v_viewmats, float* v_means2d,
                                          float* v_means2d_abs, void* workspace, size_t* workspace_bytes,
                                          mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && n_cams >= 1 && width > 0 && height > 0, "render_frames_backward: bad sizes");
  MGS_REQUIRE(channels == 3 || channels == 4, "render_frames_backward: channels must be 3 or 4, got %d", channels);
  MGS_REQUIRE(workspace_bytes, "render_frames_backward: workspace_bytes is null");
  MGS_REQUIRE(isect_capacity > 0, "render_frames_backward: zero capacity");
  const int tile_w = (width + MGS_TILE_SIZE - 1) / MGS_TILE_SIZE, tile_h = (height + MGS_TILE_SIZE - 1) / MGS_TILE_SIZE;
  float* const dummy = reinterpret_cast<float*>(16);      // (size query: the absgrad record layout follows this pointer)
  size_t raster_ws = 0;
  int rc = mgs_rasterize_bwd_det(0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, channels, width, height, tile_w,
                                 tile_h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                 isect_capacity, nullptr, nullptr, 0, 0, nullptr, v_means2d_abs ? dummy : nullptr, nullptr,
                                 nullptr, nullptr, nullptr, &raster_ws, stream);
  if (rc) return rc;
  if (checkpoint_interval) {       // the unit tables of the segmented launch live in the same workspace
    size_t with_ckpt = 0;
    rc = mgs_rasterize_bwd_det(0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, channels, width, height, tile_w,
                               tile_h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                               isect_capacity, dummy, dummy, checkpoint_interval, 0, nullptr, v_means2d_abs ? dummy : nullptr,
                               nullptr, nullptr, nullptr, nullptr, &with_ckpt, stream);
    if (rc) return rc;
    raster_ws = with_ckpt > raster_ws ? with_ckpt : raster_ws;
  }
  const BwdWs ws(n, channels, raster_ws);
  if (!workspace) {
    *workspace_bytes = ws.total;
    return MGS_OK;
  }
  if (*workspace_bytes < ws.total)
    return mgs::set_error(MGS_ERR_WORKSPACE_TOO_SMALL, "render_frames_backward: workspace %zu < %zu bytes", *workspace_bytes, ws.total);
  MGS_REQUIRE(viewmats && Ks && render && alphas && v_render && state && v_means && v_quats && v_scales && v_sh_coeffs &&
                  v_opacities, "render_frames_backward: null pointer");
  MGS_REQUIRE(((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(state)) & 255u) == 0,
              "render_frames_backward: workspace and state must be 256-byte aligned");
  const TrainState st(n, width, height, channels, isect_capacity, antialiased != 0, checkpoint_interval);
  char* w = static_cast<char*>(workspace);
  const size_t n_px = (size_t)width * height;
  hipStream_t hs = (hipStream_t)stream;
  for (int c = 0; c < n_cams; ++c) {
    const char* s = static_cast<const char*>(state) + st.total * (size_t)c;
    auto F = [&](int f) { return reinterpret_cast<const float*>(s + st.at[f]); };
    auto I = [&](int f) { return reinterpret_cast<const int32_t*>(s + st.at[f]); };
    const float* opac = antialiased ? F(TF_OPAC) : opacities;
    float* g_m2d = v_means2d ? v_means2d + 2 * (size_t)n * c : reinterpret_cast<float*>(w + ws.v_means2d);
    float* g_abs = v_means2d_abs ? v_means2d_abs + 2 * (size_t)n * c : nullptr;
    float* g_con = reinterpret_cast<float*>(w + ws.v_conics);
    float* g_feat = reinterpret_cast<float*>(w + ws.v_feats);
    // (not anti-aliased: the blend's opacity gradient IS the parameter's -- the first camera's reduce writes it in place)
    float* g_opac = (!antialiased && c == 0) ? v_opacities : reinterpret_cast<float*>(w + ws.v_opac);
    const float* frame = render + n_px * channels * c;
    const bool ed = (flags & MGS_RASTER_EXPECTED_LAST) != 0;
    size_t rw = raster_ws;
    rc = mgs_rasterize_bwd_det(n, F(TF_MEANS2D), F(TF_CONICS), F(TF_FEATS), opac, F(TF_SPLATS),
                               backgrounds ? backgrounds + (size_t)channels * c : nullptr, channels, width, height, tile_w,
                               tile_h, I(TF_OFFSETS), I(TF_FLATTEN), alphas + n_px * c, I(TF_LAST_IDS),
                               v_render + n_px * channels * c, v_alphas ? v_alphas + n_px * c : nullptr, ed ? frame : nullptr,
                               I(TF_PAIR_INFO), I(TF_ORDER), isect_capacity, checkpoint_interval ? frame : nullptr,
                               checkpoint_interval ? F(TF_CKPT) : nullptr, checkpoint_interval, MGS_RASTER_BWD_SPLAT_SLOTS, g_m2d, g_abs, g_con, g_feat,
                               g_opac, w + ws.raster, &rw, stream);
    if (rc) return rc;
    rc = mgs_project_color_bwd(n, means, quats, scales, opacities, sh_degree, coeff_stride, sh_coeffs,
                               viewmats + 16 * (size_t)c, Ks + 9 * (size_t)c, width, height, eps2d, I(TF_RADII), F(TF_CONICS),
                               antialiased, channels, F(TF_FEATS), g_feat, g_m2d, g_con, nullptr,
                               antialiased ? g_opac : nullptr, v_means, v_quats, v_scales, v_sh_coeffs,
                               antialiased ? v_opacities : nullptr, v_viewmats ? v_viewmats + 16 * (size_t)c : nullptr,
                               c > 0 ? 1 : 0, stream);
    if (rc) return rc;
    if (!antialiased && n > 0 && c > 0)     // later cameras add theirs
      hipLaunchKernelGGL(add_rows_kernel, dim3(mgs::div_up((unsigned)n, 256u)), dim3(256), 0, hs, (size_t)n, g_opac, v_opacities, 0);
  }
  return mgs::check_launch("render_frames_backward");
}
