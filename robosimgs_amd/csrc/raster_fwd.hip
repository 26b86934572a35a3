// raster_fwd.hip -- per-tile depth-ordered alpha compositing, forward (A.2 step 9), gfx950.
// Geometry, queue and culling: raster_common.h.  VALU / v_exp bound (about 20 vector ops per
// pixel-Gaussian pair against 44 bytes per tile-Gaussian pair), so the design spends its
// effort on evaluating fewer pairs, not on moving bytes.
#include "raster_common.h"

namespace mgs {
namespace {

template <int CHT>
struct QueueEntry {
  float4 geo0;                       // mean.x, mean.y, conic.a, conic.b
  float4 geo1;                       // conic.c, opacity, quadrant mask (bits), list index (bits)
  float4 feat[(CHT + 3) / 4];
};

template <int CHT>
struct PixelState {
  float T;
  float C[CHT];
  int last;
};

// One Gaussian against the 64 pixels of one quadrant (one pixel per lane).
template <int CHT>
__device__ __forceinline__ void blend_pixel(PixelState<CHT>& px, bool& done, float pxf, float pyf,
                                            float mx, float my, float ca, float cb, float cc,
                                            float opac, const float* feat, int idx) {
  float dx = mx - pxf, dy = my - pyf;
  float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
  float alpha = fminf(kAlphaMax, opac * __expf(-sigma));
  bool valid = !done && sigma >= 0.f && alpha >= kAlphaMin;
  float next_T = px.T * (1.0f - alpha);
  bool stop = valid && next_T <= kTStop;
  done = done || stop;
  bool acc = valid && !stop;
  float w = acc ? alpha * px.T : 0.f;
#pragma unroll
  for (int c = 0; c < CHT; ++c) px.C[c] += w * feat[c];
  px.T = acc ? next_T : px.T;
  px.last = acc ? idx : px.last;
}

template <int CHT>
__global__ __launch_bounds__(64) void raster_fwd_kernel(
    const float* __restrict__ means2d, const float* __restrict__ conics,
    const float* __restrict__ feats, const float* __restrict__ opacities,
    const float* __restrict__ background, int channels, int width, int height, int tile_w,
    int n_tiles, const int32_t* __restrict__ tile_offsets,
    const int32_t* __restrict__ flatten_ids, float* __restrict__ render,
    float* __restrict__ alphas, int32_t* __restrict__ last_ids) {
  __shared__ QueueEntry<CHT> queue[kQueue + 1];
  const int tile = blockIdx.x;
  if (tile >= n_tiles) return;
  const unsigned lane = threadIdx.x;
  const int tx = tile % tile_w, ty = tile / tile_w;
  const float tile_x = (float)(tx * 16), tile_y = (float)(ty * 16);
  const int start = tile_offsets[tile], end = tile_offsets[tile + 1];

  // pixel centres of quadrant 0; quadrant k adds (8*(k&1), 8*(k>>1))
  const int ix = tx * 16 + (int)(lane & 7), iy = ty * 16 + (int)(lane >> 3);
  const float px0 = (float)ix + 0.5f, py0 = (float)iy + 0.5f;

  PixelState<CHT> st[4];
  bool done[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    st[k].T = 1.f;
    st[k].last = 0;
#pragma unroll
    for (int c = 0; c < CHT; ++c) st[k].C[c] = 0.f;
    done[k] = !(ix + 8 * (k & 1) < width && iy + 8 * (k >> 1) < height);
  }

  // raw batch registers (software prefetch of the next 64 list entries)
  int r_idx = start + (int)lane;
  bool r_ok = r_idx < end;
  float2 r_xy = make_float2(0.f, 0.f);
  float r_ca = 1.f, r_cb = 0.f, r_cc = 1.f, r_op = 0.f;
  float r_feat[CHT];
#pragma unroll
  for (int c = 0; c < CHT; ++c) r_feat[c] = 0.f;
  auto fetch = [&](int idx, bool ok) {
    if (ok) {
      int g = flatten_ids[idx];
      r_xy = reinterpret_cast<const float2*>(means2d)[g];
      r_ca = conics[3 * (size_t)g + 0];
      r_cb = conics[3 * (size_t)g + 1];
      r_cc = conics[3 * (size_t)g + 2];
      r_op = opacities[g];
#pragma unroll
      for (int c = 0; c < CHT; ++c) r_feat[c] = c < channels ? feats[(size_t)g * channels + c] : 0.f;
    }
  };
  fetch(r_idx, r_ok);

  for (int b = start; b < end; b += kQueue) {
    // which quadrants still have an unfinished pixel
    unsigned live = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (__ballot(!done[k]) != 0ull) live |= 1u << k;
    if (live == 0) break;

    // take the prefetched batch, start the next one
    const int c_idx = r_idx;
    const bool c_ok = r_ok;
    const float2 c_xy = r_xy;
    const float c_ca = r_ca, c_cb = r_cb, c_cc = r_cc, c_op = r_op;
    float c_feat[CHT];
#pragma unroll
    for (int c = 0; c < CHT; ++c) c_feat[c] = r_feat[c];
    r_idx = b + kQueue + (int)lane;
    r_ok = r_idx < end;
    fetch(r_idx, r_ok);

    unsigned qmask = 0;
    if (c_ok) qmask = quadrant_mask(c_xy.x, c_xy.y, c_ca, c_cb, c_cc, c_op, tile_x, tile_y) & live;
    const unsigned long long keep = __ballot(qmask != 0u);
    const int count = __popcll(keep);
    if (qmask != 0u) {
      QueueEntry<CHT>& e = queue[mask_rank(keep)];
      e.geo0 = make_float4(c_xy.x, c_xy.y, c_ca, c_cb);
      e.geo1 = make_float4(c_cc, c_op, __uint_as_float(qmask), __int_as_float(c_idx));
#pragma unroll
      for (int f = 0; f < (CHT + 3) / 4; ++f) {
        float4 v;
        v.x = c_feat[4 * f];
        v.y = 4 * f + 1 < CHT ? c_feat[4 * f + 1] : 0.f;
        v.z = 4 * f + 2 < CHT ? c_feat[4 * f + 2] : 0.f;
        v.w = 4 * f + 3 < CHT ? c_feat[4 * f + 3] : 0.f;
        e.feat[f] = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    for (int j = 0; j < count; ++j) {
      const QueueEntry<CHT>& e = queue[j];
      const float4 g0 = e.geo0, g1 = e.geo1;
      float feat[CHT];
#pragma unroll
      for (int f = 0; f < (CHT + 3) / 4; ++f) {
        float4 v = e.feat[f];
        feat[4 * f] = v.x;
        if (4 * f + 1 < CHT) feat[4 * f + 1] = v.y;
        if (4 * f + 2 < CHT) feat[4 * f + 2] = v.z;
        if (4 * f + 3 < CHT) feat[4 * f + 3] = v.w;
      }
      const unsigned m = __builtin_amdgcn_readfirstlane(__float_as_uint(g1.z));
      const int idx = __float_as_int(g1.w);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (m & (1u << k))
          blend_pixel<CHT>(st[k], done[k], px0 + 8.f * (k & 1), py0 + 8.f * (k >> 1), g0.x, g0.y,
                           g0.z, g0.w, g1.x, g1.y, feat, idx);
      }
    }
    __builtin_amdgcn_wave_barrier();   // queue is rewritten by the next batch
  }

#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int x = ix + 8 * (k & 1), y = iy + 8 * (k >> 1);
    if (x < width && y < height) {
      const size_t p = (size_t)y * width + x;
#pragma unroll
      for (int c = 0; c < CHT; ++c)
        if (c < channels)
          render[p * channels + c] = st[k].C[c] + (background ? st[k].T * background[c] : 0.f);
      alphas[p] = 1.0f - st[k].T;
      last_ids[p] = st[k].last;
    }
  }
}

}  // namespace
}  // namespace mgs

using namespace mgs;

extern "C" int mgs_rasterize_fwd(int n, const float* means2d, const float* conics,
                                 const float* feats, const float* opacities,
                                 const float* background, int channels, int width, int height,
                                 int tile_w, int tile_h, const int32_t* tile_offsets,
                                 const int32_t* flatten_ids, float* render, float* alphas,
                                 int32_t* last_ids, mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && width > 0 && height > 0, "rasterize_fwd: bad sizes");
  MGS_REQUIRE(channels >= 1 && channels <= MGS_MAX_CHANNELS, "rasterize_fwd: channels %d outside 1..%d", channels, MGS_MAX_CHANNELS);
  MGS_REQUIRE(tile_w == (width + 15) / 16 && tile_h == (height + 15) / 16,
              "rasterize_fwd: tile grid %dx%d does not match %dx%d at tile size 16", tile_w, tile_h, width, height);
  MGS_REQUIRE(means2d && conics && feats && opacities && tile_offsets && flatten_ids && render &&
                  alphas && last_ids, "rasterize_fwd: null pointer");
  const int n_tiles = tile_w * tile_h;
  hipStream_t s = (hipStream_t)stream;
#define MGS_RF_LAUNCH(C)                                                                       \
  hipLaunchKernelGGL((raster_fwd_kernel<C>), dim3(n_tiles), dim3(64), 0, s, means2d, conics,   \
                     feats, opacities, background, channels, width, height, tile_w, n_tiles,   \
                     tile_offsets, flatten_ids, render, alphas, last_ids)
  if (channels == 1) MGS_RF_LAUNCH(1);
  else if (channels == 2) MGS_RF_LAUNCH(2);
  else if (channels == 3) MGS_RF_LAUNCH(3);
  else if (channels == 4) MGS_RF_LAUNCH(4);
  else if (channels <= 8) MGS_RF_LAUNCH(8);
  else if (channels <= 16) MGS_RF_LAUNCH(16);
  else MGS_RF_LAUNCH(32);
#undef MGS_RF_LAUNCH
  return check_launch("rasterize_fwd");
}
