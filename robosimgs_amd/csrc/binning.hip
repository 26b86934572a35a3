// binning.hip -- tile binning for gfx950: depth-ordered per-tile Gaussian lists.
//
// The textbook pipeline (gsplat `isect_tiles`) emits one 64-bit key (tile << 32 | depth bits)
// per (Gaussian, tile) pair and radix-sorts all n_isect pairs on ~45 bits: six 8-bit passes
// over 12-byte elements.  The same ordering is produced here with ~4.5x less sort traffic and
// a third of the dependent launches of this repo's first version:
//   1. per Gaussian: tile rectangle and tile count; exclusive scan of the counts in INDEX order
//      (total = n_isect, kept on the device; the same scan gives the backward's record slots);
//   2. emit (tile, gaussian) pairs in index order with a load-balanced search so that stores are
//      lane-linear;
//   3. stable radix sort on the tile bits only (13 bits at 1080p: a 7-bit and a 6-bit pass over
//      8-byte pairs): every tile's list is now contiguous, in Gaussian-index order;
//   4. first index of every tile;
//   5. tile_sort.hip: one workgroup per tile orders its list by (depth bits, index).
// Sorting each tile's index-ordered list by (depth, index) == the stable sort on (tile, depth)
// with index-order ties: flatten_ids / isect_ids are bit-identical to
// the single-key formulation.  (Round 1 sorted the N Gaussians by depth first -- twelve dependent
// launches for 1 M keys -- and gathered their rectangles into rank order.)
#include "mgs_common.h"
#include "tile_rect.h"

namespace mgs {
namespace {

// 128-thread workgroups interleave better with other frames' raster waves than 256 (3035 -> 3090
// frames/s at three frames in flight); 64 makes the kernels themselves slower (binning 224 -> 235 us).
constexpr int kBlock = 128;

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// per Gaussian: tile rectangle (packed) + tile count, and the count sum of every 64 consecutive
// Gaussians (the standalone operator's version of what mgs_project_color_fwd seeds the binning with)
constexpr int kSum = 64;
__global__ __launch_bounds__(kBlock) void tile_count_kernel(
    int n, const float* __restrict__ means2d, const int32_t* __restrict__ radii,
    const float* __restrict__ conics, const float* __restrict__ opacities, float tile_size,
    int tile_w, int tile_h, uint2* __restrict__ ginfo, uint32_t* __restrict__ sums) {
  int g = blockIdx.x * kBlock + threadIdx.x;
  int radius = g < n ? radii[g] : 0;
  uint2 info = make_uint2(kEmptyTileRect, 0u);
  if (radius > 0) {
    float2 m = reinterpret_cast<const float2*>(means2d)[g];
    TileRect r = tile_rect(m.x, m.y, radius, tile_size, tile_w, tile_h);
    if (conics)
      r = tighten_rect(r, m.x, m.y, conics[3 * (size_t)g], conics[3 * (size_t)g + 1],
                       conics[3 * (size_t)g + 2], opacities[g], tile_size);
    info = pack_tile_rect(r);
  }
  if (g < n) ginfo[g] = info;
  const uint32_t c = wave_sum(info.y);
  const int first = blockIdx.x * kBlock + (int)(threadIdx.x & ~63u);
  if ((threadIdx.x & 63) == 0 && first < n) sums[first / kSum] = c;
}

// single workgroup: exclusive scan of the per-64 sums in place; publishes n_isect / overflow.
// 16 consecutive sums per thread as four 16-byte loads in flight together (the kernel is one
// latency chain: load -> wave scan -> LDS -> store), 16384 sums per trip = 1 M Gaussians.
constexpr int kScanThreads = 1024;
constexpr int kScanPerThread = 16;
__global__ __launch_bounds__(kScanThreads) void scan_blocksums_kernel(
    uint32_t nblk, uint32_t* __restrict__ blocksums, uint32_t capacity,
    uint32_t* __restrict__ n_isect, uint32_t* __restrict__ status) {
  __shared__ uint32_t ws[kScanThreads / 64];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t running = 0;
  bool wrapped = false;            // the uint32 total overflowed: report it, the lists are not usable
  for (uint32_t b0 = 0; b0 < nblk; b0 += kScanThreads * kScanPerThread) {
    const uint32_t b = b0 + threadIdx.x * kScanPerThread;
    uint32_t v[kScanPerThread], sum = 0;
    if (b + kScanPerThread <= nblk) {          // the buffer is 256-byte aligned and b is a multiple of 16
      const uint4* src = reinterpret_cast<const uint4*>(blocksums + b);
      const uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
      v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
      v[8] = q2.x; v[9] = q2.y; v[10] = q2.z; v[11] = q2.w; v[12] = q3.x; v[13] = q3.y; v[14] = q3.z; v[15] = q3.w;
    } else {
#pragma unroll
      for (int j = 0; j < kScanPerThread; ++j) v[j] = b + j < nblk ? blocksums[b + j] : 0u;
    }
#pragma unroll
    for (int j = 0; j < kScanPerThread; ++j) sum += v[j];
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t t = __shfl_up(incl, d);
      if (lane >= (unsigned)d) incl += t;
    }
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    uint32_t off = 0, tot = 0;
    for (int w = 0; w < kScanThreads / 64; ++w) {
      if ((unsigned)w < wave) off += ws[w];
      tot += ws[w];
    }
    __syncthreads();
    uint32_t ex = running + off + incl - sum;
    uint32_t o[kScanPerThread];
#pragma unroll
    for (int j = 0; j < kScanPerThread; ++j) { o[j] = ex; ex += v[j]; }
    if (b + kScanPerThread <= nblk) {
      uint4* dst = reinterpret_cast<uint4*>(blocksums + b);
      dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
      dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
      dst[2] = make_uint4(o[8], o[9], o[10], o[11]);
      dst[3] = make_uint4(o[12], o[13], o[14], o[15]);
    } else {
#pragma unroll
      for (int j = 0; j < kScanPerThread; ++j)
        if (b + j < nblk) blocksums[b + j] = o[j];
    }
    wrapped |= running + tot < running;
    running += tot;
  }
  if (threadIdx.x == 0 && n_isect) {
    *n_isect = wrapped ? 0xffffffffu : running;
    *status = (wrapped || running > capacity) ? MGS_STATUS_ISECT_OVERFLOW : 0u;   // this call's result: no zero-fill needed
  }
}

// Slots for the deterministic backward, Gaussian-index-major: Gaussian g owns the slots
// [base, base + w*h) with base = exclusive scan of the tile counts in INDEX order, so that the
// per-Gaussian reduction reads contiguous memory from consecutive lanes.
__global__ __launch_bounds__(kBlock) void pair_info_kernel(
    int n, const uint2* __restrict__ ginfo, int tile_h, const uint32_t* __restrict__ blockbase,
    int4* __restrict__ pair_info) {
  __shared__ uint32_t ws[kBlock / 64];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int g = blockIdx.x * kBlock + threadIdx.x;
  uint2 info = g < n ? ginfo[g] : make_uint2(1u << 20, 0u);
  uint32_t cnt = info.y;
  uint32_t incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(incl, d);
    if (lane >= (unsigned)d) incl += t;
  }
  if (lane == 63) ws[wave] = incl;
  __syncthreads();
  uint32_t off = 0;
  for (int w = 0; w < kBlock / 64; ++w)
    if ((unsigned)w < wave) off += ws[w];
  if (g < n) {
    const uint32_t w = info.x >> 20;
    pair_info[g] = cnt ? make_int4((int)(blockbase[blockIdx.x * (kBlock / kSum)] + off + incl - cnt), (int)(info.x & 1023u),
                                   (int)((info.x >> 10) & 1023u), (int)(w | ((cnt / w) << 16)))
                       : make_int4(0, 0, 0, 0);
  }
}

// Load-balanced emit: a workgroup owns kBlock consecutive Gaussians; its output range is
// walked lane-linearly and each slot finds its Gaussian by binary search in LDS.
__global__ __launch_bounds__(kBlock) void emit_kernel(
    int n, const uint2* __restrict__ ginfo, int tile_w, const uint32_t* __restrict__ blockbase,
    uint32_t capacity, uint32_t* __restrict__ tile_out, uint32_t* __restrict__ id_out,
    int32_t* __restrict__ tiles_per_gauss) {
  __shared__ uint32_t prefix[kBlock + 1];
  __shared__ uint32_t gid[kBlock];
  __shared__ uint32_t rpack[kBlock];
  __shared__ uint32_t ws[kBlock / 64];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int r = blockIdx.x * kBlock + threadIdx.x;
  uint32_t cnt = 0, g = (uint32_t)r, pack = 1u << 20;
  if (r < n) {
    uint2 info = ginfo[r];
    pack = info.x;
    cnt = info.y;
    if (tiles_per_gauss) tiles_per_gauss[r] = (int32_t)cnt;
  }
  uint32_t incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(incl, d);
    if (lane >= (unsigned)d) incl += t;
  }
  if (lane == 63) ws[wave] = incl;
  __syncthreads();
  uint32_t off = 0, total = 0;
  for (int w = 0; w < kBlock / 64; ++w) {
    if ((unsigned)w < wave) off += ws[w];
    total += ws[w];
  }
  prefix[threadIdx.x] = off + incl - cnt;
  if (threadIdx.x == 0) prefix[kBlock] = total;
  gid[threadIdx.x] = g;
  rpack[threadIdx.x] = pack;
  __syncthreads();
  const uint32_t base = blockbase[blockIdx.x * (kBlock / kSum)];
  // each thread emits four consecutive slots: one binary search, then a linear walk
  for (uint32_t k0 = threadIdx.x * 4u; k0 < total; k0 += kBlock * 4u) {
    int lo = 0, hi = kBlock;   // invariant: prefix[lo] <= k0 < prefix[hi]
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (prefix[mid] <= k0) lo = mid; else hi = mid;
    }
    uint32_t next = prefix[lo + 1];
    uint32_t tiles4[4], ids4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t k = k0 + i;
      while (k >= next && lo < kBlock - 1) next = prefix[++lo + 1];   // skips empty Gaussians
      uint32_t local = k - prefix[lo];
      uint32_t p = rpack[lo];
      uint32_t w = p >> 20;
      uint32_t dy = local / w;
      uint32_t dx = local - dy * w;
      tiles4[i] = (((p >> 10) & 1023u) + dy) * (uint32_t)tile_w + (p & 1023u) + dx;
      ids4[i] = gid[lo];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t out = base + k0 + i;
      if (k0 + i < total && out < capacity) {
        tile_out[out] = tiles4[i];
        id_out[out] = ids4[i];
      }
    }
  }
}

// first sorted index of every tile; offsets[n_tiles] = n_isect.  Eight consecutive list entries per
// thread (two 16-byte loads in flight): an eighth of the waves, each as short-lived as before --
// with other frames' raster kernels running beside it, wave-slot time is what this kernel costs.
constexpr int kOffsetsPerThread = 8;
__global__ __launch_bounds__(kBlock) void tile_offsets_kernel(
    const uint32_t* __restrict__ n_ptr, uint32_t capacity, const uint32_t* __restrict__ tiles,
    int n_tiles, int32_t* __restrict__ offsets) {
  uint32_t n = min(*n_ptr, capacity);
  uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (n == 0) {
    for (uint32_t k = t; k <= (uint32_t)n_tiles; k += gridDim.x * kBlock) offsets[k] = 0;
    return;
  }
  uint32_t i0 = t * kOffsetsPerThread;
  if (i0 >= n) return;
  uint32_t v[kOffsetsPerThread];
  if (i0 + kOffsetsPerThread <= n) {
    const uint4 a = *reinterpret_cast<const uint4*>(tiles + i0), b = *reinterpret_cast<const uint4*>(tiles + i0 + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int j = 0; j < kOffsetsPerThread; ++j) v[j] = i0 + j < n ? tiles[i0 + j] : 0u;
  }
  int prev = i0 ? (int)tiles[i0 - 1] : -1;
#pragma unroll
  for (int j = 0; j < kOffsetsPerThread; ++j) {
    const uint32_t i = i0 + j;
    if (i < n) {
      const int cur = (int)v[j];
      for (int k = prev + 1; k <= cur; ++k) offsets[k] = (int32_t)i;
      prev = cur;
      if (i == n - 1)
        for (int k = cur + 1; k <= n_tiles; ++k) offsets[k] = (int32_t)n;
    }
  }
}

__global__ __launch_bounds__(kBlock) void isect_ids_kernel(
    const uint32_t* __restrict__ n_ptr, uint32_t capacity, const uint32_t* __restrict__ tiles,
    const int32_t* __restrict__ ids, const float* __restrict__ depths, int64_t cam_shifted,
    int64_t* __restrict__ isect_ids) {
  uint32_t n = min(*n_ptr, capacity);
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  int64_t hi = cam_shifted | (int64_t)tiles[i];
  isect_ids[i] = (hi << 32) | (int64_t)__float_as_uint(depths[ids[i]]);
}

__global__ __launch_bounds__(kBlock) void offset_encode_kernel(
    uint32_t n, const int64_t* __restrict__ isect_ids, int n_cams, int n_tiles, int tile_bits,
    int32_t* __restrict__ offsets) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  int total = n_cams * n_tiles;
  if (n == 0) {
    for (uint32_t k = i; k < (uint32_t)total; k += gridDim.x * kBlock) offsets[k] = 0;
    return;
  }
  if (i >= n) return;
  auto flat = [&](int64_t key) {
    int64_t t = key >> 32;
    return (int)((t >> tile_bits) * n_tiles + (t & ((1ll << tile_bits) - 1)));
  };
  int cur = flat(isect_ids[i]);
  int prev = i ? flat(isect_ids[i - 1]) : -1;
  for (int k = prev + 1; k <= cur && k < total; ++k) offsets[k] = (int32_t)i;
  if (i == n - 1)
    for (int k = cur + 1; k < total; ++k) offsets[k] = (int32_t)n;
}

int bits_for(uint32_t count) {   // bits needed to hold values 0..count-1
  int b = 0;
  while (count > 1 && (1ull << b) < count) ++b;
  return b;
}

struct Workspace {
  size_t total;
  size_t ginfo, blocksums, tile_alt, id_alt, radix, tsort;
  Workspace(int n, uint32_t cap) {
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align_up(bytes, 256); return at; };
    size_t nn = (size_t)(n > 0 ? n : 1), cc = cap ? cap : 1;
    ginfo = take(nn * 8);
    blocksums = take((size_t)div_up((unsigned)nn, kSum) * 4);
    tile_alt = take(cc * 4); id_alt = take(cc * 4);
    radix = take(radix_sort_temp_bytes((uint32_t)cc));
    tsort = take(tile_depth_sort_temp_bytes((uint32_t)cc));
    total = o;
  }
};

}  // namespace
}  // namespace mgs

using namespace mgs;

extern "C" int mgs_isect_tiles(int n, const float* means2d, const int32_t* radii,
                               const float* depths, const float* conics, const float* opacities,
                               int tile_size, int tile_w, int tile_h,
                               int cam_id, int n_cams, uint32_t isect_capacity,
                               int32_t* tiles_per_gauss, uint32_t* n_isect, uint32_t* tile_ids,
                               int32_t* flatten_ids, int64_t* isect_ids, int32_t* tile_offsets,
                               int32_t* pair_info, uint32_t* status, const uint32_t* seed_info,
                               uint32_t* seed_sums, void* workspace,
                               size_t* workspace_bytes, mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && tile_size > 0 && tile_w > 0 && tile_h > 0, "isect_tiles: bad sizes");
  MGS_REQUIRE(tile_w <= 1023 && tile_h <= 1023, "isect_tiles: tile grid %dx%d exceeds 1023x1023", tile_w, tile_h);
  MGS_REQUIRE(workspace_bytes, "isect_tiles: workspace_bytes is null");
  MGS_REQUIRE(cam_id >= 0 && n_cams > cam_id, "isect_tiles: cam_id %d outside 0..%d", cam_id, n_cams);
  Workspace ws(n, isect_capacity);
  if (!workspace) {
    *workspace_bytes = ws.total;
    return MGS_OK;
  }
  if (*workspace_bytes < ws.total)
    return set_error(MGS_ERR_WORKSPACE_TOO_SMALL, "isect_tiles: workspace %zu < %zu bytes",
                     *workspace_bytes, ws.total);
  MGS_REQUIRE(isect_capacity > 0, "isect_tiles: zero capacity");
  MGS_REQUIRE((conics == nullptr) == (opacities == nullptr),
              "isect_tiles: tight tile bounds need both conics and opacities");
  MGS_REQUIRE(n == 0 || (seed_info == nullptr) == (seed_sums == nullptr),
              "isect_tiles: seed_info and seed_sums come together (mgs_project_color_fwd writes both)");
  MGS_REQUIRE((n == 0 || ((seed_info || (means2d && radii)) && depths)) && n_isect && tile_ids && flatten_ids &&
                  tile_offsets && status, "isect_tiles: null pointer");
  hipStream_t s = (hipStream_t)stream;
  char* w = static_cast<char*>(workspace);
  auto u32 = [&](size_t off) { return reinterpret_cast<uint32_t*>(w + off); };
  const int n_tiles = tile_w * tile_h;
  const uint32_t cap = isect_capacity;
  int rc;

  if (n == 0) {
    (void)hipMemsetAsync(n_isect, 0, 4, s);
    (void)hipMemsetAsync(status, 0, 4, s);
  } else {
    const unsigned nblk = div_up(n, kBlock), nsum = div_up(n, kSum);
    // rectangles + counts + per-64 sums: seeded by the fused projection kernel, or computed here
    const uint2* ginfo = reinterpret_cast<const uint2*>(seed_info);
    uint32_t* sums = seed_sums;
    if (!seed_info) {
      uint2* gi = reinterpret_cast<uint2*>(w + ws.ginfo);
      sums = u32(ws.blocksums);
      hipLaunchKernelGGL(tile_count_kernel, dim3(nblk), dim3(kBlock), 0, s, n, means2d, radii, conics,
                         opacities, (float)tile_size, tile_w, tile_h, gi, sums);
      ginfo = gi;
    }
    hipLaunchKernelGGL(scan_blocksums_kernel, dim3(1), dim3(kScanThreads), 0, s, nsum, sums, cap,
                       n_isect, status);
    // tile sort: result must land in the caller's buffers
    const int tile_bits = bits_for((uint32_t)n_tiles);
    const int passes = (tile_bits + 7) / 8;
    uint32_t* user_t = tile_ids;
    uint32_t* user_i = reinterpret_cast<uint32_t*>(flatten_ids);
    uint32_t *a_t = user_t, *a_i = user_i, *b_t = u32(ws.tile_alt), *b_i = u32(ws.id_alt);
    if (passes & 1) { a_t = u32(ws.tile_alt); a_i = u32(ws.id_alt); b_t = user_t; b_i = user_i; }
    hipLaunchKernelGGL(emit_kernel, dim3(nblk), dim3(kBlock), 0, s, n, ginfo, tile_w, sums, cap, a_t,
                       a_i, tiles_per_gauss);
    if (pair_info)     // training only: the record slots of the backward are the same index-order scan
      hipLaunchKernelGGL(pair_info_kernel, dim3(nblk), dim3(kBlock), 0, s, n, ginfo, tile_h, sums,
                         reinterpret_cast<int4*>(pair_info));
    rc = radix_sort_pairs(n_isect, cap, tile_bits, a_t, a_i, b_t, b_i, w + ws.radix, s);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(tile_offsets_kernel, dim3(div_up(cap, kBlock * kOffsetsPerThread)), dim3(kBlock), 0,
                     s, n_isect, cap, tile_ids, n_tiles, tile_offsets);
  if (n > 0) {        // depth order inside every tile's list
    rc = tile_depth_sort(n_tiles, tile_offsets, depths, cap, reinterpret_cast<uint32_t*>(flatten_ids),
                         w + ws.tsort, s);
    if (rc) return rc;
  }
  const unsigned gblk = div_up(cap, kBlock);
  if (isect_ids) {
    const int tile_bits_key = bits_for((uint32_t)n_tiles + 1);   // floor(log2(n_tiles)) + 1
    hipLaunchKernelGGL(isect_ids_kernel, dim3(gblk), dim3(kBlock), 0, s, n_isect, cap, tile_ids,
                       flatten_ids, depths, (int64_t)cam_id << tile_bits_key, isect_ids);
  }
  return check_launch("isect_tiles");
}

extern "C" int mgs_isect_offset_encode(uint32_t n_isect, const int64_t* isect_ids, int n_cams,
                                       int tile_w, int tile_h, int32_t* offsets,
                                       mgs_stream_t stream) {
  MGS_REQUIRE(n_cams > 0 && tile_w > 0 && tile_h > 0, "isect_offset_encode: bad sizes");
  MGS_REQUIRE(offsets && (isect_ids || n_isect == 0), "isect_offset_encode: null pointer");
  const int n_tiles = tile_w * tile_h;
  const int tile_bits = bits_for((uint32_t)n_tiles + 1);
  unsigned gblk = n_isect ? div_up(n_isect, kBlock) : div_up((unsigned)(n_cams * n_tiles), kBlock);
  hipLaunchKernelGGL(offset_encode_kernel, dim3(gblk), dim3(kBlock), 0, (hipStream_t)stream,
                     n_isect, isect_ids, n_cams, n_tiles, tile_bits, offsets);
  return check_launch("isect_offset_encode");
}
