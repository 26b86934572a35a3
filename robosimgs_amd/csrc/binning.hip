// binning.hip -- tile binning for gfx950: depth-ordered per-tile Gaussian lists.
//
// The textbook pipeline (gsplat `isect_tiles`) emits one 64-bit key (tile << 32 | depth bits)
// per (Gaussian, tile) pair and radix-sorts all n_isect pairs on ~45 bits: six 8-bit passes
// over 12-byte elements.  The same ordering is produced here in five launches and two trips over the pairs:
//   1. per Gaussian: tile rectangle and tile count (written by the fused projection kernel, or by
//      tile_count_kernel for the standalone operator);
//   2. DIRECT PATH (tile grids up to ~65 k tiles, below): a counting sort of the pairs on their tile GROUP (four
//      consecutive tiles) with LDS-resident counters -- histogram per run of Gaussians, column scan, scatter;
//   3. tile_sort.hip: one workgroup per tile picks its entries out of its group's segment and orders them by
//      (depth bits, index); it also stores the tile's offset and the entries' tile ids.
// Sorting each tile's list by the full (depth, index) composite == the stable sort on (tile, depth) with
// index-order ties: flatten_ids / isect_ids are bit-identical to the single-key
// formulation, whatever order step 2 delivers the entries in.
// RADIX PATH (larger grids; mgs_debug_set_sort_opts(4) forces it): exclusive scan of the counts in index order,
// load-balanced emit of (tile, gaussian) pairs, stable radix sort on the tile bits only (13 bits at 1080p: a
// 7-bit and a 6-bit pass over 8-byte pairs), first index of every tile, then the same per-tile sort: ten launches,
// 131 us at config 2 against 98 for the direct path (profiles/r2/00_experiments.md).  The index-order scan also
// gives the backward's record slots (pair_info), so training runs it on either path.
#include "mgs_common.h"
#include "tile_rect.h"
#include "tile_order.h"

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
    const int32_t* __restrict__ radii_y, const float* __restrict__ conics, const float* __restrict__ opacities,
    float tile_size, int tile_w, int tile_h, uint2* __restrict__ ginfo, uint32_t* __restrict__ sums) {
  int g = blockIdx.x * kBlock + threadIdx.x;
  int radius = g < n ? radii[g] : 0;
  uint2 info = make_uint2(kEmptyTileRect, 0u);
  if (radius > 0) {
    float2 m = reinterpret_cast<const float2*>(means2d)[g];
    TileRect r = tile_rect(m.x, m.y, radius, radii_y ? radii_y[g] : radius, tile_size, tile_w, tile_h);
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
template <int kScanThreads>
__device__ __forceinline__ void scan_blocksums_body(
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
__global__ __launch_bounds__(kScanThreads) void scan_blocksums_kernel(
    uint32_t nblk, uint32_t* __restrict__ blocksums, uint32_t capacity,
    uint32_t* __restrict__ n_isect, uint32_t* __restrict__ status) {
  scan_blocksums_body<kScanThreads>(nblk, blocksums, capacity, n_isect, status);
}

// Slots for the deterministic backward, Gaussian-index-major: Gaussian g owns the slots
// [base, base + w*h) with base = exclusive scan of the tile counts in INDEX order, so that the
// per-Gaussian reduction reads contiguous memory from consecutive lanes.
// splat_slots (nullable): the packed 48-byte splat records [N,12]; words 10 and 11 of a Gaussian's record -- padding of the
// forward -- receive its slot base and its tile rectangle x0 | y0 << 10 | w << 20, so that the raster backward, which has
// the record in registers anyway, needs no gather of pair_info (16 bytes out of a 128-byte line per queued entry).
__device__ __forceinline__ void store_splat_slots(float* __restrict__ splat_slots, int g, uint32_t slot_base, uint32_t rect) {
  if (splat_slots)
    *reinterpret_cast<uint2*>(splat_slots + 12 * (size_t)g + 10) = make_uint2(slot_base, rect & 0x3fffffffu);
}

__global__ __launch_bounds__(kBlock) void pair_info_kernel(
    int n, const uint2* __restrict__ ginfo, int tile_h, const uint32_t* __restrict__ blockbase,
    int4* __restrict__ pair_info, float* __restrict__ splat_slots) {
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
    const uint32_t slot_base = blockbase[blockIdx.x * (kBlock / kSum)] + off + incl - cnt;
    pair_info[g] = cnt ? make_int4((int)slot_base, (int)(info.x & 1023u),
                                   (int)((info.x >> 10) & 1023u), (int)(w | ((cnt / w) << 16)))
                       : make_int4(0, 0, 0, 0);
    if (cnt) store_splat_slots(splat_slots, g, slot_base, info.x);
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


// ---- direct path (up to kDirectMaxTiles bins) ---------------------------------------------------------------
// The per-tile sort orders a list by the full (depth bits, index) composite, so the order in which a tile's
// entries arrive does not matter and the radix partition by tile (emit + two histogram / scan / scatter passes
// + tile offsets: nine launches, four trips over the pairs) can be replaced by a counting sort whose counters
// live in LDS.  A bin is a group of 2^shift consecutive tiles (four by default, kGroupShift):
//   direct_hist_kernel     each of <= kDirectMaxBlocks workgroups owns a run of consecutive Gaussians and counts
//                          their tiles into an LDS histogram over all bins (LDS atomics), then stores its row of
//                          the [workgroup][bin] table;
//   direct_colscan_kernel  per bin, exclusive prefix down the table's column (16 bins x 16 row groups per
//                          workgroup) and the bin's total;
//   direct_scatter_kernel  every workgroup scans the bin totals into the bin offsets (in LDS; workgroup 0 also
//                          stores them, with n_isect and the overflow status), adds its own row = its cursors,
//                          walks its Gaussians again and drops each pair at the cursor's next position (LDS
//                          atomic with return) as id | tile's place in the group << (32 - shift); one more
//                          workgroup computes the raster kernels' launch order (tile_order.h) meanwhile.
// tile_sort.hip then lets every tile pick its entries out of its group's segment.
// Device-scope atomics were measured for the same job and are no option: ~25 G atomics/s on 8160 hot counters
// (scripts/ubench/atomic_rate.hip), i.e. 108 us per pass at config 2.
#ifndef MGS_DIRECT_DEAL
#define MGS_DIRECT_DEAL 1        // runs of Gaussians dealt round-robin to the histogram / scatter workgroups (DealtIndex)
#endif
#ifndef MGS_DIRECT_DEAL_RUN
#define MGS_DIRECT_DEAL_RUN 256
#endif
#ifndef MGS_DIRECT_THREADS
#define MGS_DIRECT_THREADS 512
#endif
// Workgroup of the histogram / scatter kernels and Gaussians per thread.  Alone the stage likes big workgroups
// (1024 x 4: 110.7 us, 512 x 8: 113.3, 512 x 4: 118.1, 256 x 8: 125.0, 256 x 4: 137.8 -- fewer table rows, fewer
// copies of the scatter's prologue), but a 16-wave workgroup has to wait for a quarter of a CU to drain while other
// frames' raster waves hold the slots: with three frames in flight 1024 x 4 renders 3,281 frames/s, 512 x 8 3,669,
// 256 x 8 3,561 (the radix partition: 3,585).
constexpr int kDirectThreads = MGS_DIRECT_THREADS;
// DealtIndex covers the index range without holes, and the PAIRS path's "a wave is 64 consecutive Gaussians", only then:
static_assert(MGS_DIRECT_DEAL_RUN % 64 == 0 && kDirectThreads % MGS_DIRECT_DEAL_RUN == 0 && kDirectThreads >= MGS_DIRECT_DEAL_RUN,
              "MGS_DIRECT_THREADS must be a multiple of MGS_DIRECT_DEAL_RUN, and the run a multiple of the wave");
#ifndef MGS_DIRECT_PER_THREAD
// Round 4: 4 (was 8).  At 1 M Gaussians 512 x 8 is 245 workgroups -- fewer than CUs, two waves per SIMD -- and on a
// Morton-ordered scene their shares of the pairs differ by 2.7 x (15 k on average, 40 k at most: near Gaussians are
// neighbours in memory AND cover many tiles).  490 workgroups, rocprofv3 per kernel in bench.py's run: scatter 30.3 ->
// 25.7 us, histogram 14.3 -> 11.3, the column scan (twice the rows) 6.9 -> 11.6 (8.8 once it kept its rows in registers);
// three frames in flight 4,503 -> 4,561 frames/s.  With the scene in random order the shares are even and the extra rows
// cost the scatter ~4 us (fragmented stores), the same frames/s.
#define MGS_DIRECT_PER_THREAD 4
#endif
constexpr int kDirectPerThread = MGS_DIRECT_PER_THREAD;   // Gaussians per thread of those kernels
constexpr int kDirectMaxBlocks = 1024;         // table rows (beyond threads x per-thread x this many Gaussians: longer runs per workgroup)
constexpr int kDirectMaxTiles = 15000;         // the LDS histogram: 4 bytes per bin in 64 KiB, less the static part
constexpr int kGroupShift = 2;                 // 2^2 consecutive tiles share a segment (1 / 2 / 4 / 8 / 16 tiles: 124 / 113 / 112 / 124 / 150 us)
constexpr uint32_t kCoopRect = 24;             // rectangles above this many tiles are walked by the whole wave
constexpr uint32_t kRunRect = 128;             // ... above this many tiles (4K frames are full of rectangles of 25-100 tiles: one trip either way, and the runs' index arithmetic costs more)
#ifndef MGS_DIRECT_RUNS
#define MGS_DIRECT_RUNS 1        // ... run of tiles by run (one LDS atomic per bin a row touches: for_each_tile_run); 0: tile by tile
#endif

// f(tile, g) for every tile of every lane's rectangle; all 64 lanes must arrive together
template <class F>
__device__ __forceinline__ void for_each_tile(uint32_t pack, uint32_t cnt, uint32_t g, int tile_w, F f) {
  const bool big = cnt > kCoopRect;
  if (!big) {
    const uint32_t w = pack >> 20;
    uint32_t t = ((pack >> 10) & 1023u) * (uint32_t)tile_w + (pack & 1023u), dx = 0;
    for (uint32_t k = 0; k < cnt; ++k) {
      f(t + dx, g);
      if (++dx == w) { dx = 0; t += (uint32_t)tile_w; }
    }
  }
  unsigned long long m = ballot(big);
  const unsigned lane = lane_id();
  while (m) {
    const int src = __ffsll((long long)m) - 1;
    m &= m - 1;
    const uint32_t p = __shfl(pack, src), c = __shfl(cnt, src), gs = __shfl(g, src);
    const uint32_t w = p >> 20, x0 = p & 1023u, y0 = (p >> 10) & 1023u;
    for (uint32_t k = lane; k < c; k += 64) {
      const uint32_t dy = k / w, dx = k - dy * w;
      f((y0 + dy) * (uint32_t)tile_w + x0 + dx, gs);
    }
  }
}

// The same walk for the GROUPED partition (bins of 2^shift consecutive tiles): f(first tile, n, g) for every RUN of tiles of a
// rectangle's row that fall into one bin -- one LDS atomic per run instead of one per tile (a needle's box of 50 x 50 tiles, a
// screen-filling 120 x 68: a clustered scene's histogram took 3 x, its scatter 2.2 x their time on SURVEY 8(d)'s scene for 1.37 x
// the pairs).  Small rectangles go tile by tile as before (n = 1).  A row of w tiles starting at t0 touches at most R = (w + G - 2)
// / G + 1 bins; lanes are (row of the trip, run of the row) with R rounded up to a power of two: no division anywhere.
template <class F>
__device__ __forceinline__ void for_each_tile_run(uint32_t pack, uint32_t cnt, uint32_t g, int tile_w, int shift, F f) {
  const bool big = cnt > kCoopRect;
  if (!big) {
    const uint32_t w = pack >> 20;
    uint32_t t = ((pack >> 10) & 1023u) * (uint32_t)tile_w + (pack & 1023u), dx = 0;
    for (uint32_t k = 0; k < cnt; ++k) {
      f(t + dx, 1u, g);
      if (++dx == w) { dx = 0; t += (uint32_t)tile_w; }
    }
  }
  unsigned long long m = ballot(big);
  const unsigned lane = lane_id();
  const uint32_t G = 1u << shift;
  while (m) {
    const int src = __ffsll((long long)m) - 1;
    m &= m - 1;
    const uint32_t p = __shfl(pack, src), c = __shfl(cnt, src), gs = __shfl(g, src);
    const uint32_t w = p >> 20, x0 = p & 1023u, y0 = (p >> 10) & 1023u, h = c / w;      // (uniform: one scalar division per rectangle)
    const uint32_t R = (w + G - 2u) / G + 1u;                                           // runs of a row, at most
    const int lr = R > 1u ? 32 - __clz((int)(R - 1u)) : 0;                              // log2 of R rounded up to a power of two
    if (lr > 6 || c <= kRunRect) {       // (a row of more than 64 runs, or a rectangle of one or two trips anyway: tile by tile)
      for (uint32_t k = lane; k < c; k += 64) {
        const uint32_t dy = k / w, dx = k - dy * w;
        f((y0 + dy) * (uint32_t)tile_w + x0 + dx, 1u, gs);
      }
      continue;
    }
    const uint32_t rows_per_trip = 64u >> lr, r = lane >> lr, j = lane & ((1u << lr) - 1u);
    for (uint32_t dy0 = 0; dy0 < h; dy0 += rows_per_trip) {
      const uint32_t dy = dy0 + r;
      const uint32_t t0 = (y0 + dy) * (uint32_t)tile_w + x0, bin = (t0 >> shift) + j;
      const uint32_t lo = max(t0, bin << shift), hi = min(t0 + w, (bin + 1u) << shift);
      if (dy < h && lo < hi) f(lo, hi - lo, gs);
    }
  }
}

// Which Gaussian thread t of workgroup w takes as item i of the pass that starts at `base`.  DEALT (deal_nb = the number
// of workgroups; the default): runs of MGS_DIRECT_DEAL_RUN consecutive Gaussians go round-robin to the workgroups -- run
// ((pass * per-thread + i) * runs-per-pass + t / R) * deal_nb + w -- instead of one run of `chunk` per workgroup.  On a
// Morton-ordered scene the workgroups' shares of the pairs differed by 2.7 x (near Gaussians are neighbours in memory AND
// cover many tiles), and a scene whose large rectangles are contiguous in index (appended by a densifier, a background
// block) left two workgroups with most of the pairs; dealt, both even out: binning 77 -> 68 us at 1080p, 417 -> 357 us at
// 4K, a clustered scene's training step 1.70 -> 1.54 ms (profiles/r5/00_experiments.md 18, 20).  Runs of 128 ... 512
// measure alike, 64 loses (a wave's stores of a bin stop being one run).  deal_nb == 0: the consecutive runs.
// -1: no such Gaussian.
struct DealtIndex {
  int g0, g1, n, deal_nb;
  __device__ __forceinline__ int operator()(int base, int i) const {
    if (deal_nb > 0) {
      const int pass = (base - g0) / (kDirectThreads * kDirectPerThread);
      constexpr int R = MGS_DIRECT_DEAL_RUN, kSub = kDirectThreads / R;      // runs of R consecutive Gaussians
      const int sub = (int)threadIdx.x / R, within = (int)threadIdx.x % R;
      const long long g = (((long long)(pass * kDirectPerThread + i) * kSub + sub) * deal_nb + blockIdx.x) * R + within;
      return g < n ? (int)g : -1;
    }
    const int g = base + i * kDirectThreads + (int)threadIdx.x;
    return g < g1 ? g : -1;
  }
};

__global__ __launch_bounds__(kDirectThreads) void direct_hist_kernel(
    int n, int chunk, const uint2* __restrict__ ginfo, int tile_w, int n_tiles, int shift,
    uint32_t* __restrict__ table, int32_t* __restrict__ tiles_per_gauss, uint32_t* __restrict__ scan_sums, uint32_t n_sums,
    int deal_nb) {
  extern __shared__ uint32_t hist[];
  if (scan_sums && blockIdx.x == gridDim.x - 1) {
    // training only, one workgroup more than the histogram needs: the per-64 sums of the tile counts scanned in place
    // (the backward's record slots are the index-order scan of the counts; the scatter kernel finishes them per
    // Gaussian) -- beside the histogram instead of a launch of one workgroup in front of it
    scan_blocksums_body<kDirectThreads>(n_sums, scan_sums, 0u, nullptr, nullptr);
    return;
  }
  n_tiles = (n_tiles + (1 << shift) - 1) >> shift;           // bins: groups of 2^shift consecutive tiles
  for (int i = threadIdx.x; i < n_tiles; i += kDirectThreads) hist[i] = 0u;
  __syncthreads();
  const int g0 = blockIdx.x * chunk, g1 = min(n, g0 + chunk);
  // the workgroup's rectangles are fetched kDirectPerThread at a time, all loads in flight together: at 1 M Gaussians
  // the launch is 245 workgroups -- two waves per SIMD -- and a load per trip was one exposed round trip per trip
  const DealtIndex gi{g0, g1, n, deal_nb};
  const int g_end = deal_nb > 0 ? g0 + chunk : g1;            // (dealt: every workgroup makes every pass)
  for (int base = g0; base < g_end; base += kDirectThreads * kDirectPerThread) {
    uint2 info[kDirectPerThread];
#pragma unroll
    for (int i = 0; i < kDirectPerThread; ++i) {
      const int g = gi(base, i);
      info[i] = g >= 0 ? ginfo[g] : make_uint2(kEmptyTileRect, 0u);
    }
#pragma unroll
    for (int i = 0; i < kDirectPerThread; ++i) {
      const int g = gi(base, i);
      if (g >= 0 && tiles_per_gauss) tiles_per_gauss[g] = (int32_t)info[i].y;
      if (shift > 0 && MGS_DIRECT_RUNS)
        for_each_tile_run(info[i].x, info[i].y, (uint32_t)g, tile_w, shift, [&](uint32_t tile, uint32_t n_run, uint32_t) { atomicAdd(&hist[tile >> shift], n_run); });
      else
        for_each_tile(info[i].x, info[i].y, (uint32_t)g, tile_w, [&](uint32_t tile, uint32_t) { atomicAdd(&hist[tile >> shift], 1u); });
    }
  }
  __syncthreads();
  uint32_t* row = table + (size_t)blockIdx.x * n_tiles;
  for (int i = threadIdx.x; i < n_tiles; i += kDirectThreads) row[i] = hist[i];
}

// 256 threads: 16 bins (64 bytes of a row) x 16 row groups; two passes over the group's rows (sum, then rewrite as
// the exclusive prefix), 16 loads in flight per thread
#ifndef MGS_COLSCAN_BINS
// 256 threads: kColBins bins (4 bytes each, consecutive in a table row) x 256 / kColBins row groups.  rocprofv3 per
// kernel at the 490 table rows of 1 M Gaussians (two trips over the column, round 4): 16 bins x 16 groups 11.6 us,
// 8 x 32 11.3, 4 x 64 14.2, 32 x 8 16.4; 16 x 32 with 512 threads 2 us faster alone but 4,170 against 4,504 frames/s with
// three frames in flight (a 512-thread workgroup waits for room between other frames' raster waves).  With the rows kept
// in registers (one trip, below) 16 x 16 takes 8.8 us.
#define MGS_COLSCAN_BINS 16
#endif
#ifndef MGS_COLSCAN_KEEP
// Table rows a thread keeps in registers between the sum and the rewrite (one trip over the column instead of two) in a
// TRAINING step, where the launch has the GPU to itself: 11.6 -> 8.8 us at 32 (92 VGPRs).  Inference frames keep the
// two-trip scan (kColKeep 1, ~50 VGPRs): with three frames in flight the one-trip scan renders 4,421 against 4,503
// frames/s (3,846 against 3,927 in the caller's order) -- its fatter waves wait for room between the raster's.
#define MGS_COLSCAN_KEEP 32
#endif
constexpr int kColThreads = 256, kColBins = MGS_COLSCAN_BINS, kColGroups = kColThreads / kColBins;
template <int kColKeep>
__global__ __launch_bounds__(kColThreads) void direct_colscan_kernel(
    int nb, int n_tiles, uint32_t* __restrict__ table, uint32_t* __restrict__ tile_count) {
  __shared__ uint32_t part[kColGroups][kColBins];
  const int bl = threadIdx.x % kColBins, rg = threadIdx.x / kColBins;
  const int t = blockIdx.x * kColBins + bl;
  const int rpg = (nb + kColGroups - 1) / kColGroups;
  const int r0 = rg * rpg, r1 = min(nb, r0 + rpg);
  const bool ok = t < n_tiles;
  // up to kColKeep rows per thread (1 M Gaussians: 490 rows, 31 per thread; 64 would cover every table but needs 124
  // VGPRs) stay in registers between the sum and the rewrite: one trip over the column instead of two (the kernel is its
  // chain of load round trips).  kColKeep 1: the two-trip scan only.
  constexpr bool kCanKeep = kColKeep > 1;
  const bool keep = kCanKeep && rpg <= kColKeep;              // uniform
  uint32_t kept[kColKeep];
  uint32_t sum = 0;
  if (kCanKeep && keep) {
#pragma unroll
    for (int j = 0; j < kColKeep; ++j) {
      if (j >= rpg) { kept[j] = 0u; continue; }    // (uniform: whole 16-row pieces past the group's share are skipped)
      kept[j] = (ok && r0 + j < r1) ? table[(size_t)(r0 + j) * n_tiles + t] : 0u;
    }
#pragma unroll
    for (int j = 0; j < kColKeep; ++j) sum += kept[j];
  } else {
    for (int r = r0; r < r1; r += 16) {
      uint32_t v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = (ok && r + j < r1) ? table[(size_t)(r + j) * n_tiles + t] : 0u;
#pragma unroll
      for (int j = 0; j < 16; ++j) sum += v[j];
    }
  }
  part[rg][bl] = sum;
  __syncthreads();
  uint32_t off = 0, tot = 0;
#pragma unroll 16
  for (int g = 0; g < kColGroups; ++g) {
    const uint32_t s = part[g][bl];
    if (g < rg) off += s;
    tot += s;
  }
  if (kCanKeep && keep) {
#pragma unroll
    for (int j = 0; j < kColKeep; ++j) {
      if (j >= rpg) continue;
      if (ok && r0 + j < r1) {
        table[(size_t)(r0 + j) * n_tiles + t] = off;
        off += kept[j];
      }
    }
  } else {
    for (int r = r0; r < r1; r += 16) {
      uint32_t v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = (ok && r + j < r1) ? table[(size_t)(r + j) * n_tiles + t] : 0u;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (ok && r + j < r1) {
          table[(size_t)(r + j) * n_tiles + t] = off;
          off += v[j];
        }
    }
  }
  if (rg == 0 && ok) tile_count[t] = tot;
}

// PAIRS (training): also the record slots of the deterministic backward, Gaussian-index-major -- Gaussian g owns the
// slots [base, base + w * h) with base = the exclusive scan of the tile counts in INDEX order (scanned_sums: per 64
// Gaussians, by the histogram launch's extra workgroup; the wave finishes it), so that the per-Gaussian reduction reads
// contiguous memory from consecutive lanes.  (pair_info_kernel, for the radix path, is the same as a launch of its own.)
template <bool PAIRS>
__global__ __launch_bounds__(kDirectThreads) void direct_scatter_kernel(
    int n, int chunk, const uint2* __restrict__ ginfo, int tile_w, int n_tiles, int shift,
    const uint32_t* __restrict__ table, const uint32_t* __restrict__ tile_count, uint32_t capacity,
    uint32_t* __restrict__ flatten_ids, int32_t* __restrict__ tile_offsets,
    uint32_t* __restrict__ n_isect, uint32_t* __restrict__ status, int32_t* __restrict__ group_order,
    const uint32_t* __restrict__ scanned_sums, int4* __restrict__ pair_info, uint32_t* __restrict__ zero_word,
    float* __restrict__ splat_slots, int deal_nb) {
  extern __shared__ uint32_t cursor[];
  n_tiles = (n_tiles + (1 << shift) - 1) >> shift;           // bins (see direct_hist_kernel)
  if (group_order && blockIdx.x == gridDim.x - 1) {
    // one workgroup more than the scatter needs: the launch order of the raster kernels (tile_order.h) from the
    // bin totals -- bins are groups of four tiles here (shift == 2) -- beside the scatter, at no cost in time
    order_groups_by_total<kDirectThreads>(n_tiles, [&](int g) { return tile_count[g]; }, group_order);
    return;
  }
  const uint32_t local_mask = (1u << shift) - 1u;
  __shared__ unsigned long long wsum[kDirectThreads / 64];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // exclusive scan of the tile totals: `per` consecutive tiles per thread (64-bit: the total may pass 2^32)
  const int per = (n_tiles + kDirectThreads - 1) / kDirectThreads;
  const int t0 = (int)threadIdx.x * per;
  unsigned long long sum = 0;
  for (int k = 0; k < per; ++k)
    if (t0 + k < n_tiles) sum += tile_count[t0 + k];
  unsigned long long incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t lo = __shfl_up((uint32_t)incl, d), hi = __shfl_up((uint32_t)(incl >> 32), d);
    if (lane >= (unsigned)d) incl += ((unsigned long long)hi << 32) | lo;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  unsigned long long off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kDirectThreads / 64; ++w) {
    if ((unsigned)w < wave) off += wsum[w];
    total += wsum[w];
  }
  unsigned long long ex = off + incl - sum;
  const uint32_t* row = table + (size_t)blockIdx.x * n_tiles;
  for (int k = 0; k < per; ++k) {
    const int t = t0 + k;
    if (t < n_tiles) {
      // past the capacity nothing is stored and the lists are cut there (status says so)
      const uint32_t at = ex < capacity ? (uint32_t)ex : capacity;
      cursor[t] = ex < capacity ? at + row[t] : capacity;
      if (blockIdx.x == 0) tile_offsets[t] = (int32_t)at;
      ex += tile_count[t];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    tile_offsets[n_tiles] = (int32_t)(total < capacity ? (uint32_t)total : capacity);
    *n_isect = total > 0xffffffffull ? 0xffffffffu : (uint32_t)total;
    *status = total > capacity ? MGS_STATUS_ISECT_OVERFLOW : 0u;
    // the header of the per-tile sort's list of long tiles (tile_sort.hip: counts of long lists, giant descriptors, pool
    // entries), instead of a memset
    if (zero_word) reinterpret_cast<uint4*>(zero_word)[0] = reinterpret_cast<uint4*>(zero_word)[1] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  const int g0 = blockIdx.x * chunk, g1 = min(n, g0 + chunk);
  const DealtIndex gi{g0, g1, n, deal_nb};                                        // (the histogram launch's assignment)
  const int g_end = deal_nb > 0 ? g0 + chunk : g1;
  for (int base = g0; base < g_end; base += kDirectThreads * kDirectPerThread) {   // (loads in flight together: direct_hist_kernel)
    uint2 info[kDirectPerThread];
#pragma unroll
    for (int i = 0; i < kDirectPerThread; ++i) {
      const int g = gi(base, i);
      info[i] = g >= 0 ? ginfo[g] : make_uint2(kEmptyTileRect, 0u);
    }
    if (PAIRS) {
      // (a wave's 64 lanes are 64 consecutive Gaussians starting at a multiple of 64: chunk and the pass are multiples of 512)
      uint32_t sbase[kDirectPerThread];
#pragma unroll
      for (int i = 0; i < kDirectPerThread; ++i) {
        const int g = gi(base, i);
        sbase[i] = g >= 0 ? scanned_sums[g / kSum] : 0u;
      }
#pragma unroll
      for (int i = 0; i < kDirectPerThread; ++i) {
        const int g = gi(base, i);
        const uint32_t cnt = info[i].y;
        uint32_t incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t t = __shfl_up(incl, d);
          if (lane >= (unsigned)d) incl += t;
        }
        if (g >= 0) {
          const uint32_t w = info[i].x >> 20;
          pair_info[g] = cnt ? make_int4((int)(sbase[i] + incl - cnt), (int)(info[i].x & 1023u), (int)((info[i].x >> 10) & 1023u),
                                         (int)(w | ((cnt / w) << 16)))
                             : make_int4(0, 0, 0, 0);
          if (cnt) store_splat_slots(splat_slots, g, sbase[i] + incl - cnt, info[i].x);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < kDirectPerThread; ++i) {
      const int g = gi(base, i);
      if (shift > 0 && MGS_DIRECT_RUNS)
        for_each_tile_run(info[i].x, info[i].y, (uint32_t)g, tile_w, shift, [&](uint32_t tile, uint32_t n_run, uint32_t gs) {
          const uint32_t p = atomicAdd(&cursor[tile >> shift], n_run);
          // grouped: the entry carries its tile's place in the group above the id (ids < 2^(32 - shift))
          for (uint32_t k = 0; k < n_run; ++k)
            if (p + k < capacity) flatten_ids[p + k] = gs | (((tile + k) & local_mask) << (32 - shift));
        });
      else
      for_each_tile(info[i].x, info[i].y, (uint32_t)g, tile_w, [&](uint32_t tile, uint32_t gs) {
        const uint32_t p = atomicAdd(&cursor[tile >> shift], 1u);
        // grouped: the entry carries its tile's place in the group above the id (ids < 2^(32 - shift))
        if (p < capacity) flatten_ids[p] = shift ? gs | ((tile & local_mask) << (32 - shift)) : gs;
      });
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

__global__ __launch_bounds__(1024) void tile_group_order_kernel(int n_tiles, const int32_t* __restrict__ tile_offsets,
                                                                int32_t* __restrict__ order) {
  order_groups_by_total<1024>((n_tiles + 3) / 4, [&](int g) {
    return (uint32_t)(tile_offsets[min(4 * g + 4, n_tiles)] - tile_offsets[4 * g]);
  }, order);
}

int bits_for(uint32_t count) {   // bits needed to hold values 0..count-1
  int b = 0;
  while (count > 1 && (1ull << b) < count) ++b;
  return b;
}

// Gaussians per workgroup of the direct path's histogram / scatter kernels, and how many workgroups that makes
unsigned direct_chunk(unsigned n) {
  const unsigned per_block = max(div_up(n, kDirectMaxBlocks), (unsigned)(kDirectThreads * kDirectPerThread));
  return div_up(per_block, kDirectThreads) * kDirectThreads;
}
unsigned direct_blocks(unsigned n) { return div_up(n, direct_chunk(n)); }

struct Workspace {
  size_t total;
  size_t ginfo, blocksums, tile_alt, tile_priv, id_alt, radix, tsort, table, tile_count, group_offsets;
  Workspace(int n, uint32_t cap, int n_tiles) {
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align_up(bytes, 256); return at; };
    size_t nn = (size_t)(n > 0 ? n : 1), cc = cap ? cap : 1;
    ginfo = take(nn * 8);
    blocksums = take((size_t)div_up((unsigned)nn, kSum) * 4);
    tile_alt = take(cc * 4); id_alt = take(cc * 4);
    tile_priv = take(cc * 4);          // stands in for the caller's tile_ids when that is null
    radix = take(radix_sort_temp_bytes((uint32_t)cc));
    tsort = take(tile_depth_sort_temp_bytes((uint32_t)cc, n_tiles));
    const size_t nt = (size_t)(n_tiles < kDirectMaxTiles ? n_tiles : kDirectMaxTiles);   // bins: tiles or tile groups
    table = take((size_t)direct_blocks((unsigned)nn) * nt * 4);
    tile_count = take(nt * 4);
    group_offsets = take((nt + 1) * 4);
    total = o;
  }
};

}  // namespace
}  // namespace mgs

namespace mgs {
int launch_tile_group_order(int n_tiles, const int32_t* tile_offsets, int32_t* order, hipStream_t stream) {
  hipLaunchKernelGGL(tile_group_order_kernel, dim3(1), dim3(1024), 0, stream, n_tiles, tile_offsets, order);
  return check_launch("tile_group_order");
}
}  // namespace mgs

using namespace mgs;

extern "C" int mgs_isect_tiles(int n, const float* means2d, const int32_t* radii, const int32_t* radii_y,
                               const float* depths, const float* conics, const float* opacities,
                               int tile_size, int tile_w, int tile_h,
                               int cam_id, int n_cams, uint32_t isect_capacity,
                               int32_t* tiles_per_gauss, uint32_t* n_isect, uint32_t* tile_ids,
                               int32_t* flatten_ids, int64_t* isect_ids, int32_t* tile_offsets,
                               int32_t* pair_info, int32_t* tile_group_order, uint32_t* status,
                               const uint32_t* seed_info,
                               uint32_t* seed_sums, float* splat_slots, void* workspace,
                               size_t* workspace_bytes, mgs_stream_t stream) {
  MGS_REQUIRE(n >= 0 && tile_size > 0 && tile_w > 0 && tile_h > 0, "isect_tiles: bad sizes");
  MGS_REQUIRE(tile_w <= 1023 && tile_h <= 1023, "isect_tiles: tile grid %dx%d exceeds 1023x1023", tile_w, tile_h);
  MGS_REQUIRE(workspace_bytes, "isect_tiles: workspace_bytes is null");
  MGS_REQUIRE(cam_id >= 0 && n_cams > cam_id, "isect_tiles: cam_id %d outside 0..%d", cam_id, n_cams);
  Workspace ws(n, isect_capacity, tile_w * tile_h);
  if (!workspace) {
    *workspace_bytes = ws.total;
    return MGS_OK;
  }
  if (*workspace_bytes < ws.total)
    return set_error(MGS_ERR_WORKSPACE_TOO_SMALL, "isect_tiles: workspace %zu < %zu bytes",
                     *workspace_bytes, ws.total);
  MGS_REQUIRE(isect_capacity > 0, "isect_tiles: zero capacity");
  MGS_REQUIRE(seed_info || (conics == nullptr) == (opacities == nullptr),
              "isect_tiles: tight tile bounds need both conics and opacities");
  MGS_REQUIRE(n == 0 || (seed_info == nullptr) == (seed_sums == nullptr),
              "isect_tiles: seed_info and seed_sums come together (mgs_project_color_fwd writes both)");
  MGS_REQUIRE((n == 0 || ((seed_info || (means2d && radii)) && depths)) && n_isect && flatten_ids &&
                  tile_offsets && status, "isect_tiles: null pointer");
  // the seed is what mgs_project_color_fwd computed: rectangles at MGS_TILE_SIZE; its per-64 sums are scanned
  // in place with 16-byte accesses
  MGS_REQUIRE(!seed_info || tile_size == MGS_TILE_SIZE, "isect_tiles: a seed is for tile_size %d, got %d", MGS_TILE_SIZE, tile_size);
  MGS_REQUIRE(!seed_sums || (reinterpret_cast<uintptr_t>(seed_sums) & 15u) == 0, "isect_tiles: seed_sums must be 16-byte aligned");
  MGS_REQUIRE(!splat_slots || pair_info, "isect_tiles: splat_slots are written with pair_info");
  MGS_REQUIRE(!splat_slots || (reinterpret_cast<uintptr_t>(splat_slots) & 15u) == 0, "isect_tiles: splat_slots must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  char* w = static_cast<char*>(workspace);
  auto u32 = [&](size_t off) { return reinterpret_cast<uint32_t*>(w + off); };
  // tile_ids is optional: an inference frame never reads it.  The direct path then skips the store (15 MB at
  // 3.7 M pairs); the radix path and the isect_ids output keep a private copy in the workspace.
  const bool want_tile_ids = tile_ids != nullptr;
  if (!tile_ids) tile_ids = u32(ws.tile_priv);
  const int n_tiles = tile_w * tile_h;
  const uint32_t cap = isect_capacity;
  // tiles per segment of the direct path: 2^gshift (sort_opts bits 4..6 override; 0 = one tile per segment)
  const int gshift = (sort_opts() & 8) ? (sort_opts() >> 4) & 7 : kGroupShift;
  const bool direct = ((n_tiles + (1 << gshift) - 1) >> gshift) <= kDirectMaxTiles && !(sort_opts() & 4) &&
                      (gshift == 0 || (unsigned)n <= (1u << (32 - gshift)));
  int rc;
  bool order_done = false;

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
      hipLaunchKernelGGL(tile_count_kernel, dim3(nblk), dim3(kBlock), 0, s, n, means2d, radii, radii_y, conics,
                         opacities, (float)tile_size, tile_w, tile_h, gi, sums);
      ginfo = gi;
    }
    if (direct) {
      // (training: the backward's record slots -- pair_info -- come out of the histogram and scatter launches)
      const int chunk = (int)direct_chunk((unsigned)n);
      const unsigned nb = direct_blocks((unsigned)n);
      const int bins = (n_tiles + (1 << gshift) - 1) >> gshift;
      const size_t lds = (size_t)bins * sizeof(uint32_t);
      // runs of Gaussians are dealt round-robin to the workgroups (DealtIndex)
      const int deal_nb = MGS_DIRECT_DEAL ? (int)nb : 0;
      hipLaunchKernelGGL(direct_hist_kernel, dim3(nb + (pair_info ? 1 : 0)), dim3(kDirectThreads), lds, s, n, chunk, ginfo, tile_w,
                         n_tiles, gshift, u32(ws.table), tiles_per_gauss, pair_info ? sums : nullptr, nsum, deal_nb);
      if (pair_info)
        hipLaunchKernelGGL(direct_colscan_kernel<MGS_COLSCAN_KEEP>, dim3(div_up((unsigned)bins, (unsigned)kColBins)), dim3(kColThreads), 0, s,
                           (int)nb, bins, u32(ws.table), u32(ws.tile_count));
      else
        hipLaunchKernelGGL(direct_colscan_kernel<1>, dim3(div_up((unsigned)bins, (unsigned)kColBins)), dim3(kColThreads), 0, s,
                           (int)nb, bins, u32(ws.table), u32(ws.tile_count));
      const bool order_in_scatter = tile_group_order && gshift == 2;
      order_done = order_in_scatter;
#define MGS_SCATTER(P)                                                                                                   \
      hipLaunchKernelGGL(direct_scatter_kernel<P>, dim3(nb + (order_in_scatter ? 1 : 0)), dim3(kDirectThreads), lds, s, n, \
                         chunk, ginfo, tile_w, n_tiles, gshift, u32(ws.table), u32(ws.tile_count), cap,                   \
                         gshift ? u32(ws.id_alt) : reinterpret_cast<uint32_t*>(flatten_ids),                             \
                         gshift ? reinterpret_cast<int32_t*>(w + ws.group_offsets) : tile_offsets, n_isect, status,      \
                         order_in_scatter ? tile_group_order : nullptr, sums, reinterpret_cast<int4*>(pair_info),           \
                         tile_depth_sort_long_list(w + ws.tsort, cap), pair_info ? splat_slots : nullptr, deal_nb)
      if (pair_info) MGS_SCATTER(true);
      else MGS_SCATTER(false);
#undef MGS_SCATTER
    } else {
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
                           reinterpret_cast<int4*>(pair_info), splat_slots);
      rc = radix_sort_pairs(n_isect, cap, tile_bits, a_t, a_i, b_t, b_i, w + ws.radix, s);
      if (rc) return rc;
    }
  }
  if (!direct || n == 0)
    hipLaunchKernelGGL(tile_offsets_kernel, dim3(div_up(cap, kBlock * kOffsetsPerThread)), dim3(kBlock), 0,
                       s, n_isect, cap, tile_ids, n_tiles, tile_offsets);
  if (n > 0) {        // depth order inside every tile's list (the direct path's lists get their tile ids here)
    const bool grouped = direct && gshift > 0;
    rc = tile_depth_sort(n_tiles, tile_offsets, depths, cap, reinterpret_cast<uint32_t*>(flatten_ids),
                         direct && (want_tile_ids || isect_ids) ? tile_ids : nullptr, w + ws.tsort, s,
                         /*scratch: the radix ping-pong buffers, dead by now (the direct path's staging is id_alt)*/ u32(ws.tile_alt), u32(ws.id_alt),
                         grouped ? u32(ws.id_alt) : nullptr,
                         grouped ? reinterpret_cast<const int32_t*>(w + ws.group_offsets) : nullptr, gshift, /*long_list_zeroed=*/direct);
    if (rc) return rc;
  }
  if (tile_group_order && !order_done) {
    rc = launch_tile_group_order(n_tiles, tile_offsets, tile_group_order, s);
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
