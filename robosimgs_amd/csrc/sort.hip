// sort.hip -- stable LSD radix sort of (uint32 key, uint32 value) pairs for gfx950, with the
// element count resident in device memory (no host read-back anywhere in the frame).
//
// 8 bits per pass, three kernels per pass:
//   hist     per-block digit counts                     -> counts[digit][block]
//   rowscan  one workgroup per digit scans its row      -> exclusive prefix per (digit, block),
//                                                          digit totals
//   scatter  wave64 ballot match ranks every element among its equal-digit peers
//            (8 ballots + mbcnt), adds the wave / block / global bases and writes it out
// Grids are sized from the caller's capacity; workgroups past ceil(n / kTile) exit at once.
// Byte / integer work, HBM-bound: per pass each element is read twice and written once.
#include "mgs_common.h"

namespace mgs {
namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kItems = 8;                       // per thread
constexpr int kTile = kThreads * kItems;        // 2048 elements per workgroup
constexpr int kWaveTile = 64 * kItems;          // 512 per wave
constexpr int kRadix = 256;

__device__ __forceinline__ unsigned digit_of(uint32_t key, int shift) {
  return (key >> shift) & (kRadix - 1);
}

__global__ __launch_bounds__(kThreads) void radix_hist_kernel(
    const uint32_t* __restrict__ n_ptr, uint32_t capacity, int shift,
    const uint32_t* __restrict__ keys, uint32_t* __restrict__ counts, uint32_t nblk_cap) {
  __shared__ uint32_t hist[kRadix];
  uint32_t n = min(*n_ptr, capacity);
  uint32_t nblk = (n + kTile - 1) / kTile;
  uint32_t blk = blockIdx.x;
  if (blk >= nblk) return;
  hist[threadIdx.x] = 0;
  __syncthreads();
  uint32_t base = blk * kTile;
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    uint32_t idx = base + i * kThreads + threadIdx.x;
    if (idx < n) atomicAdd(&hist[digit_of(keys[idx], shift)], 1u);
  }
  __syncthreads();
  counts[(size_t)threadIdx.x * nblk_cap + blk] = hist[threadIdx.x];
}

// block-wide exclusive scan helper over kThreads values (one per thread)
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wave_sums,
                                                         uint32_t* total) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(incl, d);
    if (lane >= (unsigned)d) incl += t;
  }
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  uint32_t off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    uint32_t s = wave_sums[w];
    if ((unsigned)w < wave) off += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return off + incl - v;
}

__global__ __launch_bounds__(kThreads) void radix_rowscan_kernel(
    const uint32_t* __restrict__ n_ptr, uint32_t capacity, uint32_t* __restrict__ counts,
    uint32_t* __restrict__ digit_totals, uint32_t nblk_cap) {
  __shared__ uint32_t wave_sums[kWaves];
  uint32_t n = min(*n_ptr, capacity);
  uint32_t nblk = (n + kTile - 1) / kTile;
  uint32_t* row = counts + (size_t)blockIdx.x * nblk_cap;
  uint32_t running = 0;
  for (uint32_t b0 = 0; b0 < nblk; b0 += kThreads) {
    uint32_t b = b0 + threadIdx.x;
    uint32_t v = b < nblk ? row[b] : 0u;
    uint32_t tot;
    uint32_t ex = block_exclusive_scan(v, wave_sums, &tot);
    if (b < nblk) row[b] = running + ex;
    running += tot;
  }
  if (threadIdx.x == 0) digit_totals[blockIdx.x] = running;
}

__global__ __launch_bounds__(kThreads) void radix_scatter_kernel(
    const uint32_t* __restrict__ n_ptr, uint32_t capacity, int shift,
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
    const uint32_t* __restrict__ counts, const uint32_t* __restrict__ digit_totals,
    uint32_t nblk_cap) {
  __shared__ uint32_t wave_hist[kWaves][kRadix];   // per-wave digit counts -> bases
  __shared__ uint32_t wave_sums[kWaves];
  uint32_t n = min(*n_ptr, capacity);
  uint32_t nblk = (n + kTile - 1) / kTile;
  uint32_t blk = blockIdx.x;
  if (blk >= nblk) return;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) wave_hist[w][threadIdx.x] = 0;
  __syncthreads();

  // phase A: rank inside the wave's 512-element slice (rounds of 64 consecutive elements)
  uint32_t key[kItems], val[kItems], rank[kItems];
  const uint32_t wbase = blk * kTile + wave * kWaveTile;
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    uint32_t idx = wbase + r * 64 + lane;
    bool valid = idx < n;
    key[r] = valid ? keys_in[idx] : 0xffffffffu;
    val[r] = valid ? vals_in[idx] : 0u;
    unsigned d = digit_of(key[r], shift);
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    unsigned below = mask_rank(peers);
    unsigned cnt = __popcll(peers);
    uint32_t prev = 0;
    if (valid && below == 0) prev = atomicAdd(&wave_hist[wave][d], cnt);   // leader of the peer group
    int leader = __ffsll((long long)peers) - 1;
    prev = __shfl(prev, leader < 0 ? 0 : leader);
    rank[r] = prev + below;
  }
  __syncthreads();

  // phase B: thread t owns digit t: base = global digit offset + this block's prefix in the
  // digit's row + counts of the earlier waves of this block
  {
    uint32_t tot;
    uint32_t digit_base = block_exclusive_scan(digit_totals[threadIdx.x], wave_sums, &tot);
    uint32_t run = digit_base + counts[(size_t)threadIdx.x * nblk_cap + blk];
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      uint32_t c = wave_hist[w][threadIdx.x];
      wave_hist[w][threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();

  // phase C: scatter
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    uint32_t idx = wbase + r * 64 + lane;
    if (idx < n) {
      uint32_t dst = wave_hist[wave][digit_of(key[r], shift)] + rank[r];
      keys_out[dst] = key[r];
      vals_out[dst] = val[r];
    }
  }
}

}  // namespace

size_t radix_sort_temp_bytes(uint32_t capacity) {
  size_t nblk_cap = div_up(capacity ? capacity : 1u, kTile);
  return align_up((kRadix * nblk_cap + kRadix) * sizeof(uint32_t), 256);
}

// Result lands in (keys_b, vals_b) when the pass count ceil(key_bits/8) is odd, else in
// (keys_a, vals_a); callers pick their buffers with radix_sort_passes().
int radix_sort_pairs(const uint32_t* n_dev, uint32_t capacity, int key_bits, uint32_t* keys_a,
                     uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, void* temp,
                     hipStream_t stream) {
  if (capacity == 0) return MGS_OK;
  uint32_t nblk_cap = div_up(capacity, kTile);
  uint32_t* counts = static_cast<uint32_t*>(temp);
  uint32_t* totals = counts + (size_t)kRadix * nblk_cap;
  int passes = (key_bits + 7) / 8;
  for (int p = 0; p < passes; ++p) {
    int shift = 8 * p;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nblk_cap), dim3(kThreads), 0, stream, n_dev,
                       capacity, shift, keys_a, counts, nblk_cap);
    hipLaunchKernelGGL(radix_rowscan_kernel, dim3(kRadix), dim3(kThreads), 0, stream, n_dev,
                       capacity, counts, totals, nblk_cap);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblk_cap), dim3(kThreads), 0, stream, n_dev,
                       capacity, shift, keys_a, vals_a, keys_b, vals_b, counts, totals, nblk_cap);
    uint32_t* t;
    t = keys_a; keys_a = keys_b; keys_b = t;
    t = vals_a; vals_a = vals_b; vals_b = t;
  }
  return check_launch("radix_sort_pairs");
}

}  // namespace mgs
