// sort.hip -- stable LSD radix sort of (uint32 key, uint32 value) pairs for gfx950, with the
// element count resident in device memory (no host read-back anywhere in the frame).
//
// Up to 8 bits per pass, three kernels per pass:
//   hist     per-block digit counts                     -> counts[digit][block]
//   rowscan  one workgroup per digit scans its row      -> exclusive prefix per (digit, block),
//                                                          digit totals
//   scatter  wave64 ballot match ranks every element among its equal-digit peers
//            (ballots + mbcnt), the block's elements are re-ordered by digit in LDS, and each
//            digit's run then leaves as ONE contiguous, lane-linear store (a 4-byte scatter
//            straight to HBM defeats write combining: measured 64 us per 5 M-element pass
//            against ~20 us staged).
// Grids are sized from the caller's capacity; workgroups past ceil(n / tile) exit at once.
// Block tile: 1024 elements for small inputs (enough workgroups to fill 256 CUs), 4096 for
// large ones (longer per-digit runs).  Byte / integer work, HBM-bound: per pass each element is
// read twice and written once.  The render path sorts the (tile, Gaussian) pairs on their tile bits
// with it (two passes at 1080p); the depth order inside a tile is tile_sort.hip's job.
#include "mgs_common.h"

namespace mgs {
namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kRadix = 256;
constexpr uint32_t kSmallLimit = 2u << 20;     // capacities up to 2 M use the small tile

template <int ITEMS>
struct Cfg {
  static constexpr int kTile = kThreads * ITEMS;
  static constexpr int kWaveTile = 64 * ITEMS;
};

__device__ __forceinline__ unsigned digit_of(uint32_t key, int shift, uint32_t mask) {
  return (key >> shift) & mask;
}

template <int ITEMS>
__global__ __launch_bounds__(kThreads) void radix_hist_kernel(
    const uint32_t* __restrict__ n_ptr, uint32_t capacity, int shift, uint32_t mask,
    const uint32_t* __restrict__ keys, uint32_t* __restrict__ counts, uint32_t nblk_cap) {
  constexpr int kTile = Cfg<ITEMS>::kTile;
  __shared__ uint32_t hist[kRadix];
  uint32_t n = min(*n_ptr, capacity);
  uint32_t nblk = (n + kTile - 1) / kTile;
  uint32_t blk = blockIdx.x;
  if (blk >= nblk) return;
  hist[threadIdx.x] = 0;
  __syncthreads();
  uint32_t base = blk * kTile;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    uint32_t idx = base + i * kThreads + threadIdx.x;
    if (idx < n) atomicAdd(&hist[digit_of(keys[idx], shift, mask)], 1u);
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
    const uint32_t* __restrict__ n_ptr, uint32_t capacity, uint32_t tile,
    uint32_t* __restrict__ counts, uint32_t* __restrict__ digit_totals, uint32_t nblk_cap) {
  __shared__ uint32_t wave_sums[kWaves];
  uint32_t n = min(*n_ptr, capacity);
  uint32_t nblk = (n + tile - 1) / tile;
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

template <int ITEMS, int BITS>
__global__ __launch_bounds__(kThreads) void radix_scatter_kernel(
    const uint32_t* __restrict__ n_ptr, uint32_t capacity, int shift, uint32_t mask,
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
    const uint32_t* __restrict__ counts, const uint32_t* __restrict__ digit_totals,
    uint32_t nblk_cap) {
  constexpr int kTile = Cfg<ITEMS>::kTile, kWaveTile = Cfg<ITEMS>::kWaveTile;
  __shared__ uint32_t wave_hist[kWaves][kRadix];   // per-wave digit counts -> per-wave local bases
  __shared__ uint32_t gdst[kRadix];                // global address of the digit's run minus its local start
  __shared__ uint32_t wave_sums[kWaves];
  __shared__ uint32_t stage_k[kTile];
  __shared__ uint32_t stage_v[kTile];
  uint32_t n = min(*n_ptr, capacity);
  uint32_t nblk = (n + kTile - 1) / kTile;
  uint32_t blk = blockIdx.x;
  if (blk >= nblk) return;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) wave_hist[w][threadIdx.x] = 0;
  __syncthreads();

  // phase A: rank inside the wave's slice (rounds of 64 consecutive elements)
  uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
  const uint32_t wbase = blk * kTile + wave * kWaveTile;
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    uint32_t idx = wbase + r * 64 + lane;
    bool valid = idx < n;
    key[r] = valid ? keys_in[idx] : 0xffffffffu;
    val[r] = valid ? vals_in[idx] : 0u;
  }
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    uint32_t idx = wbase + r * 64 + lane;
    bool valid = idx < n;
    unsigned d = digit_of(key[r], shift, mask);
    unsigned long long peers = ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
      unsigned long long m = ballot((d >> b) & 1u);
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

  // phase B: thread t owns digit t
  {
    uint32_t c[kWaves], block_cnt = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      c[w] = wave_hist[w][threadIdx.x];
      block_cnt += c[w];
    }
    uint32_t tot;
    uint32_t digit_base = block_exclusive_scan(digit_totals[threadIdx.x], wave_sums, &tot);
    uint32_t local_start = block_exclusive_scan(block_cnt, wave_sums, &tot);
    uint32_t run = local_start;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      wave_hist[w][threadIdx.x] = run;      // local position of this wave's first element of the digit
      run += c[w];
    }
    gdst[threadIdx.x] = digit_base + counts[(size_t)threadIdx.x * nblk_cap + blk] - local_start;
  }
  __syncthreads();

  // phase C: re-order by digit inside LDS
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    uint32_t idx = wbase + r * 64 + lane;
    if (idx < n) {
      uint32_t pos = wave_hist[wave][digit_of(key[r], shift, mask)] + rank[r];
      stage_k[pos] = key[r];
      stage_v[pos] = val[r];
    }
  }
  __syncthreads();

  // phase D: lane-linear stores; consecutive threads hit consecutive addresses inside a run
  const uint32_t n_block = min((uint32_t)kTile, n - blk * kTile);
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    uint32_t p = i * kThreads + threadIdx.x;
    if (p < n_block) {
      uint32_t k = stage_k[p];
      uint32_t dst = gdst[digit_of(k, shift, mask)] + p;
      keys_out[dst] = k;
      vals_out[dst] = stage_v[p];
    }
  }
}

template <int ITEMS>
void launch_pass(const uint32_t* n_dev, uint32_t capacity, int shift, int bits, uint32_t* keys_a,
                 uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t* counts,
                 uint32_t* totals, hipStream_t stream) {
  constexpr int kTile = Cfg<ITEMS>::kTile;
  uint32_t nblk_cap = div_up(capacity, kTile);
  uint32_t mask = (1u << bits) - 1u;
  hipLaunchKernelGGL((radix_hist_kernel<ITEMS>), dim3(nblk_cap), dim3(kThreads), 0, stream, n_dev,
                     capacity, shift, mask, keys_a, counts, nblk_cap);
  hipLaunchKernelGGL(radix_rowscan_kernel, dim3(kRadix), dim3(kThreads), 0, stream, n_dev,
                     capacity, (uint32_t)kTile, counts, totals, nblk_cap);
#define MGS_SCATTER(B)                                                                          \
  hipLaunchKernelGGL((radix_scatter_kernel<ITEMS, B>), dim3(nblk_cap), dim3(kThreads), 0,      \
                     stream, n_dev, capacity, shift, mask, keys_a, vals_a, keys_b, vals_b,      \
                     counts, totals, nblk_cap)
  switch (bits) {
    case 1: MGS_SCATTER(1); break;
    case 2: MGS_SCATTER(2); break;
    case 3: MGS_SCATTER(3); break;
    case 4: MGS_SCATTER(4); break;
    case 5: MGS_SCATTER(5); break;
    case 6: MGS_SCATTER(6); break;
    case 7: MGS_SCATTER(7); break;
    default: MGS_SCATTER(8); break;
  }
#undef MGS_SCATTER
}


// ---- one-sweep variant (measured against hist + rowscan + scatter; scripts/binning_ab.py) ----------
// One histogram kernel counts the digits of EVERY pass at once (global totals only), and each pass is a
// single scatter kernel whose blocks find their per-digit offsets by decoupled look-back over the blocks
// before them (Adinets & Merrill's Onesweep): a block takes a ticket (so every lower ticket is resident),
// ranks its tile, publishes its per-digit counts as AGGREGATE words, sums the words of its predecessors
// back to the first INCLUSIVE one, publishes its own inclusive prefix and scatters.  A status word is
// {2 flag bits | 30 count bits} written by one relaxed agent-scope store, so no payload needs a fence
// (MI355X_MICROARCH.md, "granule").  Every poll loop is bounded.
constexpr uint32_t kFlagAgg = 1u << 30, kFlagInc = 2u << 30, kCountMask = (1u << 30) - 1u;
constexpr int kMaxPasses = 4;

template <int ITEMS>
__global__ __launch_bounds__(kThreads) void onesweep_hist_kernel(
    const uint32_t* __restrict__ n_ptr, uint32_t capacity, int passes, int shift0, int bits0, int bits1,
    int bits2, int bits3, const uint32_t* __restrict__ keys, uint32_t* __restrict__ ghist) {
  constexpr int kTile = Cfg<ITEMS>::kTile;
  __shared__ uint32_t hist[kMaxPasses][kRadix];
  const uint32_t n = min(*n_ptr, capacity);
  const uint32_t nblk = (n + kTile - 1) / kTile;
  if (blockIdx.x >= nblk) return;
  for (int p = 0; p < passes; ++p) hist[p][threadIdx.x] = 0;
  __syncthreads();
  const int bits[kMaxPasses] = {bits0, bits1, bits2, bits3};
  const uint32_t base = blockIdx.x * kTile;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t idx = base + i * kThreads + threadIdx.x;
    if (idx < n) {
      uint32_t k = keys[idx] >> shift0;
      for (int p = 0; p < passes; ++p) {
        atomicAdd(&hist[p][k & ((1u << bits[p]) - 1u)], 1u);
        k >>= bits[p];
      }
    }
  }
  __syncthreads();
  for (int p = 0; p < passes; ++p) {
    const uint32_t c = hist[p][threadIdx.x];
    if (c) atomicAdd(&ghist[p * kRadix + threadIdx.x], c);
  }
}

template <int ITEMS, int BITS>
__global__ __launch_bounds__(kThreads) void onesweep_scatter_kernel(
    const uint32_t* __restrict__ n_ptr, uint32_t capacity, int shift, uint32_t mask,
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
    const uint32_t* __restrict__ ghist /* this pass: [kRadix] */, uint32_t* __restrict__ ticket,
    uint32_t* __restrict__ status /* this pass: [nblk_cap][kRadix] */) {
  constexpr int kTile = Cfg<ITEMS>::kTile, kWaveTile = Cfg<ITEMS>::kWaveTile;
  __shared__ uint32_t wave_hist[kWaves][kRadix];
  __shared__ uint32_t gdst[kRadix];
  __shared__ uint32_t wave_sums[kWaves];
  __shared__ uint32_t stage_k[kTile];
  __shared__ uint32_t stage_v[kTile];
  __shared__ uint32_t s_blk;
  const uint32_t n = min(*n_ptr, capacity);
  const uint32_t nblk = (n + kTile - 1) / kTile;
  if (threadIdx.x == 0) s_blk = atomicAdd(ticket, 1u);      // dynamic block id: every lower id is already running
#pragma unroll
  for (int w = 0; w < kWaves; ++w) wave_hist[w][threadIdx.x] = 0;
  __syncthreads();
  const uint32_t blk = s_blk;
  if (blk >= nblk) return;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  // phase A: rank inside the wave's slice (as radix_scatter_kernel)
  uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
  const uint32_t wbase = blk * kTile + wave * kWaveTile;
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    uint32_t idx = wbase + r * 64 + lane;
    bool valid = idx < n;
    key[r] = valid ? keys_in[idx] : 0xffffffffu;
    val[r] = valid ? vals_in[idx] : 0u;
  }
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    uint32_t idx = wbase + r * 64 + lane;
    bool valid = idx < n;
    unsigned d = digit_of(key[r], shift, mask);
    unsigned long long peers = ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
      unsigned long long m = ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    unsigned below = mask_rank(peers);
    unsigned cnt = __popcll(peers);
    uint32_t prev = 0;
    if (valid && below == 0) prev = atomicAdd(&wave_hist[wave][d], cnt);
    int leader = __ffsll((long long)peers) - 1;
    prev = __shfl(prev, leader < 0 ? 0 : leader);
    rank[r] = prev + below;
  }
  __syncthreads();

  // phase B: thread t owns digit t -- block count, look-back, digit base
  {
    uint32_t c[kWaves], block_cnt = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      c[w] = wave_hist[w][threadIdx.x];
      block_cnt += c[w];
    }
    uint32_t* mine = status + (size_t)blk * kRadix + threadIdx.x;
    __hip_atomic_store(mine, block_cnt | (blk == 0 ? kFlagInc : kFlagAgg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // sum of this digit's counts over the blocks before this one: walk back, eight words in flight
    uint32_t excl = 0;
    int i = (int)blk - 1;
    bool done = i < 0;
    while (!done) {
      uint32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = i - u >= 0 ? __hip_atomic_load(status + (size_t)(i - u) * kRadix + threadIdx.x, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT) : kFlagInc;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (done) break;
        uint32_t w = v[u];
        for (int spin = 0; (w >> 30) == 0u && spin < (1 << 22); ++spin) {      // not published yet (bounded)
          __builtin_amdgcn_s_sleep(1);
          w = __hip_atomic_load(status + (size_t)(i - u) * kRadix + threadIdx.x, __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_AGENT);
        }
        // a predecessor that never published (2^22 polls: the block was not scheduled, which cannot happen
        // for tickets taken in dispatch order) must not turn into a silently wrong prefix: abort the launch,
        // the next synchronisation reports it
        if ((w >> 30) == 0u) __builtin_trap();
        excl += w & kCountMask;
        if ((w >> 30) != 1u) done = true;          // INCLUSIVE (or the front of the array)
      }
      i -= 8;
      if (i < 0) done = true;
    }
    if (blk != 0)
      __hip_atomic_store(mine, (excl + block_cnt) | kFlagInc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t tot;
    uint32_t digit_base = block_exclusive_scan(ghist[threadIdx.x], wave_sums, &tot);
    uint32_t local_start = block_exclusive_scan(block_cnt, wave_sums, &tot);
    uint32_t run = local_start;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      wave_hist[w][threadIdx.x] = run;
      run += c[w];
    }
    gdst[threadIdx.x] = digit_base + excl - local_start;
  }
  __syncthreads();

  // phases C, D: re-order by digit inside LDS, lane-linear stores (as radix_scatter_kernel)
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    uint32_t idx = wbase + r * 64 + lane;
    if (idx < n) {
      uint32_t pos = wave_hist[wave][digit_of(key[r], shift, mask)] + rank[r];
      stage_k[pos] = key[r];
      stage_v[pos] = val[r];
    }
  }
  __syncthreads();
  const uint32_t n_block = min((uint32_t)kTile, n - blk * kTile);
#pragma unroll
  for (int i2 = 0; i2 < ITEMS; ++i2) {
    uint32_t p = i2 * kThreads + threadIdx.x;
    if (p < n_block) {
      uint32_t k = stage_k[p];
      uint32_t dst = gdst[digit_of(k, shift, mask)] + p;
      keys_out[dst] = k;
      vals_out[dst] = stage_v[p];
    }
  }
}

}  // namespace

// passes: key bits split as evenly as possible into digits of at most 8 bits
int radix_sort_passes(int key_bits) { return (key_bits + 7) / 8; }

#ifdef MGS_DEBUG_HOOKS       // libmgs_debug.so only: the shipped library keeps no process-global state
static int g_sort_opts = 0;
extern "C" void mgs_debug_set_sort_opts(int opts) { g_sort_opts = opts; }
#else
static constexpr int g_sort_opts = 0;
#endif
int sort_opts() { return g_sort_opts; }

size_t radix_sort_temp_bytes(uint32_t capacity) {
  size_t nblk_cap = div_up(capacity ? capacity : 1u, Cfg<4>::kTile);   // the smaller tile bounds both
  // three-kernel passes: counts[kRadix][nblk] + totals; one-sweep: ghist + tickets + status[passes][nblk][kRadix]
  size_t three = (kRadix * nblk_cap + kRadix) * sizeof(uint32_t);
  size_t sweep = ((size_t)kMaxPasses * kRadix + 64 + (size_t)kMaxPasses * nblk_cap * kRadix) * sizeof(uint32_t);
  return align_up(three > sweep ? three : sweep, 256);
}

// Result lands in (keys_b, vals_b) when the pass count ceil(key_bits/8) is odd, else in
// (keys_a, vals_a).
int radix_sort_pairs(const uint32_t* n_dev, uint32_t capacity, int key_bits, uint32_t* keys_a,
                     uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, void* temp,
                     hipStream_t stream) {
  if (capacity == 0) return MGS_OK;
  if ((g_sort_opts & 1) && (capacity > kSmallLimit || (g_sort_opts & 2)) && radix_sort_passes(key_bits) <= kMaxPasses) {
    // one-sweep variant (large inputs; bit 1 of the knob forces it for small ones too: tests)
    constexpr int IT = 16;
    const uint32_t nblk_cap = div_up(capacity, Cfg<IT>::kTile);
    const int passes = radix_sort_passes(key_bits);
    uint32_t* ghist = static_cast<uint32_t*>(temp);
    uint32_t* tickets = ghist + kMaxPasses * kRadix;
    uint32_t* status = tickets + 64;
    const size_t zero_bytes = ((size_t)kMaxPasses * kRadix + 64 + (size_t)passes * nblk_cap * kRadix) * sizeof(uint32_t);
    hipError_t e = hipMemsetAsync(temp, 0, zero_bytes, stream);
    if (e != hipSuccess) return set_error((int)e, "radix_sort_pairs: memset: %s", hipGetErrorString(e));
    int bits[kMaxPasses] = {0, 0, 0, 0}, sh = 0;
    for (int p = 0; p < passes; ++p) { bits[p] = (key_bits - sh + (passes - p) - 1) / (passes - p); sh += bits[p]; }
    hipLaunchKernelGGL((onesweep_hist_kernel<IT>), dim3(nblk_cap), dim3(kThreads), 0, stream, n_dev, capacity,
                       passes, 0, bits[0], bits[1], bits[2], bits[3], keys_a, ghist);
    int shift = 0;
    for (int p = 0; p < passes; ++p) {
      const uint32_t mask = (1u << bits[p]) - 1u;
#define MGS_SWEEP(B)                                                                                       \
      hipLaunchKernelGGL((onesweep_scatter_kernel<IT, B>), dim3(nblk_cap), dim3(kThreads), 0, stream, n_dev, \
                         capacity, shift, mask, keys_a, vals_a, keys_b, vals_b, ghist + p * kRadix,          \
                         tickets + p, status + (size_t)p * nblk_cap * kRadix)
      switch (bits[p]) {
        case 1: MGS_SWEEP(1); break; case 2: MGS_SWEEP(2); break; case 3: MGS_SWEEP(3); break;
        case 4: MGS_SWEEP(4); break; case 5: MGS_SWEEP(5); break; case 6: MGS_SWEEP(6); break;
        case 7: MGS_SWEEP(7); break; default: MGS_SWEEP(8); break;
      }
#undef MGS_SWEEP
      shift += bits[p];
      uint32_t* t;
      t = keys_a; keys_a = keys_b; keys_b = t;
      t = vals_a; vals_a = vals_b; vals_b = t;
    }
    return check_launch("radix_sort_pairs(one-sweep)");
  }
  const bool small = capacity <= kSmallLimit;
  uint32_t nblk_cap = div_up(capacity, small ? Cfg<4>::kTile : Cfg<16>::kTile);
  uint32_t* counts = static_cast<uint32_t*>(temp);
  uint32_t* totals = counts + (size_t)kRadix * nblk_cap;
  const int passes = radix_sort_passes(key_bits);
  int shift = 0;
  for (int p = 0; p < passes; ++p) {
    int bits = (key_bits - shift + (passes - p) - 1) / (passes - p);   // even split, <= 8
    if (small)
      launch_pass<4>(n_dev, capacity, shift, bits, keys_a, vals_a, keys_b, vals_b, counts, totals, stream);
    else
      launch_pass<16>(n_dev, capacity, shift, bits, keys_a, vals_a, keys_b, vals_b, counts, totals, stream);
    shift += bits;
    uint32_t* t;
    t = keys_a; keys_a = keys_b; keys_b = t;
    t = vals_a; vals_a = vals_b; vals_b = t;
  }
  return check_launch("radix_sort_pairs");
}

}  // namespace mgs
