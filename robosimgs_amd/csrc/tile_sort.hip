// tile_sort.hip -- depth order INSIDE each tile's list, one workgroup per tile (gfx950).
//
// The binning partitions the (tile, Gaussian) pairs by tile (radix path) or by group of 2^shift tiles (direct
// path, binning.hip), in no particular order inside a list.  This kernel then sorts each
// list by (depth bits, Gaussian index) -- exactly the order of the textbook single sort on
// (tile << 32 | depth) with index-order ties -- so the 1 M-key global
// depth sort (12 dependent launches) and the rank gather of the first version disappear: one
// fully parallel launch, no cross-workgroup dependency, ~8160 independent segments of ~450 entries.
//
// Per segment: most-significant-digit bucketing on the 64-bit composite key (depth << 32 | id):
//   min / max of the composites -> highest differing bit hb -> 10-bit digit (comp >> (hb - 9)) & 1023
//   LDS histogram (atomics), exclusive scan, scatter into the other scratch buffer (bucket order,
//   arbitrary order inside a bucket), then every element of a bucket of <= 48 entries finds its final
//   place by counting the smaller composites of its own bucket.  Depth keys of one tile share their
//   exponent bits, so the digit lands on the top mantissa bits and buckets hold ~1 element: O(n).
//   A bucket of more than 48 entries (many near-identical depths) goes onto an LDS stack and is
//   bucketed again on ITS highest differing bit (ten bits further down at least), so any input
//   terminates in at most seven levels; composites are distinct (ids are), ties cannot loop.
// Lists of up to kFast entries (nearly all of them) take the fast path: ids and gathered depth keys
// stay in registers through the histogram and the scatter, the scattered list sits in LDS, and the
// only global traffic is the id load, the depth gather and the final store.  Longer lists, and the
// heavy buckets of any list, run the generic loop whose elements live in two global scratch buffers
// (L2-resident for the workgroup) -- only the 1024 counters are in LDS, so any length is handled.
#include <type_traits>

#include "mgs_common.h"

namespace mgs {
namespace {

constexpr int kTSMain = 256;        // threads per tile
// Lists longer than the main kernel's LDS list (kFast entries) are not walked by its 256 threads through global scratch
// -- 540 us for ONE tile of 31 k entries on a clustered scene, the tail of the whole binning stage -- but handed to a
// second launch of a few large workgroups: 1024 threads, an LDS list of kFastXL entries, the same algorithm.  The main
// kernel appends such tiles to a list.  The second launch costs 4.6 us even when the list is empty, so it exists only where
// the capacity says lists are long (tile_depth_sort below).
constexpr int kTSLong = 1024, kFastXL = 8192, kLongGrid = 256;
#ifndef MGS_TSORT_LONG_BUCKETS
#define MGS_TSORT_LONG_BUCKETS 2048    // buckets per level in the long lists' kernel (the main kernel: 1024); clustered scene 206 / 186 / 203 us at 1024 / 2048 / 4096
#endif
// buckets of one MSD level
constexpr int buckets_for(int fast) { return fast > 2048 ? MGS_TSORT_LONG_BUCKETS : 1024; }   // (256 for the short-list variant: fewer counters to zero / scan, but ~2 entries per
                                                  //  bucket make the wave's rank loop as long as its fullest bucket: 22.6 M VALU against 21.8 M)
constexpr int log2i(int v) { return v <= 1 ? 0 : 1 + log2i(v >> 1); }
#ifndef MGS_TSORT_SMALL
// buckets up to this size are finished by rank counting out of LDS; larger ones are bucketed again, one after the other, through
// global scratch.  48 until round 5; on a clustered scene most tiles of 1-2 k entries hold a few dozen buckets of 50-150
// near-identical depths: at 128 the main kernel takes 75 instead of 104 us there (256: 90), on SURVEY 8(d)'s scene 33.0 either way
#define MGS_TSORT_SMALL 128
#endif
constexpr int kSmall = MGS_TSORT_SMALL;
#ifndef MGS_TSORT_SMALL_LONG
#define MGS_TSORT_SMALL_LONG 256
#endif
constexpr int kSmallLong = MGS_TSORT_SMALL_LONG;   // ... in the kernel of the long lists (sort_one_tile)
constexpr int kStack = 96;          // pending heavy buckets; beyond it a bucket is rank-counted whatever its size
#ifndef MGS_TSORT_STOP
#define MGS_TSORT_STOP 0      // measurement only: leave the kernel after phase 1..4 (filter / keys / scan / scatter)
#endif
#ifndef MGS_TSORT_LDS_LEVELS
#define MGS_TSORT_LDS_LEVELS 1     // popped heavy buckets that fit the LDS list take a level out of LDS (sort_one_tile)
#endif
#ifndef MGS_TSORT_FAST
#define MGS_TSORT_FAST 2048
#endif
constexpr int kFastLong = MGS_TSORT_FAST;   // longest list of the LDS-resident fast path (8 KiB of LDS per 1024) ...
constexpr int kFastShort = 1024;            // ... and where the capacity says lists are short on average: 17 instead of
                                            // 25 KiB of LDS per workgroup = 8 instead of 6 workgroups per CU
constexpr int kFastMid = 1536;              // ... and in between (up to 1,000 entries per tile on average: configs[4] has 806,
                                            // its longest list 1,479): 21 KiB, 7 workgroups per CU -- the stage 428 -> 414 us at
                                            // 4K, 1,024 -> 1,049 frames/s with three in flight (1,280 entries: 443 us, the
                                            // long-list launch gets work); lists over it go to the second launch as ever
constexpr uint32_t kBrute = 0x80000000u;

__device__ __forceinline__ bool comp_less(uint32_t ka, uint32_t ia, uint32_t kb, uint32_t ib) {
  return ka < kb || (ka == kb && ia < ib);
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int d) {
  unsigned lo = __shfl_xor((unsigned)v, d), hi = __shfl_xor((unsigned)(v >> 32), d);
  return ((unsigned long long)hi << 32) | lo;
}

// exclusive scan of one value per thread over the workgroup; *total = sum
template <int kTS>
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* wave_sums, uint32_t* total) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // inclusive scan over the wave in six DPP adds: row_shr 1 / 2 / 4 / 8 inside the rows of 16 (zero fill), then lane 15
  // of rows 0 / 2 onto rows 1 / 3 and lane 31 onto rows 2 and 3
  uint32_t incl = v;
#define MGS_SCAN_STEP(CTRL, RMASK, BOUND) \
  incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, CTRL, RMASK, 0xf, BOUND);
  MGS_SCAN_STEP(0x111, 0xf, true)
  MGS_SCAN_STEP(0x112, 0xf, true)
  MGS_SCAN_STEP(0x114, 0xf, true)
  MGS_SCAN_STEP(0x118, 0xf, true)
  MGS_SCAN_STEP(0x142, 0xa, false)
  MGS_SCAN_STEP(0x143, 0xc, false)
#undef MGS_SCAN_STEP
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  uint32_t off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kTS / 64; ++w) {
    uint32_t s = wave_sums[w];
    if ((unsigned)w < wave) off += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return off + incl - v;
}

// GROUPED: the binning's direct path has dropped the pairs of every 2^shift consecutive tiles into one segment
// of `staging` (entry = id | tile's place in the group << (32 - shift), any order).  The tile's workgroup reads
// its group's segment (the 2^shift workgroups of a group run side by side: L2 hits), keeps its own entries and
// counts those of the group's earlier tiles -- which is where its list starts; it stores that offset too.
// LONG: the second launch (lists over the main kernel's kFast).  DEFER (main kernel): such a tile is appended to long_list and
// left to that launch; without it the main kernel walks the list itself through global scratch (the generic path below).
template <bool GROUPED, int kFast, int kTS, bool LONG, bool DEFER = true>
__device__ __forceinline__ void sort_one_tile(
    const int tile, int n_tiles, const int32_t* __restrict__ offsets, const float* __restrict__ depths,
    uint32_t* ids_final, uint32_t* __restrict__ tile_ids, uint32_t* key0, uint32_t* id0, uint32_t* key1,
    uint32_t* id1, const uint32_t* __restrict__ staging, int shift, int32_t* __restrict__ offsets_out,
    uint32_t* __restrict__ long_list) {
  constexpr int kBuckets = buckets_for(kFast), kDigitBits = log2i(kBuckets);
  // lds_level's packed scan carries the running entry count in 16 bits and the heavy-bucket count above it
  static_assert(kFast < 65536 && kBuckets < 65536, "the packed bucket scan holds counts below 2^16");
  static_assert(kBuckets % kTS == 0 && kFast % kTS == 0, "buckets and list entries are dealt evenly to the threads");
  // buckets up to kSm entries are finished by rank counting.  The long lists' kernel takes far larger ones: its candidates
  // come out of LDS (the fast path's list, the generic path's windows), and every bucket it does NOT rank is one more
  // level taken by the whole 1024-thread workgroup, one bucket after the other -- a clustered scene's 31 k-entry list
  // had dozens of buckets of 50-500 near-identical depths
  constexpr int kSm = LONG ? kSmallLong : kSmall;
  __shared__ uint32_t cnt[kBuckets];
  __shared__ uint32_t cur[kBuckets];
  __shared__ unsigned long long red_min[kTS / 64], red_max[kTS / 64];
  __shared__ uint32_t red_or[kTS / 64], red_and[kTS / 64];
  __shared__ uint32_t wave_sums[kTS / 64];
  __shared__ int stack_lo[kStack], stack_hi[kStack];
  __shared__ uint8_t stack_src[kStack];
  __shared__ int stack_n;
  constexpr int kItems = kFast / kTS;
  // the tile's list in LDS: first its ids (li, written by the group filter), later -- the ids are in registers by then,
  // two barriers earlier -- the scattered composites {id, depth bits} as one 64-bit word each (lc), so that the rank loop
  // reads one ds_read_b64 and makes one 64-bit compare per candidate
  __shared__ unsigned long long lc[kFast];
  uint32_t* li = reinterpret_cast<uint32_t*>(lc);
  const int tid = threadIdx.x;
  int s, e, gs = 0, ge = 0;
  uint32_t local = 0, id_mask = ~0u;
  __shared__ uint32_t gcount[2];           // GROUPED: entries of earlier tiles of the group / of this tile
  if (GROUPED) {
    const int grp = tile >> shift;
    local = (uint32_t)tile & ((1u << shift) - 1u);
    id_mask = (1u << (32 - shift)) - 1u;
    gs = offsets[grp]; ge = offsets[grp + 1];          // `offsets` are the groups' here
    if (tid < 2) gcount[tid] = 0u;
    __syncthreads();
    uint32_t below = 0;
    constexpr int kInFlight = 8;                        // loads in flight per thread (the trips are one latency chain)
    for (int i0 = gs; i0 < ge; i0 += kTS * kInFlight) { // uniform trips: the wave ranks its matches with one ballot
      uint32_t vv[kInFlight];
#pragma unroll
      for (int j = 0; j < kInFlight; ++j) {
        const int i = i0 + j * kTS + tid;
        vv[j] = i < ge ? staging[i] : 0u;
      }
      // the wave's matches of the whole trip take ONE LDS atomic: per item a ballot and a running count
      unsigned long long mm[kInFlight];
      bool mine[kInFlight];
      uint32_t run[kInFlight], wave_total = 0;
#pragma unroll
      for (int j = 0; j < kInFlight; ++j) {
        const int i = i0 + j * kTS + tid;
        const uint32_t l = vv[j] >> (32 - shift);
        const bool in = i < ge;                          // (items past the segment: vv = 0 and in = false)
        below += (uint32_t)__popcll(ballot(in && l < local));      // wave-uniform: scalar popcounts, no lane sums
        mine[j] = in && l == local;
        mm[j] = ballot(mine[j]);
        run[j] = wave_total;
        wave_total += (uint32_t)__popcll(mm[j]);
      }
      uint32_t base = 0;
      if ((tid & 63) == 0 && wave_total) base = atomicAdd(&gcount[1], wave_total);
      base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
      for (int j = 0; j < kInFlight; ++j) {
        if (mine[j]) {
          const uint32_t pos = base + run[j] + mask_rank(mm[j]);
          if (pos < (uint32_t)kFast) li[pos] = vv[j] & id_mask;
        }
      }
    }
    if ((tid & 63) == 0 && below) atomicAdd(&gcount[0], below);
    __syncthreads();
    s = gs + (int)gcount[0];
    e = s + (int)gcount[1];
    if (tid == 0 && !LONG) {
      offsets_out[tile] = s;
      if (tile == n_tiles - 1) offsets_out[n_tiles] = e;
    }
  } else {
    s = offsets[tile]; e = offsets[tile + 1];
  }
  if (!LONG && DEFER && e - s > kFast) {      // (uniform) a long list: the second launch sorts it (its tile ids are filled here)
    if (tile_ids)
      for (int i = s + tid; i < e; i += kTS) tile_ids[i] = (uint32_t)tile;
    if (tid == 0) long_list[1 + atomicAdd(long_list, 1u)] = (uint32_t)tile;
    return;
  }
#if MGS_TSORT_STOP == 1
  return;
#endif
  if (tile_ids && !LONG)
    for (int i = s + tid; i < e; i += kTS) tile_ids[i] = (uint32_t)tile;
  if (e - s <= 1) {
    if (GROUPED && e - s == 1 && tid == 0) ids_final[s] = li[0];
    return;
  }
  const int n_list = e - s;

  // ---- one bucketing level out of registers + LDS: the n <= kFast composites of [s, s + n) -----------------------
  // FIRST: the tile's whole list (ids out of the group filter's LDS list or flatten_ids, depths gathered); otherwise a
  // heavy bucket the generic loop popped (keys and ids out of the scratch buffer it lies in: one coalesced load instead
  // of the four passes over global memory that a level of the generic path is -- round 5: a clustered scene's lists hold
  // dozens of buckets of a few hundred near-identical depths each).  Light buckets are ranked out of LDS and stored;
  // heavy ones go to buffer (hk, hi) with stack slots from stack_base on.  Returns the number of heavy buckets.
  auto lds_level = [&](auto first_tag, const int s, const int n, const uint32_t* sk, const uint32_t* si, uint32_t* hk, uint32_t* hi,
                       const uint8_t hsrc, const int stack_base) -> uint32_t {
    constexpr bool FIRST = decltype(first_tag)::value;
    uint32_t rk[kItems], ri[kItems];
    // (every item loop below stops, wave-uniformly, at the first item no thread of the workgroup owns:
    //  a 455-entry list executes two of the eight unrolled trips)
    // (no per-lane branches around the loads: the id of a slot past the end is the list's last one, masked out
    //  below -- a branch per slot made the compiler wait for every gather before issuing the next)
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      ri[it] = 0u;
      if (it * kTS >= n) continue;
      const int idx = min(it * kTS + tid, n - 1);
      if constexpr (FIRST) ri[it] = GROUPED ? li[idx] : ids_final[s + idx];
      else ri[it] = si[s + idx];
    }
    // highest bit in which two composites (depth bits << 32 | id) differ: the bits where the depth keys are not
    // all alike are OR & ~AND over the list (32-bit reductions); only a list of identical depths looks at the ids
    uint32_t kor = 0u, kand = ~0u;
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      rk[it] = 0u;
      if (it * kTS >= n) continue;
      if constexpr (FIRST) rk[it] = __float_as_uint(depths[ri[it]]);
      else rk[it] = sk[s + min(it * kTS + tid, n - 1)];
    }
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      if (it * kTS >= n) continue;
      const bool in = it * kTS + tid < n;
      kor |= in ? rk[it] : 0u;
      kand &= in ? rk[it] : ~0u;
    }
    kor = wave_or(kor);
    kand = wave_and(kand);
    if ((tid & 63) == 0) { red_or[tid >> 6] = kor; red_and[tid >> 6] = kand; }
#pragma unroll
    for (int k = 0; k < kBuckets / kTS; ++k) cnt[tid + k * kTS] = 0;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kTS / 64; ++w) { kor |= red_or[w]; kand &= red_and[w]; }
    int hb;
    if (kor != kand) {                                      // uniform
      hb = 32 + (31 - __clz((int)(kor ^ kand)));
    } else {                                                // all depths identical: the ids decide (distinct: n >= 2)
      uint32_t ior = 0u, iand = ~0u;
#pragma unroll
      for (int it = 0; it < kItems; ++it)
        if (it * kTS + tid < n) { ior |= ri[it]; iand &= ri[it]; }
      ior = wave_or(ior);
      iand = wave_and(iand);
      __syncthreads();
      if ((tid & 63) == 0) { red_or[tid >> 6] = ior; red_and[tid >> 6] = iand; }
      __syncthreads();
#pragma unroll
      for (int w = 0; w < kTS / 64; ++w) { ior |= red_or[w]; iand &= red_and[w]; }
      hb = 31 - __clz((int)(ior ^ iand));
    }
#if MGS_TSORT_STOP == 2
    if (hb >= 0) { if (tid == 0) ids_final[s] = (uint32_t)hb; return 0u; }
#endif
    const int shift = hb > kDigitBits - 1 ? hb - (kDigitBits - 1) : 0;
    // shift >= 32 (the depths differ above their lowest kDigitBits - 1 bits: nearly always): the digit is a bit
    // field of the 32-bit key
    const bool key_digit = shift >= 32;
    const int kshift = shift - 32;
    auto digit = [&](uint32_t k, uint32_t id) -> unsigned {
      if (key_digit) return (k >> kshift) & (kBuckets - 1);
      const unsigned long long c = ((unsigned long long)k << 32) | id;
      return (unsigned)(c >> shift) & (kBuckets - 1);
    };
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      if (it * kTS >= n) continue;
      if (it * kTS + tid < n) atomicAdd(&cnt[digit(rk[it], ri[it])], 1u);
    }
    __syncthreads();
    uint32_t htot;
    {
      // one scan for both: bucket sizes in the low 16 bits (n <= 2048), heavy-bucket count above
      uint32_t c4[kBuckets / kTS], packed = 0;
#pragma unroll
      for (int k = 0; k < kBuckets / kTS; ++k) {
        c4[k] = cnt[tid * (kBuckets / kTS) + k];
        packed += c4[k] + (c4[k] > (uint32_t)kSm ? 0x10000u : 0u);
      }
      uint32_t tot;
      uint32_t exp = block_scan_excl<kTS>(packed, wave_sums, &tot);
      uint32_t ex = exp & 0xffffu, hx = exp >> 16;
      htot = tot >> 16;
#pragma unroll
      for (int k = 0; k < kBuckets / kTS; ++k) {
        const int d = tid * (kBuckets / kTS) + k;
        cur[d] = ex;
        if (c4[k] > (uint32_t)kSm) {            // the generic loop below takes it from buffer (hk, hi)
          const int slot = stack_base + (int)hx;
          if (slot < kStack) {
            stack_lo[slot] = s + (int)ex; stack_hi[slot] = s + (int)(ex + c4[k]); stack_src[slot] = hsrc;
          } else {
            cnt[d] = c4[k] | kBrute;
          }
          ++hx;
        }
        ex += c4[k];
      }
      if (tid == 0) stack_n = stack_base + (int)htot < kStack ? stack_base + (int)htot : kStack;
    }
    __syncthreads();
#if MGS_TSORT_STOP == 3
    if (n > 0) { if (tid == 0) ids_final[s] = cur[3]; return 0u; }
#endif
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      if (it * kTS >= n) continue;
      if (it * kTS + tid < n) {
        const uint32_t p = atomicAdd(&cur[digit(rk[it], ri[it])], 1u);
        lc[p] = ((unsigned long long)rk[it] << 32) | ri[it];
      }
    }
    __syncthreads();
#if MGS_TSORT_STOP == 4
    if (n > 0) { if (tid == 0) ids_final[s] = (uint32_t)lc[3]; return 0u; }
#endif
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      const int i = it * kTS + tid;
      if (it * kTS >= n) continue;
      if (i < n) {
        const unsigned long long me = lc[i];
        const uint32_t k = (uint32_t)(me >> 32), id = (uint32_t)me;
        const unsigned d = digit(k, id);
        const uint32_t craw = cnt[d];
        const uint32_t b = craw & ~kBrute;
        if (b > (uint32_t)kSm && !(craw & kBrute)) {   // heavy: hand it to the generic loop
          hk[s + i] = k;
          hi[s + i] = id;
        } else {
          const int be = (int)cur[d], bs = be - (int)b;
          int c = 0;
          for (int j = bs; j < be; ++j) c += lc[j] < me ? 1 : 0;
          ids_final[s + bs + c] = id;
        }
      }
    }
    return htot;
  };

  const int n = n_list;
  if (n <= kFast) {
    // ---- fast path: the whole list is one level out of LDS ---------------------------------------
    const uint32_t htot = lds_level(std::true_type{}, s, n, (const uint32_t*)nullptr, (const uint32_t*)nullptr, key1, id1, (uint8_t)1, 0);
    if (htot == 0) return;                          // uniform
    __syncthreads();
  } else {
    // level 0 input: the tile's ids; keys are gathered once and parked in buffer 0
    if (GROUPED) {          // longer than the LDS list: collect again, straight into buffer 0
      if (tid == 0) gcount[1] = 0u;
      __syncthreads();
      // (eight segment loads, then the matches' depth gathers, in flight together: one trip was two exposed round trips)
      constexpr int kIF = 8;
      for (int i0 = gs; i0 < ge; i0 += kTS * kIF) {
        uint32_t vv[kIF], pos[kIF], kk[kIF];
        bool mine[kIF];
#pragma unroll
        for (int j = 0; j < kIF; ++j) {
          const int i = i0 + j * kTS + tid;
          vv[j] = i < ge ? staging[i] : 0u;
        }
        unsigned long long mm[kIF];
        uint32_t run[kIF], wave_total = 0;
#pragma unroll
        for (int j = 0; j < kIF; ++j) {
          const int i = i0 + j * kTS + tid;
          mine[j] = i < ge && (vv[j] >> (32 - shift)) == local;
          mm[j] = ballot(mine[j]);
          run[j] = wave_total;
          wave_total += (uint32_t)__popcll(mm[j]);
        }
        uint32_t base = 0;
        if ((tid & 63) == 0 && wave_total) base = atomicAdd(&gcount[1], wave_total);
        base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
        for (int j = 0; j < kIF; ++j) {
          pos[j] = base + run[j] + mask_rank(mm[j]);
          kk[j] = mine[j] ? __float_as_uint(depths[vv[j] & id_mask]) : 0u;
        }
#pragma unroll
        for (int j = 0; j < kIF; ++j)
          if (mine[j]) {
            key0[s + pos[j]] = kk[j];
            id0[s + pos[j]] = vv[j] & id_mask;
          }
      }
    } else {
      constexpr int kIF = 4;
      for (int i0 = s; i0 < e; i0 += kTS * kIF) {
        uint32_t id[kIF], kk[kIF];
#pragma unroll
        for (int j = 0; j < kIF; ++j) { const int i = i0 + j * kTS + tid; id[j] = i < e ? ids_final[i] : 0u; }
#pragma unroll
        for (int j = 0; j < kIF; ++j) { const int i = i0 + j * kTS + tid; kk[j] = i < e ? __float_as_uint(depths[id[j]]) : 0u; }
#pragma unroll
        for (int j = 0; j < kIF; ++j) {
          const int i = i0 + j * kTS + tid;
          if (i < e) { key0[i] = kk[j]; id0[i] = id[j]; }
        }
      }
    }
    if (tid == 0) {
      stack_n = 1;
      stack_lo[0] = s; stack_hi[0] = e; stack_src[0] = 0;
    }
    __syncthreads();
  }

  // (measured and rejected, round 5: pending buckets of up to kFast / waves entries finished by ONE wave each, all waves at
  //  once, by rank counting out of a slice of the LDS list -- 285 instead of 252 us for the long lists of the clustered
  //  scene, 189 instead of 102 us for the main kernel: m^2 / 64 candidates per lane lose to one more bucketing level)
  while (true) {
    const int sn = stack_n;                        // uniform: read behind a barrier
    if (sn == 0) break;
    const int lo = stack_lo[sn - 1], hi = stack_hi[sn - 1];
    const int src = stack_src[sn - 1];
    __syncthreads();
    if (tid == 0) stack_n = sn - 1;
    const uint32_t* sk = src ? key1 : key0;
    const uint32_t* si = src ? id1 : id0;
    uint32_t* dk = src ? key0 : key1;
    uint32_t* di = src ? id0 : id1;
    const int m = hi - lo;

    // (the long lists' kernel only: a second copy of the level takes the main kernel from 47-53 to 64-86 VGPRs, a
    //  workgroup per CU less for every scene, and its heavy buckets are few)
    if constexpr (LONG && MGS_TSORT_LDS_LEVELS != 0) {
      if (m > kSm && m <= kFast) {              // fits the LDS list: one level out of LDS (heavy sub-buckets to the other buffer)
        (void)lds_level(std::false_type{}, lo, m, sk, si, dk, di, (uint8_t)(src ^ 1), sn - 1);
        __syncthreads();
        continue;
      }
    }
    if (m <= kSm) {                             // whole segment by rank counting
      for (int i = lo + tid; i < hi; i += kTS) {
        const uint32_t k = sk[i], id = si[i];
        int c = 0;
        for (int j = lo; j < hi; ++j) c += comp_less(sk[j], si[j], k, id) ? 1 : 0;
        ids_final[lo + c] = id;
      }
      __syncthreads();
      continue;
    }

    // highest differing bit of the composites
    constexpr int kIF = 4;                         // loads in flight per thread in the passes below
    unsigned long long mn = ~0ull, mx = 0ull;
    for (int i0 = lo; i0 < hi; i0 += kTS * kIF) {
      uint32_t kk[kIF], ii[kIF];
#pragma unroll
      for (int j = 0; j < kIF; ++j) {
        const int i = i0 + j * kTS + tid;
        kk[j] = i < hi ? sk[i] : 0u;
        ii[j] = i < hi ? si[i] : 0u;
      }
#pragma unroll
      for (int j = 0; j < kIF; ++j)
        if (i0 + j * kTS + tid < hi) {
          const unsigned long long c = ((unsigned long long)kk[j] << 32) | ii[j];
          mn = c < mn ? c : mn;
          mx = c > mx ? c : mx;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const unsigned long long a = shfl_xor_u64(mn, d), b = shfl_xor_u64(mx, d);
      mn = a < mn ? a : mn;
      mx = b > mx ? b : mx;
    }
    if ((tid & 63) == 0) { red_min[tid >> 6] = mn; red_max[tid >> 6] = mx; }
#pragma unroll
    for (int k = 0; k < kBuckets / kTS; ++k) cnt[tid + k * kTS] = 0;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kTS / 64; ++w) {
      mn = red_min[w] < mn ? red_min[w] : mn;
      mx = red_max[w] > mx ? red_max[w] : mx;
    }
    const int hb = 63 - __clzll((long long)(mn ^ mx));       // composites are distinct: mn != mx
    const int shift = hb > kDigitBits - 1 ? hb - (kDigitBits - 1) : 0;
    auto digit = [&](uint32_t k, uint32_t id) -> unsigned {
      const unsigned long long c = ((unsigned long long)k << 32) | id;
      return (unsigned)(c >> shift) & (kBuckets - 1);
    };

    for (int i0 = lo; i0 < hi; i0 += kTS * kIF) {
      uint32_t kk[kIF], ii[kIF];
#pragma unroll
      for (int j = 0; j < kIF; ++j) {
        const int i = i0 + j * kTS + tid;
        kk[j] = i < hi ? sk[i] : 0u;
        ii[j] = i < hi ? si[i] : 0u;
      }
#pragma unroll
      for (int j = 0; j < kIF; ++j)
        if (i0 + j * kTS + tid < hi) atomicAdd(&cnt[digit(kk[j], ii[j])], 1u);
    }
    __syncthreads();

    // exclusive scan of the bucket sizes (four consecutive buckets per thread); heavy buckets get their
    // stack slots here, deterministically -- past the stack's room they are finished by rank counting
    {
      uint32_t c4[kBuckets / kTS], sum = 0, heavy = 0;
#pragma unroll
      for (int k = 0; k < kBuckets / kTS; ++k) {
        c4[k] = cnt[tid * (kBuckets / kTS) + k];
        sum += c4[k];
        heavy += c4[k] > (uint32_t)kSm ? 1u : 0u;
      }
      uint32_t tot, htot;
      uint32_t ex = block_scan_excl<kTS>(sum, wave_sums, &tot);
      uint32_t hx = block_scan_excl<kTS>(heavy, wave_sums, &htot);
      const int base = sn - 1;                     // stack height after the pop
#pragma unroll
      for (int k = 0; k < kBuckets / kTS; ++k) {
        const int d = tid * (kBuckets / kTS) + k;
        cur[d] = ex;
        if (c4[k] > (uint32_t)kSm) {
          const int slot = base + (int)hx;
          if (slot < kStack) {
            stack_lo[slot] = lo + (int)ex; stack_hi[slot] = lo + (int)(ex + c4[k]); stack_src[slot] = (uint8_t)(src ^ 1);
          } else {
            cnt[d] = c4[k] | kBrute;
          }
          ++hx;
        }
        ex += c4[k];
      }
      if (tid == 0) stack_n = base + (int)htot < kStack ? base + (int)htot : kStack;
    }
    __syncthreads();

    for (int i0 = lo; i0 < hi; i0 += kTS * kIF) {
      uint32_t kk[kIF], ii[kIF];
#pragma unroll
      for (int j = 0; j < kIF; ++j) {
        const int i = i0 + j * kTS + tid;
        kk[j] = i < hi ? sk[i] : 0u;
        ii[j] = i < hi ? si[i] : 0u;
      }
#pragma unroll
      for (int j = 0; j < kIF; ++j)
        if (i0 + j * kTS + tid < hi) {
          const uint32_t p = atomicAdd(&cur[digit(kk[j], ii[j])], 1u);
          dk[lo + p] = kk[j];
          di[lo + p] = ii[j];
        }
    }
    __syncthreads();                               // the scattered segment is visible to the workgroup

    // cur[d] is now the END of bucket d: every element of a light bucket ranks itself inside it.  The scattered
    // segment goes through the LDS list in windows of kFast - 2 kSm entries with kSm more on either side (a light
    // bucket holds at most kSm entries, so the whole bucket of every element of the window is in LDS): one coalesced
    // load per entry instead of a dependent global load per CANDIDATE -- a 31 k-entry list has ~30 of them per element.
    constexpr int kWin = kFast - 2 * kSm;
    for (int w0 = lo; w0 < hi; w0 += kWin) {
      const int wa = max(lo, w0 - kSm), wb = min(hi, w0 + kWin + kSm), we = min(hi, w0 + kWin);
      for (int i = wa + tid; i < wb; i += kTS) lc[i - wa] = ((unsigned long long)dk[i] << 32) | di[i];
      __syncthreads();
      for (int i = w0 + tid; i < we; i += kTS) {
        const unsigned long long me = lc[i - wa];
        const uint32_t k = (uint32_t)(me >> 32), id = (uint32_t)me;
        const unsigned d = digit(k, id);
        const uint32_t craw = cnt[d];
        const uint32_t b = craw & ~kBrute;
        if (b > (uint32_t)kSm && !(craw & kBrute)) continue;     // on the stack
        const int be = lo + (int)cur[d], bs = be - (int)b;
        int c = 0;
        if (b <= (uint32_t)kSm) {
          for (int j = bs; j < be; ++j) c += lc[j - wa] < me ? 1 : 0;
        } else {                                                    // past the stack's room: any size, out of global memory
          for (int j = bs; j < be; ++j) c += comp_less(dk[j], di[j], k, id) ? 1 : 0;
        }
        ids_final[bs + c] = id;
      }
      __syncthreads();                             // the next window overwrites the list
    }
  }
}

template <bool GROUPED, int kFast, bool DEFER>
__global__ __launch_bounds__(kTSMain) void tile_depth_sort_kernel(
    int n_tiles, const int32_t* __restrict__ offsets, const float* __restrict__ depths,
    uint32_t* ids_final, uint32_t* __restrict__ tile_ids, uint32_t* key0, uint32_t* id0, uint32_t* key1,
    uint32_t* id1, const uint32_t* __restrict__ staging, int shift, int32_t* __restrict__ offsets_out,
    uint32_t* __restrict__ long_list) {
  // GROUPED: the 2^shift workgroups of a group all read the group's segment.  Workgroup b runs on XCD b % 8 (observed
  // placement, used for speed only), each XCD has its own L2: with tile = blockIdx.x the four readers sat on four XCDs
  // and the segment came out of HBM / Infinity Cache four times (FETCH_SIZE 2 x 39 MB for 15 MB of entries).  Blocks
  // of 8 * 2^shift consecutive workgroups take 8 groups, one per XCD: b = 8 G q + r -> group 8 q + r % 8, tile r / 8
  // of it, so a group's readers share one L2 and are dispatched within 8 G workgroups of each other.
  int tile = blockIdx.x;
  if (GROUPED) {
    const int G = 1 << shift, r = blockIdx.x % (8 * G);
    tile = ((blockIdx.x / (8 * G)) * 8 + (r & 7)) * G + (r >> 3);
  }
  if (tile >= n_tiles) return;
  sort_one_tile<GROUPED, kFast, kTSMain, false, DEFER>(tile, n_tiles, offsets, depths, ids_final, tile_ids, key0, id0, key1, id1,
                                                       staging, shift, offsets_out, long_list);
}

// The second launch: the tiles the main kernel listed (long_list[0] of them), one at a time per workgroup.
template <bool GROUPED>
__global__ __launch_bounds__(kTSLong) void tile_depth_sort_long_kernel(
    int n_tiles, const int32_t* __restrict__ offsets, const float* __restrict__ depths,
    uint32_t* ids_final, uint32_t* key0, uint32_t* id0, uint32_t* key1,
    uint32_t* id1, const uint32_t* __restrict__ staging, int shift, int32_t* __restrict__ offsets_out,
    uint32_t* __restrict__ long_list) {
  const uint32_t n_long = min(long_list[0], (uint32_t)n_tiles);
  for (uint32_t i = blockIdx.x; i < n_long; i += gridDim.x) {
    sort_one_tile<GROUPED, kFastXL, kTSLong, true>((int)long_list[1 + i], n_tiles, offsets, depths, ids_final, nullptr, key0, id0,
                                                   key1, id1, staging, shift, offsets_out, long_list);
    __syncthreads();                 // the LDS of one tile is done with before the next one's first store
  }
}

}  // namespace

// temp: four scratch arrays of `capacity` words + the list of long tiles (a counter and up to n_tiles entries)
size_t tile_depth_sort_temp_bytes(uint32_t capacity, int n_tiles) {
  return 4 * align_up((size_t)(capacity ? capacity : 1) * sizeof(uint32_t), 256) +
         align_up(((size_t)(n_tiles > 0 ? n_tiles : 0) + 1) * sizeof(uint32_t), 256);
}

// Sorts flatten_ids[offsets[t] .. offsets[t+1]) of every tile by (depth bits, id).  temp:
// tile_depth_sort_temp_bytes(capacity, n_tiles) bytes.  long_list_zeroed: the caller's earlier kernel has already stored
// a zero in the first word of the long-tile list (tile_depth_sort_long_list(temp, capacity)); otherwise a memset does.
uint32_t* tile_depth_sort_long_list(void* temp, uint32_t capacity) {
  return static_cast<uint32_t*>(temp) + 4 * (align_up((size_t)(capacity ? capacity : 1) * sizeof(uint32_t), 256) / sizeof(uint32_t));
}
int tile_depth_sort(int n_tiles, const int32_t* tile_offsets, const float* depths, uint32_t capacity,
                    uint32_t* flatten_ids, uint32_t* tile_ids_fill, void* temp, hipStream_t stream,
                    const uint32_t* staging, const int32_t* group_offsets, int group_shift, bool long_list_zeroed) {
  if (n_tiles <= 0) return MGS_OK;
  const size_t stride = align_up((size_t)(capacity ? capacity : 1) * sizeof(uint32_t), 256) / sizeof(uint32_t);
  uint32_t* t = static_cast<uint32_t*>(temp);
  uint32_t* long_list = tile_depth_sort_long_list(temp, capacity);
  if (!long_list_zeroed) {       // (only the two-launch form reads the counter; four bytes)
    hipError_t e = hipMemsetAsync(long_list, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return set_error((int)e, "tile_depth_sort: memset: %s", hipGetErrorString(e));
  }
  // average list length the capacity allows: short lists -> the fast path with the smaller LDS list, and NO second launch:
  // where the capacity leaves 640 entries per tile on average a list over 1,024 entries is the rare exception and takes the
  // main kernel's generic path, while the launch of 256 empty 1,024-thread workgroups would cost every frame 4.6 us
  // (bench.py's headline: 4,313 against 4,345 frames/s).  Scenes whose capacity says lists are long get both launches.
#ifdef MGS_TSORT_FORCE_LONG     // measurement (scripts/ab_builds.py): the two-launch form whatever the capacity says
  const bool short_lists = false, mid_lists = false;
#else
  const bool short_lists = (size_t)capacity <= (size_t)n_tiles * 640;
  const bool mid_lists = !short_lists && (size_t)capacity <= (size_t)n_tiles * 1000;
#endif
  // (GROUPED launches are padded to whole blocks of 8 groups: the kernel's XCD-aware tile numbering)
  const int per = 8 << group_shift, n_wg = staging ? (n_tiles + per - 1) / per * per : n_tiles;
#define MGS_TS_LAUNCH(G, F, D, OFFS, STG, SH, OUT)                                                              \
  hipLaunchKernelGGL((tile_depth_sort_kernel<G, F, D>), dim3(n_wg), dim3(kTSMain), 0, stream, n_tiles, OFFS,   \
                     depths, flatten_ids, tile_ids_fill, t, t + stride, t + 2 * stride, t + 3 * stride,    \
                     STG, SH, OUT, long_list)
  if (staging) {
    if (short_lists) MGS_TS_LAUNCH(true, kFastShort, false, group_offsets, staging, group_shift, const_cast<int32_t*>(tile_offsets));
    else if (mid_lists) MGS_TS_LAUNCH(true, kFastMid, true, group_offsets, staging, group_shift, const_cast<int32_t*>(tile_offsets));
    else MGS_TS_LAUNCH(true, kFastLong, true, group_offsets, staging, group_shift, const_cast<int32_t*>(tile_offsets));
  } else {
    if (short_lists) MGS_TS_LAUNCH(false, kFastShort, false, tile_offsets, (const uint32_t*)nullptr, 0, (int32_t*)nullptr);
    else if (mid_lists) MGS_TS_LAUNCH(false, kFastMid, true, tile_offsets, (const uint32_t*)nullptr, 0, (int32_t*)nullptr);
    else MGS_TS_LAUNCH(false, kFastLong, true, tile_offsets, (const uint32_t*)nullptr, 0, (int32_t*)nullptr);
  }
#undef MGS_TS_LAUNCH
  if (short_lists) return check_launch("tile_depth_sort");
  // the lists over kFast entries, listed by the main kernel
  const int n_long_wg = n_tiles < kLongGrid ? n_tiles : kLongGrid;
  if (staging)
    hipLaunchKernelGGL((tile_depth_sort_long_kernel<true>), dim3(n_long_wg), dim3(kTSLong), 0, stream, n_tiles, group_offsets, depths,
                       flatten_ids, t, t + stride, t + 2 * stride, t + 3 * stride, staging, group_shift,
                       const_cast<int32_t*>(tile_offsets), long_list);
  else
    hipLaunchKernelGGL((tile_depth_sort_long_kernel<false>), dim3(n_long_wg), dim3(kTSLong), 0, stream, n_tiles, tile_offsets, depths,
                       flatten_ids, t, t + stride, t + 2 * stride, t + 3 * stride, (const uint32_t*)nullptr, 0, (int32_t*)nullptr,
                       long_list);
  return check_launch("tile_depth_sort");
}

}  // namespace mgs
