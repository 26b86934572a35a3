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
//
// LISTS OVER THE LDS LIST ARE SPLIT INTO UNITS (round 6).  Where the capacity says lists can be long, a tile whose list
// does not fit the main kernel's LDS list is not one workgroup's job (round 5: a second launch of 1,024-thread
// workgroups; one 31 k-entry tile of a clustered scene held the binning for 186-220 us, and lists of 2-8 k entries with
// clustered depths took 50-60 us of bucket levels each).  It becomes ceil(n / kUnit) UNITS by sample sort:
//   main kernel          the tile's workgroup (it has streamed the group's segment and knows [s, e)) claims a descriptor and
//                        nb table entries, ranks a sample of the first entries' composites (depth bits << 32 | id) and
//                        stores every (ns / nb)-th as a SPLITTER: bucket(c) = number of splitters <= c is monotone for any
//                        sample and even for any distribution of depths, identical depths included (the ids split them);
//   unit_collect_kernel  chunks of the group's segment: filter, depth gather, {key, id} into buffer 0 at the tile's own
//                        range (any order: positions are atomic reservations), bucket counts through an LDS histogram;
//   tile_sort_units_kernel   one workgroup per unit: its start is the sum of the lower buckets' counts, its entries are
//                        picked out of the tile's collected list (read-only, L2-resident) into the LDS list and sorted like
//                        a list of their own, straight into flatten_ids.  Work per unit ~kUnit entries whatever the list.
// Same total order on (depth bits, id): flatten_ids stay bit-identical.
#include <type_traits>

#include "mgs_common.h"

namespace mgs {
namespace {

constexpr int kTSMain = 256;        // threads per tile
// Lists longer than the main kernel's LDS list (kFast entries) are not walked by its 256 threads through global scratch
// -- 540 us for ONE tile of 31 k entries on a clustered scene, the tail of the whole binning stage -- but split into units
// of ~kUnit entries (file comment) which a second and a third launch take: the collect kernel and the units' kernel
// (kTSUnit threads, an LDS list of kFastUnit entries, the same levels).  The launches cost a few microseconds even when
// no list is long, so they exist only where the capacity says lists are long (tile_depth_sort below).
#ifndef MGS_TSORT_UNIT
#define MGS_TSORT_UNIT 2048            // entries per unit aimed at (the units' LDS list holds kFastUnit: room for an uneven split)
#endif
#ifndef MGS_TSORT_UNIT_THREADS
#define MGS_TSORT_UNIT_THREADS 512
#endif
#ifndef MGS_TSORT_UNIT_GRID
#define MGS_TSORT_UNIT_GRID 768
#endif
#ifndef MGS_TSORT_COLLECT_GRID
#define MGS_TSORT_COLLECT_GRID 2048
#endif
constexpr int kUnit = MGS_TSORT_UNIT, kTSUnit = MGS_TSORT_UNIT_THREADS, kFastUnit = 4096, kUnitGrid = MGS_TSORT_UNIT_GRID;
constexpr int kMaxUnits = 256;                        // units per list (a list over kMaxUnits * kUnit entries: larger units)
constexpr int kUnitSample = 256;                      // sample a unit's own bucket function is taken from (lds_level) ...
#ifndef MGS_TSORT_ADAPT
#define MGS_TSORT_ADAPT 48
#endif
constexpr int kAdapt = MGS_TSORT_ADAPT;               // ... when the bit field leaves a bucket of more entries than this
#ifndef MGS_TSORT_SAMPLE_FROM
#define MGS_TSORT_SAMPLE_FROM 3     // the group filter starts sampling a list once it holds this many quarters of the LDS list
#endif
constexpr int kSamplePerUnit = 24;                    // ... of which this many per unit are ranked
constexpr int kSample = 512;                          // entries the splitters are taken from at most: every 2^k-th of the list
constexpr int kCollectThreads = 256, kCollectPer = 8, kCollectGrid = MGS_TSORT_COLLECT_GRID;   // unit_collect_kernel
// header of the deferred lists (two 16-byte stores zero it): [0] descriptors claimed, [1] units claimed from the bottom of the
// unit tables (lists of kBigList entries and more: their units read the whole list, the units' kernel starts with them),
// [2] collect chunks claimed, [3] units handed out by the units' kernel beyond its workgroups' first ones, [4] units
// claimed from the top of the tables (all other lists)
constexpr int kListHeader = 8;
constexpr int kBigList = 6144;
// descriptor of a deferred list
enum { kDTile = 0, kDStart, kDCount, kDSegLo, kDSegHi, kDUnits, kDTable, kDCollected, kDChunk, kDescWords = 12 };
// what the three kernels share (all inside tile_depth_sort's temp)
struct SortAux {
  uint32_t* hdr;                     // kListHeader words
  uint32_t max_units;                // entries of the unit tables
  uint32_t* desc;                    // [n_tiles][kDescWords]
  // (records instead of indices into desc: a kernel's first load gives it everything its next loads' addresses need --
  //  every level of dependent loads is 1-2 us in these short kernels)
  uint4* chunk_rec;                  // per chunk of the collect kernel: {descriptor, first segment entry, segment end, list start}
  uint4* unit_rec;                   // per unit: {list start, list entries, first table entry of the list, its units}
  uint32_t* unit_hist;               // per unit: its entry count (zeroed by the main kernel, counted by the collect kernel)
  unsigned long long* unit_split;    // per unit u of a list (but its last): the splitter between buckets u and u + 1
};
#ifndef MGS_TSORT_LONG_BUCKETS
#define MGS_TSORT_LONG_BUCKETS 512     // buckets per level in the units' kernel (the main kernel: 1024): ~2,000 entries per unit; 1,024 would cost the third workgroup per CU its LDS
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
constexpr int kSmallLong = MGS_TSORT_SMALL_LONG;   // ... in the units' kernel (sort_one_tile)
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

#ifdef MGS_TSORT_TIMING
__device__ unsigned g_tsort_pops[4096];     // per workgroup: buckets popped off the stack (+ 0x10000 per bucket over the LDS list)
constexpr unsigned kTsortLog = 16384;
__device__ unsigned long long g_tsort_log[6 * kTsortLog];
__device__ unsigned g_tsort_log_n;
__device__ unsigned long long g_tsort_tf[1 << 16];      // per workgroup of the main kernel: the clock behind the group filter
#endif

// GROUPED: the binning's direct path has dropped the pairs of every 2^shift consecutive tiles into one segment
// of `staging` (entry = id | tile's place in the group << (32 - shift), any order).  The tile's workgroup reads
// its group's segment (the 2^shift workgroups of a group run side by side: L2 hits), keeps its own entries and
// counts those of the group's earlier tiles -- which is where its list starts; it stores that offset too.
// DEFER (main kernel): a list over kFast entries is described, sampled and left to the collect and units' kernels; without
// it the main kernel walks the list itself through global scratch (the generic path below).
// UNIT (the units' kernel): `tile` is a unit's number; its entries come out of the collected list (ck, ci), (key0, id0) is
// scratch nobody else uses.
template <bool GROUPED, int kFast, int kTS, bool UNIT, bool DEFER = true>
__device__ __forceinline__ void sort_one_tile(
    const int tile, int n_tiles, const int32_t* __restrict__ offsets, const float* __restrict__ depths,
    uint32_t* ids_final, uint32_t* __restrict__ tile_ids, uint32_t* key0, uint32_t* id0, uint32_t* key1,
    uint32_t* id1, const uint32_t* __restrict__ staging, int shift, int32_t* __restrict__ offsets_out,
    const SortAux aux, const uint32_t* __restrict__ ck = nullptr, const uint32_t* __restrict__ ci = nullptr) {
  constexpr bool LONG = UNIT;
  constexpr int kBuckets = buckets_for(kFast), kDigitBits = log2i(kBuckets);
  // lds_level's packed scan carries the running entry count in 16 bits and the heavy-bucket count above it
  static_assert(kFast < 65536 && kBuckets < 65536, "the packed bucket scan holds counts below 2^16");
  static_assert(kBuckets % kTS == 0 && kFast % kTS == 0, "buckets and list entries are dealt evenly to the threads");
  // buckets up to kSm entries are finished by rank counting.  The units' kernel takes far larger ones: its candidates
  // come out of LDS (the fast path's list, the generic path's windows), and every bucket it does NOT rank is one more
  // level taken by the whole workgroup, one bucket after the other
  constexpr int kSm = LONG ? kSmallLong : kSmall;
  // One word per bucket.  lds_level (n <= kFast < 2^15 entries) packs {running offset: 16 bits, size: 15 bits, kBrute} into it;
  // the generic loop below (segments of any length) keeps sizes here and its offsets in `gcur`, the last kBuckets words of the
  // LDS list (its windows are that much shorter): 4 KB of LDS less per workgroup -- a seventh workgroup per CU for the
  // 2,048-entry instantiation (206 -> 194 us at configs[4]'s size); an eighth buys the 1,536-entry one nothing (it would need
  // <= 80 SGPRs: scripts/ubench/sgpr_occupancy.hip).
  __shared__ uint32_t cnt[kBuckets];
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
  static_assert(kFast < 32768 && kBuckets / 2 < kFast, "lds_level's packed bucket words; the generic loop's offsets fit the LDS list's tail");
  // (the short-list instantiation walks whole long lists through the generic loop and is not short of LDS: its offsets keep an
  //  array of their own and its windows the whole list)
  constexpr bool kOwnCur = !UNIT && !DEFER;
  __shared__ uint32_t gcur_own[kOwnCur ? kBuckets : 1];
  uint32_t* const gcur = kOwnCur ? gcur_own : reinterpret_cast<uint32_t*>(lc + kFast) - kBuckets;
  uint32_t* li = reinterpret_cast<uint32_t*>(lc);
  // (UNIT) lds_level's splitters: a strided sample of the list, the same ranked, and every scattered entry's bucket
  __shared__ unsigned long long smp[UNIT ? kUnitSample : 1], spl[UNIT ? kUnitSample : 1];
  __shared__ uint16_t lb[UNIT ? kFast : 1];
  // (DEFER, GROUPED) a sample of a long list for its splitters, in the half of lc the ids leave free
  uint32_t* ls = li + kFast;
  __shared__ uint32_t sample_n;
  const int tid = threadIdx.x;
  int s = 0, e = 0, gs = 0, ge = 0;
  uint32_t local = 0, id_mask = ~0u;
  __shared__ uint32_t gcount[2];           // GROUPED: entries of earlier tiles of the group / of this tile
  if constexpr (UNIT) {
    // ---- a unit of a deferred list: where it starts, then its entries out of the collected list --------------------------
    static_assert(!UNIT || kMaxUnits <= kTS, "a unit adds up its list's bucket counts one per thread");
    __shared__ uint32_t u_count;
    const uint4 rec = aux.unit_rec[tile];
    const int s0 = (int)rec.x, n0 = (int)rec.y, toff = (int)rec.z, nb = (int)rec.w, u = tile - toff;
    // everything below the record in flight together: the list's bucket counts, the unit's two splitters and the first trip
    // of the filter (most lists are one trip)
    const uint32_t h = tid < nb ? aux.unit_hist[toff + tid] : 0u;
    const unsigned long long lo = u ? aux.unit_split[toff + u - 1] : 0ull;
    const unsigned long long hi = u < nb - 1 ? aux.unit_split[toff + u] : ~0ull;
    const bool last = u == nb - 1;
    constexpr int kIF = 8;
    uint32_t kk[kIF], ii[kIF];
#pragma unroll
    for (int j = 0; j < kIF; ++j) {
      const int i = j * kTS + tid;
      kk[j] = i < n0 ? ck[s0 + i] : 0u;
      ii[j] = i < n0 ? ci[s0 + i] : 0u;
    }
    if (tid == u) u_count = h;
    if (tid == 0) gcount[1] = 0u;
    uint32_t before;
    (void)block_scan_excl<kTS>(tid < u ? h : 0u, wave_sums, &before);        // (its barriers order the two stores above)
    const int m = (int)u_count;
    if (m == 0) return;                                                      // (uniform)
    s = s0 + (int)before; e = s + m;
    const bool to_lds = m <= kFast;
    for (int i0 = 0; i0 < n0; i0 += kTS * kIF) {
      if (i0) {
#pragma unroll
        for (int j = 0; j < kIF; ++j) {
          const int i = i0 + j * kTS + tid;
          kk[j] = i < n0 ? ck[s0 + i] : 0u;
          ii[j] = i < n0 ? ci[s0 + i] : 0u;
        }
      }
      unsigned long long mm[kIF];
      bool mine[kIF];
      uint32_t run[kIF], wave_total = 0;
#pragma unroll
      for (int j = 0; j < kIF; ++j) {
        const unsigned long long c = ((unsigned long long)kk[j] << 32) | ii[j];
        mine[j] = i0 + j * kTS + tid < n0 && c >= lo && (last || c < hi);
        mm[j] = ballot(mine[j]);
        run[j] = wave_total;
        wave_total += (uint32_t)__popcll(mm[j]);
      }
      uint32_t base = 0;
      if ((tid & 63) == 0 && wave_total) base = atomicAdd(&gcount[1], wave_total);
      base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
      for (int j = 0; j < kIF; ++j)
        if (mine[j]) {
          const uint32_t pos = base + run[j] + mask_rank(mm[j]);
          if (pos < (uint32_t)m) {                      // (always: the collect kernel counted these very entries)
            if (to_lds) lc[pos] = ((unsigned long long)kk[j] << 32) | ii[j];
            else { key0[s + pos] = kk[j]; id0[s + pos] = ii[j]; }
          }
        }
    }
    __syncthreads();
    if (m == 1) {
      if (tid == 0) ids_final[s] = (uint32_t)lc[0];
      return;
    }
  } else if (GROUPED) {
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
#ifdef MGS_TSORT_TIMING
    if (!UNIT && tid == 0) g_tsort_tf[blockIdx.x & 0xffff] = wall_clock64();
#endif
    s = gs + (int)gcount[0];
    e = s + (int)gcount[1];
    if (tid == 0) {
      offsets_out[tile] = s;
      if (tile == n_tiles - 1) offsets_out[n_tiles] = e;
    }
  } else {
    s = offsets[tile]; e = offsets[tile + 1];
  }
  if (!UNIT && DEFER && __builtin_expect(e - s > kFast, 0)) {      // (uniform) a long list: units (its tile ids are filled here)
    if (tile_ids)
      for (int i = s + tid; i < e; i += kTS) tile_ids[i] = (uint32_t)tile;
    __shared__ uint32_t d_slot, d_toff, d_coff;
    const int n = e - s;
    // The sample: every st-th entry of the list itself, or (GROUPED) the list's entries among T positions spread evenly over
    // the group's segment, T chosen so that ~kSample of them are the tile's -- one or two trips of loads that only the few
    // long lists pay.  (The list's FIRST entries, which the filter left in LDS, would be no sample at all: they come from
    // the first workgroups of the binning's scatter, i.e. from a few runs of neighbouring Gaussians -- a 31 k-entry list put
    // 16.9 k entries below its first splitter of twenty.  Sampling inside the filter's loop cost every tile of a 4K frame
    // 7 % of the kernel, whether it ran or not.)
    const int st = (n + kSample - 1) / kSample;
    int ns_all = (n + st - 1) / st < kSample ? (n + st - 1) / st : kSample;
    if (GROUPED) {
      const int seg = ge - gs;
      constexpr int kMaxT = 16 * kTS;
      long long want = (long long)kSample * seg / n;                 // positions to look at
      const int T = (int)(want < kMaxT ? want : kMaxT) < seg ? (int)(want < kMaxT ? want : kMaxT) : seg;
      const int stride = seg / T;                                     // >= 1
      if (tid == 0) sample_n = 0u;
      __syncthreads();
      for (int k0 = 0; k0 < T; k0 += kTS * 8) {
        uint32_t vv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = k0 + j * kTS + tid;
          vv[j] = k < T ? staging[gs + (size_t)k * stride] : ~0u;     // (~0: no tile's entry when shift > 0 ... checked below)
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool hit = k0 + j * kTS + tid < T && (vv[j] >> (32 - shift)) == local;
          const unsigned long long m = ballot(hit);
          uint32_t base = 0;
          if ((tid & 63) == 0 && m) base = atomicAdd(&sample_n, (uint32_t)__popcll(m));
          base = __builtin_amdgcn_readfirstlane(base);
          const uint32_t pos = base + mask_rank(m);
          if (hit && pos < (uint32_t)kSample) ls[pos] = vv[j] & id_mask;
        }
      }
      __syncthreads();
      ns_all = (int)(sample_n < (uint32_t)kSample ? sample_n : (uint32_t)kSample);
    }
    int nb = (n + kUnit - 1) / kUnit < kMaxUnits ? (n + kUnit - 1) / kUnit : kMaxUnits;
    nb = nb < (ns_all > 2 ? ns_all / 2 : 1) ? nb : (ns_all > 2 ? ns_all / 2 : 1);         // two samples per unit at least
    // ~kSamplePerUnit samples per unit are enough (the ranking is quadratic in the sample): every sub-th of those kept
    const int sub = ns_all / (kSamplePerUnit * nb) > 1 ? ns_all / (kSamplePerUnit * nb) : 1;
    const int ns = (ns_all + sub - 1) / sub;
    // chunks of the collect kernel: of the group's segment (GROUPED), or of the list itself
    const int nch = ((GROUPED ? ge - gs : n) + kCollectThreads * kCollectPer - 1) / (kCollectThreads * kCollectPer);
    if (tid == 0) {
      d_slot = atomicAdd(&aux.hdr[0], 1u);
      d_toff = n >= kBigList ? atomicAdd(&aux.hdr[1], (uint32_t)nb) : aux.max_units - (uint32_t)nb - atomicAdd(&aux.hdr[4], (uint32_t)nb);
      d_coff = atomicAdd(&aux.hdr[2], (uint32_t)nch);
    }
    // splitters: the sample's composites ranked by counting, every (ns / nb)-th a splitter
    constexpr int kPer = kSample / kTS;
    static_assert(!DEFER || UNIT || (kSample % kTS == 0 && 2 * (kSample + 8) <= kFast), "the sample's composites sit below the sampled ids in the LDS list");
    uint32_t sid[kPer], skey[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int i = j * kTS + tid;
      sid[j] = i < ns ? (GROUPED ? ls[i * sub] : ids_final[s + (size_t)i * sub * st]) : 0u;
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j) skey[j] = j * kTS + tid < ns ? __float_as_uint(depths[sid[j]]) : 0u;
    __syncthreads();                                 // every id is in registers: the composites overwrite the LDS list
#pragma unroll
    for (int j = 0; j < kPer; ++j)
      if (j * kTS + tid < ns) lc[j * kTS + tid] = ((unsigned long long)skey[j] << 32) | sid[j];
    if (tid < 8) lc[ns + tid] = ~0ull;               // (the ranking reads eight at a time)
    __syncthreads();
    const uint32_t toff = d_toff, slot = d_slot;
    uint32_t rank[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) rank[j] = 0u;
    for (int k0 = 0; k0 < ns; k0 += 8) {             // (eight broadcast reads in flight; the padding compares false)
      unsigned long long c8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) c8[k] = lc[k0 + k];
#pragma unroll
      for (int j = 0; j < kPer; ++j) {
        if (j * kTS >= ns) continue;                 // (uniform)
        const unsigned long long me = ((unsigned long long)skey[j] << 32) | sid[j];
#pragma unroll
        for (int k = 0; k < 8; ++k) rank[j] += c8[k] < me ? 1u : 0u;
      }
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j)
      if (j * kTS + tid < ns) {
        // splitter q (1 .. nb - 1) is the sample's element of rank floor(q ns / nb); ns >= nb: one q per rank at most
        const uint32_t q = (rank[j] * (uint32_t)nb + (uint32_t)ns - 1u) / (uint32_t)ns;
        if (q >= 1u && q < (uint32_t)nb && (q * (uint32_t)ns) / (uint32_t)nb == rank[j])
          aux.unit_split[toff + q - 1u] = ((unsigned long long)skey[j] << 32) | sid[j];
      }
    for (int j = tid; j < nb; j += kTS) {
      aux.unit_rec[toff + j] = make_uint4((uint32_t)s, (uint32_t)n, toff, (uint32_t)nb);
      aux.unit_hist[toff + j] = 0u;
    }
    {
      const int a = GROUPED ? gs : s, b = GROUPED ? ge : e;
      for (int j = tid; j < nch; j += kTS)
        aux.chunk_rec[d_coff + j] = make_uint4(slot, (uint32_t)(a + j * (kCollectThreads * kCollectPer)), (uint32_t)b, (uint32_t)s);
    }
    if (tid == 0) {
      uint32_t* d = aux.desc + (size_t)slot * kDescWords;
      d[kDTile] = (uint32_t)tile; d[kDStart] = (uint32_t)s; d[kDCount] = (uint32_t)n;
      d[kDSegLo] = (uint32_t)gs; d[kDSegHi] = (uint32_t)ge; d[kDUnits] = (uint32_t)nb; d[kDTable] = toff;
      d[kDCollected] = 0u; d[kDChunk] = d_coff;
    }
    return;
  }
#if MGS_TSORT_STOP == 1
  return;
#endif
  if (tile_ids && !UNIT)
    for (int i = s + tid; i < e; i += kTS) tile_ids[i] = (uint32_t)tile;
  if (!UNIT && e - s <= 1) {
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
  // (mode 0: a popped bucket out of (sk, si); 1 = FIRST; 2: a unit whose composites the filter left in the LDS list)
  auto lds_level = [&](auto mode_tag, const int s, const int n, const uint32_t* sk, const uint32_t* si, uint32_t* hk, uint32_t* hi,
                       const uint8_t hsrc, const int stack_base) -> uint32_t {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr bool FIRST = MODE == 1, FROM_LDS = MODE == 2;
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
      else if constexpr (FROM_LDS) ri[it] = (uint32_t)lc[idx];
      else ri[it] = si[s + idx];
    }
    // the keys
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      rk[it] = 0u;
      if (it * kTS >= n) continue;
      if constexpr (FIRST) rk[it] = __float_as_uint(depths[ri[it]]);
      else if constexpr (FROM_LDS) rk[it] = (uint32_t)(lc[min(it * kTS + tid, n - 1)] >> 32);
      else rk[it] = sk[s + min(it * kTS + tid, n - 1)];
    }
    // The bucket of a composite (depth bits << 32 | id): a bit field below the highest bit in which two composites of the
    // list differ -- the bits where the depth keys are not all alike are OR & ~AND over the list (32-bit reductions); only a
    // list of identical depths looks at the ids.
    uint32_t kor = 0u, kand = ~0u;
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
    // A UNIT (FROM_LDS) whose bit field leaves a bucket of more than kAdapt entries switches to SPLITTERS (use_spl): it ranks
    // a strided sample of its list and takes bucket(c) = number of sample elements <= c (binary search) -- ~n / kUnitSample
    // entries per bucket whatever the depths look like.  One outlier stretches the key range until a cluster of a thousand
    // near-identical depths shares a bucket or two of the bit field, rank counting is quadratic in a bucket, and a unit of
    // 1,600 clustered entries took 25-33 us against 9 for the median unit.
    bool use_spl = false;                                   // uniform
    int n_spl = 0;
    auto digit = [&](uint32_t k, uint32_t id) -> unsigned {
      if constexpr (FROM_LDS) {
        if (use_spl) {
          const unsigned long long c = ((unsigned long long)k << 32) | id;
          int lo = 0, hi = n_spl;                            // number of sample elements <= c
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (spl[mid] <= c) lo = mid + 1; else hi = mid;
          }
          return (unsigned)lo;
        }
      }
      if (key_digit) return (k >> kshift) & (kBuckets - 1);
      const unsigned long long c = ((unsigned long long)k << 32) | id;
      return (unsigned)(c >> shift) & (kBuckets - 1);
    };
    unsigned rd[FROM_LDS ? kItems : 1];                      // (units) the entries' buckets: one search each with splitters
    // histogram of the buckets, exclusive scan, stack slots for the buckets over `heavy_over` entries (commit: a trial pass
    // only counts them).  Returns the number of such buckets.
    auto hist_and_scan = [&](const uint32_t heavy_over, const bool commit) -> uint32_t {
#pragma unroll
      for (int it = 0; it < kItems; ++it) {
        if constexpr (FROM_LDS) rd[it] = 0u;
        if (it * kTS >= n) continue;
        if (it * kTS + tid < n) {
          const unsigned d = digit(rk[it], ri[it]);
          if constexpr (FROM_LDS) rd[it] = d;
          atomicAdd(&cnt[d], 1u);
        }
      }
      __syncthreads();
      // one scan for both: bucket sizes in the low 16 bits (n < 2^16), heavy-bucket count above
      uint32_t c4[kBuckets / kTS], packed = 0;
#pragma unroll
      for (int k = 0; k < kBuckets / kTS; ++k) {
        c4[k] = cnt[tid * (kBuckets / kTS) + k];
        packed += c4[k] + (c4[k] > heavy_over ? 0x10000u : 0u);
      }
      uint32_t tot;
      uint32_t exp = block_scan_excl<kTS>(packed, wave_sums, &tot);
      uint32_t ex = exp & 0xffffu, hx = exp >> 16;
      const uint32_t heavy = tot >> 16;
#pragma unroll
      for (int k = 0; k < kBuckets / kTS; ++k) {
        const int d = tid * (kBuckets / kTS) + k;
        uint32_t word = ex | (c4[k] << 16);
        if (commit && c4[k] > heavy_over) {      // the generic loop below takes it from buffer (hk, hi)
          const int slot = stack_base + (int)hx;
          if (slot < kStack) {
            stack_lo[slot] = s + (int)ex; stack_hi[slot] = s + (int)(ex + c4[k]); stack_src[slot] = hsrc;
          } else {
            word |= kBrute;
          }
          ++hx;
        }
        cnt[d] = word;
        ex += c4[k];
      }
      if (tid == 0) stack_n = !commit ? stack_base : (stack_base + (int)heavy < kStack ? stack_base + (int)heavy : kStack);
      __syncthreads();
      return heavy;
    };
    uint32_t htot;
    if constexpr (FROM_LDS) {
      static_assert(!FROM_LDS || (kUnitSample <= kTS && kUnitSample < kBuckets && kUnitSample % 8 == 0),
                    "one sample element per thread, one bucket more than samples");
      htot = hist_and_scan((uint32_t)kAdapt, false);         // (no bucket over kAdapt: none over kSm either, nothing to commit)
      if (htot) {                                            // uniform
        const int st = (n + kUnitSample - 1) / kUnitSample;  // every st-th entry of the list as the filter left it
        n_spl = (n + st - 1) / st;
        if (tid < kUnitSample) smp[tid] = tid < n_spl ? lc[tid * st] : ~0ull;
#pragma unroll
        for (int k = 0; k < kBuckets / kTS; ++k) cnt[tid + k * kTS] = 0;
        __syncthreads();
        if (tid < ((n_spl + 63) & ~63)) {                    // (whole waves: the sample's owners)
          const unsigned long long my = smp[tid < n_spl ? tid : 0];
          uint32_t rank = 0u;
          for (int k0 = 0; k0 < n_spl; k0 += 8) {            // (eight broadcast reads in flight; the padding compares false)
            unsigned long long c8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) c8[k] = smp[k0 + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) rank += c8[k] < my ? 1u : 0u;
          }
          if (tid < n_spl) spl[rank] = my;                   // distinct composites: the ranks are a permutation
        }
        use_spl = true;
        __syncthreads();
        htot = hist_and_scan((uint32_t)kSm, true);
      }
    } else {
      htot = hist_and_scan((uint32_t)kSm, true);
    }
#if MGS_TSORT_STOP == 3
    if (n > 0) { if (tid == 0) ids_final[s] = cnt[3]; return 0u; }
#endif
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      if (it * kTS >= n) continue;
      if (it * kTS + tid < n) {
        unsigned d;
        if constexpr (FROM_LDS) d = rd[it]; else d = digit(rk[it], ri[it]);
        const uint32_t p = atomicAdd(&cnt[d], 1u) & 0xffffu;       // (the offset half never carries: it ends at the bucket's end <= n)
        lc[p] = ((unsigned long long)rk[it] << 32) | ri[it];
        if constexpr (FROM_LDS) lb[p] = (uint16_t)d;
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
        unsigned d;
        if constexpr (FROM_LDS) d = lb[i]; else d = digit(k, id);
        const uint32_t craw = cnt[d];                  // {the bucket's END by now, its size, kBrute}
        const uint32_t b = (craw >> 16) & 0x7fffu;
        if (b > (uint32_t)kSm && !(craw & kBrute)) {   // heavy: hand it to the generic loop
          hk[s + i] = k;
          hi[s + i] = id;
        } else {
          const int be = (int)(craw & 0xffffu), bs = be - (int)b;
          int c = 0;
          for (int j = bs; j < be; ++j) c += lc[j] < me ? 1 : 0;
          ids_final[s + bs + c] = id;
        }
      }
    }
    return htot;
  };

  const int n = n_list;
  using Mode0 = std::integral_constant<int, 0>;
  using ModeList = std::integral_constant<int, UNIT ? 2 : 1>;
  if (n <= kFast) {
    // ---- fast path: the whole list is one level out of LDS ---------------------------------------
    const uint32_t htot = lds_level(ModeList{}, s, n, (const uint32_t*)nullptr, (const uint32_t*)nullptr, key1, id1, (uint8_t)1, 0);
    if (htot == 0) return;                          // uniform
    __syncthreads();
  } else if (UNIT) {
    // a unit over the LDS list (the sample missed a cluster, or a list of more than kMaxUnits units): the filter left it in
    // buffer 0
    if (tid == 0) {
      stack_n = 1;
      stack_lo[0] = s; stack_hi[0] = e; stack_src[0] = 0;
    }
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
#ifdef MGS_TSORT_TIMING
    if (UNIT && tid == 0) atomicAdd(&g_tsort_pops[blockIdx.x], 1u + ((stack_hi[sn - 1] - stack_lo[sn - 1] > kFast) ? 0x10000u : 0u));
#endif
    const int lo = stack_lo[sn - 1], hi = stack_hi[sn - 1];
    const int src = stack_src[sn - 1];
    __syncthreads();
    if (tid == 0) stack_n = sn - 1;
    const uint32_t* sk = src ? key1 : key0;
    const uint32_t* si = src ? id1 : id0;
    uint32_t* dk = src ? key0 : key1;
    uint32_t* di = src ? id0 : id1;
    const int m = hi - lo;

    // (the units' kernel only: a second copy of the level takes the main kernel from 47-53 to 64-86 VGPRs, a
    //  workgroup per CU less for every scene, and its heavy buckets are few)
    if constexpr (LONG && MGS_TSORT_LDS_LEVELS != 0) {
      if (m > kSm && m <= kFast) {              // fits the LDS list: one level out of LDS (heavy sub-buckets to the other buffer)
        (void)lds_level(Mode0{}, lo, m, sk, si, dk, di, (uint8_t)(src ^ 1), sn - 1);
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
        gcur[d] = ex;
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
          const uint32_t p = atomicAdd(&gcur[digit(kk[j], ii[j])], 1u);
          dk[lo + p] = kk[j];
          di[lo + p] = ii[j];
        }
    }
    __syncthreads();                               // the scattered segment is visible to the workgroup

    // gcur[d] is now the END of bucket d: every element of a light bucket ranks itself inside it.  The scattered
    // segment goes through the LDS list in windows of kFast - 2 kSm entries with kSm more on either side (a light
    // bucket holds at most kSm entries, so the whole bucket of every element of the window is in LDS): one coalesced
    // load per entry instead of a dependent global load per CANDIDATE -- a 31 k-entry list has ~30 of them per element.
    constexpr int kWin = kFast - (kOwnCur ? 0 : kBuckets / 2) - 2 * kSm;      // (the list's last kBuckets words are gcur)
    static_assert(kWin >= 64, "a window of the generic loop's ranking pass");
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
        const int be = lo + (int)gcur[d], bs = be - (int)b;
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

#ifndef MGS_TSORT_MAIN_SGPRS
#define MGS_TSORT_MAIN_SGPRS 96
#endif
template <bool GROUPED, int kFast, bool DEFER>
__global__ __launch_bounds__(kTSMain) __attribute__((amdgpu_num_sgpr(MGS_TSORT_MAIN_SGPRS))) void tile_depth_sort_kernel(
    int n_tiles, const int32_t* __restrict__ offsets, const float* __restrict__ depths,
    uint32_t* ids_final, uint32_t* __restrict__ tile_ids, uint32_t* key0, uint32_t* id0, uint32_t* key1,
    uint32_t* id1, const uint32_t* __restrict__ staging, int shift, int32_t* __restrict__ offsets_out,
    const SortAux aux) {
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
#ifdef MGS_TSORT_TIMING            // measurement build: per tile {start, end} on the 100 MHz clock (scripts/dbg/main_sort_timeline.py)
  const unsigned long long t_start = wall_clock64();
#endif
  sort_one_tile<GROUPED, kFast, kTSMain, false, DEFER>(tile, n_tiles, offsets, depths, ids_final, tile_ids, key0, id0, key1, id1,
                                                       staging, shift, offsets_out, aux);
#ifdef MGS_TSORT_TIMING
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned k = atomicAdd(&g_tsort_log_n, 1u);
    if (k < kTsortLog) {
      unsigned long long* r = g_tsort_log + 6 * (size_t)k;
      r[0] = blockIdx.x; r[1] = g_tsort_tf[blockIdx.x & 0xffff]; r[2] = (unsigned long long)tile; r[3] = 0xffffull;      // (3: marks a tile of the main kernel)
      r[4] = t_start; r[5] = wall_clock64();
    }
  }
#endif
}

// Deferred lists, second step: the entries of every described list out of its group's segment (GROUPED) or out of the list
// itself, with their depth keys, into buffer 0 at the list's own range, and the units' entry counts.  Work items are
// chunks of kCollectThreads * kCollectPer segment entries; the main kernel has numbered them (chunk_rec).
template <bool GROUPED>
__global__ __launch_bounds__(kCollectThreads) void unit_collect_kernel(
    uint32_t max_chunks, const float* __restrict__ depths, const uint32_t* __restrict__ ids_final, const uint32_t* __restrict__ staging,
    int shift, uint32_t* __restrict__ key0, uint32_t* __restrict__ id0, const SortAux aux) {
  const uint32_t claimed = aux.hdr[2];
  const uint32_t n_chunks = claimed < max_chunks ? claimed : max_chunks;
  __shared__ uint32_t lh[kMaxUnits];
  __shared__ unsigned long long sp[kMaxUnits];
  static_assert(kMaxUnits <= kCollectThreads, "one thread per unit loads the splitters and flushes the counts");
  const int tid = threadIdx.x;
  for (uint32_t f = blockIdx.x; f < n_chunks; f += gridDim.x) {
    const uint4 rec = aux.chunk_rec[f];                // {descriptor, first segment entry, segment end, list start}
    uint32_t* d = aux.desc + (size_t)rec.x * kDescWords;
    const int i0 = (int)rec.y, b = (int)rec.z, s = (int)rec.w;
    uint32_t vv[kCollectPer], kk[kCollectPer], pos[kCollectPer];
    bool mine[kCollectPer];
#pragma unroll
    for (int j = 0; j < kCollectPer; ++j) {            // (beside the descriptor's words: both need the record only)
      const int i = i0 + j * kCollectThreads + tid;
      vv[j] = i < b ? (GROUPED ? staging[i] : ids_final[i]) : 0u;
    }
    const int n = (int)d[kDCount], nb = (int)d[kDUnits];
    const uint32_t toff = d[kDTable];
    const uint32_t local = d[kDTile] & ((1u << shift) - 1u), id_mask = GROUPED ? (1u << (32 - shift)) - 1u : ~0u;
    __syncthreads();                                   // (the previous chunk's flush is done)
    if (tid < nb - 1) sp[tid] = aux.unit_split[toff + tid];
    if (tid < nb) lh[tid] = 0u;
    if (GROUPED) {
      unsigned long long mm[kCollectPer];
      uint32_t run[kCollectPer], wave_total = 0;
#pragma unroll
      for (int j = 0; j < kCollectPer; ++j) {
        mine[j] = i0 + j * kCollectThreads + tid < b && (vv[j] >> (32 - shift)) == local;
        mm[j] = ballot(mine[j]);
        run[j] = wave_total;
        wave_total += (uint32_t)__popcll(mm[j]);
      }
      uint32_t base = 0;
      if ((tid & 63) == 0 && wave_total) base = atomicAdd(&d[kDCollected], wave_total);
      base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
      for (int j = 0; j < kCollectPer; ++j) pos[j] = base + run[j] + mask_rank(mm[j]);
    } else {
#pragma unroll
      for (int j = 0; j < kCollectPer; ++j) {
        const int i = i0 + j * kCollectThreads + tid;
        mine[j] = i < b;
        pos[j] = (uint32_t)(i - s);
      }
    }
#pragma unroll
    for (int j = 0; j < kCollectPer; ++j) kk[j] = mine[j] ? __float_as_uint(depths[vv[j] & id_mask]) : 0u;
    // the chunk's counts in LDS first: thousands of entries of a clustered list fall into one unit, and that many device
    // atomics on one address are served one after the other
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kCollectPer; ++j)
      if (mine[j] && pos[j] < (uint32_t)n) {           // (always: the tile's own count)
        const uint32_t id = vv[j] & id_mask;
        key0[s + pos[j]] = kk[j];
        id0[s + pos[j]] = id;
        const unsigned long long comp = ((unsigned long long)kk[j] << 32) | id;
        int lo = 0, hi = nb - 1;                        // bucket = number of splitters <= comp
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (sp[mid] <= comp) lo = mid + 1; else hi = mid;
        }
        atomicAdd(&lh[lo], 1u);
      }
    __syncthreads();
    if (tid < nb && lh[tid]) atomicAdd(&aux.unit_hist[toff + tid], lh[tid]);
  }
}

// Deferred lists, third step: one unit at a time per workgroup.  (key0, id0) here is the units' own scratch, the collected
// lists are (ck, ci).
// (compiled for six waves per SIMD: 71 VGPRs without a spill where the launch bound alone let the compiler take 113 -- three
//  workgroups per CU, which is what LDS allows, and with units of 2,048 entries the clustered scene's ~750 units are resident at once)
#ifndef MGS_TSORT_UNIT_WAVES
#define MGS_TSORT_UNIT_WAVES 6
#endif
__global__ __launch_bounds__(kTSUnit) __attribute__((amdgpu_waves_per_eu(MGS_TSORT_UNIT_WAVES, 8))) void tile_sort_units_kernel(
    int n_tiles, uint32_t* ids_final, const uint32_t* __restrict__ ck, const uint32_t* __restrict__ ci, uint32_t* key0,
    uint32_t* id0, uint32_t* key1, uint32_t* id1, const SortAux aux) {
  // units [0, n_big) of the tables belong to the big lists, the last n_small to the others: item i is unit i, or unit
  // max_units - 1 - (i - n_big)
  const uint32_t n_big = aux.hdr[1] < aux.max_units ? aux.hdr[1] : aux.max_units;
  const uint32_t n_small = aux.hdr[4] < aux.max_units - n_big ? aux.hdr[4] : aux.max_units - n_big;
  const uint32_t n_units = n_big + n_small;
#ifndef MGS_TSORT_DYNAMIC
#define MGS_TSORT_DYNAMIC 0        // 1: a workgroup's units beyond its first come off a counter (measured: 47 against 41-42 us)
#endif
  __shared__ uint32_t next_f;
  uint32_t f = blockIdx.x;
  while (f < n_units) {
    if (MGS_TSORT_DYNAMIC && threadIdx.x == 0) next_f = gridDim.x + atomicAdd(&aux.hdr[3], 1u);     // (in flight while the unit is sorted)
#ifdef MGS_TSORT_TIMING            // measurement build: per unit {start, end} on the 100 MHz clock (scripts/dbg/long_sort_timeline.py)
    const unsigned long long t_start = wall_clock64();
#endif
    const uint32_t unit = f < n_big ? f : aux.max_units - 1u - (f - n_big);
    sort_one_tile<false, kFastUnit, kTSUnit, true>((int)unit, n_tiles, nullptr, nullptr, ids_final, nullptr, key0, id0, key1, id1,
                                                   nullptr, 0, nullptr, aux, ck, ci);
    __syncthreads();                 // the LDS of one unit is done with before the next one's first store
#ifdef MGS_TSORT_TIMING
    if (threadIdx.x == 0) {
      const unsigned k = atomicAdd(&g_tsort_log_n, 1u);
      if (k < kTsortLog) {
        const uint4 rec = aux.unit_rec[unit];
        unsigned long long* r = g_tsort_log + 6 * (size_t)k;
        r[0] = blockIdx.x | ((unsigned long long)g_tsort_pops[blockIdx.x] << 32); r[1] = aux.unit_hist[unit]; r[2] = rec.x; r[3] = unit - rec.z;
        g_tsort_pops[blockIdx.x] = 0u;
        r[4] = t_start; r[5] = wall_clock64();
      }
    }
#endif
    if (MGS_TSORT_DYNAMIC) {
      f = next_f;
      __syncthreads();               // (read before the next round's store)
    } else {
      f += gridDim.x;
    }
  }
}

// where the capacity leaves at most this many entries per tile on average, lists are short: the LDS list is the small one
// and lists over it stay with the main kernel (tile_depth_sort below)
bool capacity_says_short_lists(uint32_t capacity, int n_tiles) { return (size_t)capacity <= (size_t)(n_tiles > 0 ? n_tiles : 0) * 640; }

// units a call can claim: a list of n entries takes ceil(n / kUnit) <= n / kUnit + 1
size_t max_units_for(uint32_t capacity, int n_tiles) { return (size_t)capacity / kUnit + (size_t)(n_tiles > 0 ? n_tiles : 0) + 1; }
// collect chunks a call can claim: every list brings its group's segment (up to 2^group_shift lists share one) or itself
size_t max_chunks_for(uint32_t capacity, int n_tiles, int group_shift) {
  return (((size_t)capacity << group_shift) / (kCollectThreads * kCollectPer)) + (size_t)(n_tiles > 0 ? n_tiles : 0) + 1;
}
constexpr int kMaxGroupShift = 7;     // (the binning groups up to 2^7 tiles under its debug knob; 2^2 by default)

struct AuxLayout {
  size_t hdr, desc, chunk_rec, unit_rec, unit_hist, unit_split, total;     // byte offsets behind the four scratch arrays
  AuxLayout(uint32_t capacity, int n_tiles) {
    const size_t nt = (size_t)(n_tiles > 0 ? n_tiles : 0), mu = max_units_for(capacity, n_tiles);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align_up(bytes, 256); return at; };
    hdr = take(kListHeader * sizeof(uint32_t));
    desc = take(nt * kDescWords * sizeof(uint32_t));
    chunk_rec = take(max_chunks_for(capacity, n_tiles, kMaxGroupShift) * sizeof(uint4));
    unit_rec = take(mu * sizeof(uint4));
    unit_hist = take(mu * sizeof(uint32_t));
    unit_split = take(mu * sizeof(unsigned long long));
    total = o;
  }
};

}  // namespace

#ifdef MGS_TSORT_TIMING
// measurement build only: copies the work-item log to the host (6 x uint64 per unit: workgroup, entries, tile, unit, start,
// end on the 100 MHz clock), returns the item count and resets it
extern "C" unsigned mgs_debug_tsort_log(unsigned long long* dst, unsigned max_items) {
  unsigned n = 0, zero = 0;
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_tsort_log_n), sizeof(n));
  if (n > kTsortLog) n = kTsortLog;
  if (n > max_items) n = max_items;
  (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tsort_log), (size_t)n * 6 * sizeof(unsigned long long));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tsort_log_n), &zero, sizeof(zero));
  return n;
}
#endif

// temp: four scratch arrays of `capacity` words + the deferred lists' header, descriptors and unit tables
size_t tile_depth_sort_temp_bytes(uint32_t capacity, int n_tiles) {
  return 4 * align_up((size_t)(capacity ? capacity : 1) * sizeof(uint32_t), 256) + AuxLayout(capacity, n_tiles).total;
}

// Sorts flatten_ids[offsets[t] .. offsets[t+1]) of every tile by (depth bits, id).  temp:
// tile_depth_sort_temp_bytes(capacity, n_tiles) bytes.  long_list_zeroed: the caller's earlier kernel has already stored
// zeros in the four header words of the deferred lists (tile_depth_sort_long_list(temp, capacity)); otherwise a memset does.
// scratch_k / scratch_i: two more arrays of `capacity` words nobody reads once the lists have been collected (the binning's
// radix ping-pong buffers; scratch_i may be `staging` itself) -- the units' second scratch pair.
uint32_t* tile_depth_sort_long_list(void* temp, uint32_t capacity) {
  return static_cast<uint32_t*>(temp) + 4 * (align_up((size_t)(capacity ? capacity : 1) * sizeof(uint32_t), 256) / sizeof(uint32_t));
}
int tile_depth_sort(int n_tiles, const int32_t* tile_offsets, const float* depths, uint32_t capacity,
                    uint32_t* flatten_ids, uint32_t* tile_ids_fill, void* temp, hipStream_t stream,
                    uint32_t* scratch_k, uint32_t* scratch_i,
                    const uint32_t* staging, const int32_t* group_offsets, int group_shift, bool long_list_zeroed) {
  if (n_tiles <= 0) return MGS_OK;
  const size_t stride = align_up((size_t)(capacity ? capacity : 1) * sizeof(uint32_t), 256) / sizeof(uint32_t);
  uint32_t* t = static_cast<uint32_t*>(temp);
  char* aux_base = reinterpret_cast<char*>(tile_depth_sort_long_list(temp, capacity));
  const AuxLayout lay(capacity, n_tiles);
  SortAux aux;
  aux.hdr = reinterpret_cast<uint32_t*>(aux_base + lay.hdr);
  aux.max_units = (uint32_t)max_units_for(capacity, n_tiles);
  aux.desc = reinterpret_cast<uint32_t*>(aux_base + lay.desc);
  aux.chunk_rec = reinterpret_cast<uint4*>(aux_base + lay.chunk_rec);
  aux.unit_rec = reinterpret_cast<uint4*>(aux_base + lay.unit_rec);
  aux.unit_hist = reinterpret_cast<uint32_t*>(aux_base + lay.unit_hist);
  aux.unit_split = reinterpret_cast<unsigned long long*>(aux_base + lay.unit_split);
  if (!long_list_zeroed) {       // (only the three-launch form reads the header; 32 bytes)
    hipError_t e = hipMemsetAsync(aux.hdr, 0, kListHeader * sizeof(uint32_t), stream);
    if (e != hipSuccess) return set_error((int)e, "tile_depth_sort: memset: %s", hipGetErrorString(e));
  }
  // average list length the capacity allows: short lists -> the fast path with the smaller LDS list, and NO further launch:
  // where the capacity leaves 640 entries per tile on average a list over 1,024 entries is the rare exception and takes the
  // main kernel's generic path, while launches that find nothing to do would cost every frame a few microseconds
  // (bench.py's headline: 4,313 against 4,345 frames/s for one of them).  Scenes whose capacity says lists are long get all three.
#ifdef MGS_TSORT_FORCE_LONG     // measurement (scripts/ab_builds.py): the three-launch form whatever the capacity says
  const bool short_lists = false, mid_lists = false;
#else
  const bool short_lists = capacity_says_short_lists(capacity, n_tiles) || !scratch_k || !scratch_i;
  const bool mid_lists = !short_lists && (size_t)capacity <= (size_t)n_tiles * 1000;
#endif
  // (GROUPED launches are padded to whole blocks of 8 groups: the kernel's XCD-aware tile numbering)
  const int per = 8 << group_shift, n_wg = staging ? (n_tiles + per - 1) / per * per : n_tiles;
#define MGS_TS_LAUNCH(G, F, D, OFFS, STG, SH, OUT)                                                              \
  hipLaunchKernelGGL((tile_depth_sort_kernel<G, F, D>), dim3(n_wg), dim3(kTSMain), 0, stream, n_tiles, OFFS,   \
                     depths, flatten_ids, tile_ids_fill, t, t + stride, t + 2 * stride, t + 3 * stride,    \
                     STG, SH, OUT, aux)
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
  // the deferred lists: their entries into buffer 0 with the units' counts, then the units (launches that read a zero and
  // leave where no list is long)
  if (staging)
    hipLaunchKernelGGL((unit_collect_kernel<true>), dim3(kCollectGrid), dim3(kCollectThreads), 0, stream,
                       (uint32_t)max_chunks_for(capacity, n_tiles, kMaxGroupShift), depths, flatten_ids, staging, group_shift, t, t + stride, aux);
  else
    hipLaunchKernelGGL((unit_collect_kernel<false>), dim3(kCollectGrid), dim3(kCollectThreads), 0, stream,
                       (uint32_t)max_chunks_for(capacity, n_tiles, kMaxGroupShift), depths, flatten_ids, (const uint32_t*)nullptr, 0, t,
                       t + stride, aux);
  hipLaunchKernelGGL(tile_sort_units_kernel, dim3(kUnitGrid), dim3(kTSUnit), 0, stream, n_tiles, flatten_ids, t, t + stride,
                     scratch_k, scratch_i, t + 2 * stride, t + 3 * stride, aux);
  return check_launch("tile_depth_sort");
}

}  // namespace mgs
