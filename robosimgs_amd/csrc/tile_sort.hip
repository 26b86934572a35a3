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
#include "mgs_common.h"

namespace mgs {
namespace {

constexpr int kTS = 256;            // threads per tile
constexpr int kBuckets = 1024;
constexpr int kSmall = 48;          // buckets up to this size are finished by rank counting
constexpr int kStack = 96;          // pending heavy buckets; beyond it a bucket is rank-counted whatever its size
#ifndef MGS_TSORT_FAST
#define MGS_TSORT_FAST 2048
#endif
constexpr int kFastLong = MGS_TSORT_FAST;   // longest list of the LDS-resident fast path (8 KiB of LDS per 1024) ...
constexpr int kFastShort = 1024;            // ... and where the capacity says lists are short on average: 17 instead of
                                            // 25 KiB of LDS per workgroup = 8 instead of 6 workgroups per CU
constexpr uint32_t kBrute = 0x80000000u;

__device__ __forceinline__ bool comp_less(uint32_t ka, uint32_t ia, uint32_t kb, uint32_t ib) {
  return ka < kb || (ka == kb && ia < ib);
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int d) {
  unsigned lo = __shfl_xor((unsigned)v, d), hi = __shfl_xor((unsigned)(v >> 32), d);
  return ((unsigned long long)hi << 32) | lo;
}

// exclusive scan of one value per thread over the workgroup; *total = sum
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* wave_sums, uint32_t* total) {
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
template <bool GROUPED, int kFast>
__global__ __launch_bounds__(kTS) void tile_depth_sort_kernel(
    int n_tiles, const int32_t* __restrict__ offsets, const float* __restrict__ depths,
    uint32_t* ids_final, uint32_t* __restrict__ tile_ids, uint32_t* key0, uint32_t* id0, uint32_t* key1,
    uint32_t* id1, const uint32_t* __restrict__ staging, int shift, int32_t* __restrict__ offsets_out) {
  __shared__ uint32_t cnt[kBuckets];
  __shared__ uint32_t cur[kBuckets];
  __shared__ unsigned long long red_min[kTS / 64], red_max[kTS / 64];
  __shared__ uint32_t wave_sums[kTS / 64];
  __shared__ int stack_lo[kStack], stack_hi[kStack];
  __shared__ uint8_t stack_src[kStack];
  __shared__ int stack_n;
  constexpr int kItems = kFast / kTS;
  __shared__ uint32_t lk[kFast], li[kFast];
  const int tile = blockIdx.x;
  if (tile >= n_tiles) return;
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
#pragma unroll
      for (int j = 0; j < kInFlight; ++j) {
        const int i = i0 + j * kTS + tid;
        if (i0 + j * kTS >= ge) continue;               // uniform
        const uint32_t v = vv[j], l = v >> (32 - shift);
        below += (i < ge && l < local) ? 1u : 0u;
        const bool mine = i < ge && l == local;
        const unsigned long long m = __ballot(mine);
        uint32_t base = 0;
        if ((tid & 63) == 0 && m) base = atomicAdd(&gcount[1], (uint32_t)__popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (mine) {
          const uint32_t pos = base + mask_rank(m);
          if (pos < (uint32_t)kFast) li[pos] = v & id_mask;
        }
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) below += __shfl_xor(below, d);
    if ((tid & 63) == 0 && below) atomicAdd(&gcount[0], below);
    __syncthreads();
    s = gs + (int)gcount[0];
    e = s + (int)gcount[1];
    if (tid == 0) {
      offsets_out[tile] = s;
      if (tile == n_tiles - 1) offsets_out[n_tiles] = e;
    }
  } else {
    s = offsets[tile]; e = offsets[tile + 1];
  }
  if (tile_ids)
    for (int i = s + tid; i < e; i += kTS) tile_ids[i] = (uint32_t)tile;
  if (e - s <= 1) {
    if (GROUPED && e - s == 1 && tid == 0) ids_final[s] = li[0];
    return;
  }
  const int n = e - s;

  if (n <= kFast) {
    // ---- fast path: registers + LDS ------------------------------------------------------------
    uint32_t rk[kItems], ri[kItems];
    // (every item loop below stops, wave-uniformly, at the first item no thread of the workgroup owns:
    //  a 455-entry list executes two of the eight unrolled trips)
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      ri[it] = 0u;
      if (it * kTS >= n) continue;
      const int idx = it * kTS + tid;
      ri[it] = idx < n ? (GROUPED ? li[idx] : ids_final[s + idx]) : 0u;
    }
    unsigned long long mn = ~0ull, mx = 0ull;
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      const int idx = it * kTS + tid;
      rk[it] = 0u;
      if (it * kTS >= n) continue;
      if (idx < n) {
        rk[it] = __float_as_uint(depths[ri[it]]);
        const unsigned long long c = ((unsigned long long)rk[it] << 32) | ri[it];
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
    if (tid == 0) stack_n = 0;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kTS / 64; ++w) {
      mn = red_min[w] < mn ? red_min[w] : mn;
      mx = red_max[w] > mx ? red_max[w] : mx;
    }
    const int hb = 63 - __clzll((long long)(mn ^ mx));
    const int shift = hb > 9 ? hb - 9 : 0;
    auto digit = [&](uint32_t k, uint32_t id) -> unsigned {
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
        packed += c4[k] + (c4[k] > (uint32_t)kSmall ? 0x10000u : 0u);
      }
      uint32_t tot;
      uint32_t exp = block_scan_excl(packed, wave_sums, &tot);
      uint32_t ex = exp & 0xffffu, hx = exp >> 16;
      htot = tot >> 16;
#pragma unroll
      for (int k = 0; k < kBuckets / kTS; ++k) {
        const int d = tid * (kBuckets / kTS) + k;
        cur[d] = ex;
        if (c4[k] > (uint32_t)kSmall) {            // the generic loop below takes it from buffer 1
          if ((int)hx < kStack) {
            stack_lo[hx] = s + (int)ex; stack_hi[hx] = s + (int)(ex + c4[k]); stack_src[hx] = 1;
          } else {
            cnt[d] = c4[k] | kBrute;
          }
          ++hx;
        }
        ex += c4[k];
      }
      if (tid == 0) stack_n = (int)htot < kStack ? (int)htot : kStack;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      if (it * kTS >= n) continue;
      if (it * kTS + tid < n) {
        const uint32_t p = atomicAdd(&cur[digit(rk[it], ri[it])], 1u);
        lk[p] = rk[it];
        li[p] = ri[it];
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      const int i = it * kTS + tid;
      if (it * kTS >= n) continue;
      if (i < n) {
        const uint32_t k = lk[i], id = li[i];
        const unsigned d = digit(k, id);
        const uint32_t craw = cnt[d];
        const uint32_t b = craw & ~kBrute;
        if (b > (uint32_t)kSmall && !(craw & kBrute)) {   // heavy: hand it to the generic loop
          key1[s + i] = k;
          id1[s + i] = id;
        } else {
          const int be = (int)cur[d], bs = be - (int)b;
          int c = 0;
          for (int j = bs; j < be; ++j) c += comp_less(lk[j], li[j], k, id) ? 1 : 0;
          ids_final[s + bs + c] = id;
        }
      }
    }
    if (htot == 0) return;                          // uniform
    __syncthreads();
  } else {
    // level 0 input: the tile's ids; keys are gathered once and parked in buffer 0
    if (GROUPED) {          // longer than the LDS list: collect again, straight into buffer 0
      if (tid == 0) gcount[1] = 0u;
      __syncthreads();
      for (int i0 = gs; i0 < ge; i0 += kTS) {
        const int i = i0 + tid;
        const uint32_t v = i < ge ? staging[i] : 0u;
        const bool mine = i < ge && (v >> (32 - shift)) == local;
        const unsigned long long m = __ballot(mine);
        uint32_t base = 0;
        if ((tid & 63) == 0 && m) base = atomicAdd(&gcount[1], (uint32_t)__popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (mine) {
          const uint32_t pos = base + mask_rank(m), id = v & id_mask;
          key0[s + pos] = __float_as_uint(depths[id]);
          id0[s + pos] = id;
        }
      }
    } else {
      for (int i = s + tid; i < e; i += kTS) {
        const uint32_t id = ids_final[i];
        key0[i] = __float_as_uint(depths[id]);
        id0[i] = id;
      }
    }
    if (tid == 0) {
      stack_n = 1;
      stack_lo[0] = s; stack_hi[0] = e; stack_src[0] = 0;
    }
    __syncthreads();
  }

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

    if (m <= kSmall) {                             // whole segment by rank counting
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
    unsigned long long mn = ~0ull, mx = 0ull;
    for (int i = lo + tid; i < hi; i += kTS) {
      const unsigned long long c = ((unsigned long long)sk[i] << 32) | si[i];
      mn = c < mn ? c : mn;
      mx = c > mx ? c : mx;
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
    const int shift = hb > 9 ? hb - 9 : 0;
    auto digit = [&](uint32_t k, uint32_t id) -> unsigned {
      const unsigned long long c = ((unsigned long long)k << 32) | id;
      return (unsigned)(c >> shift) & (kBuckets - 1);
    };

    for (int i = lo + tid; i < hi; i += kTS) atomicAdd(&cnt[digit(sk[i], si[i])], 1u);
    __syncthreads();

    // exclusive scan of the bucket sizes (four consecutive buckets per thread); heavy buckets get their
    // stack slots here, deterministically -- past the stack's room they are finished by rank counting
    {
      uint32_t c4[kBuckets / kTS], sum = 0, heavy = 0;
#pragma unroll
      for (int k = 0; k < kBuckets / kTS; ++k) {
        c4[k] = cnt[tid * (kBuckets / kTS) + k];
        sum += c4[k];
        heavy += c4[k] > (uint32_t)kSmall ? 1u : 0u;
      }
      uint32_t tot, htot;
      uint32_t ex = block_scan_excl(sum, wave_sums, &tot);
      uint32_t hx = block_scan_excl(heavy, wave_sums, &htot);
      const int base = sn - 1;                     // stack height after the pop
#pragma unroll
      for (int k = 0; k < kBuckets / kTS; ++k) {
        const int d = tid * (kBuckets / kTS) + k;
        cur[d] = ex;
        if (c4[k] > (uint32_t)kSmall) {
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

    for (int i = lo + tid; i < hi; i += kTS) {
      const uint32_t k = sk[i], id = si[i];
      const uint32_t p = atomicAdd(&cur[digit(k, id)], 1u);
      dk[lo + p] = k;
      di[lo + p] = id;
    }
    __syncthreads();                               // the scattered segment is visible to the workgroup

    // cur[d] is now the END of bucket d: every element of a light bucket ranks itself inside it
    for (int i = lo + tid; i < hi; i += kTS) {
      const uint32_t k = dk[i], id = di[i];
      const unsigned d = digit(k, id);
      const uint32_t craw = cnt[d];
      const uint32_t b = craw & ~kBrute;
      if (b > (uint32_t)kSmall && !(craw & kBrute)) continue;     // on the stack
      const int be = lo + (int)cur[d], bs = be - (int)b;
      int c = 0;
      for (int j = bs; j < be; ++j) c += comp_less(dk[j], di[j], k, id) ? 1 : 0;
      ids_final[bs + c] = id;
    }
    __syncthreads();
  }
}

}  // namespace

size_t tile_depth_sort_temp_bytes(uint32_t capacity) {
  return 4 * align_up((size_t)(capacity ? capacity : 1) * sizeof(uint32_t), 256);
}

// Sorts flatten_ids[offsets[t] .. offsets[t+1]) of every tile by (depth bits, id).  temp:
// tile_depth_sort_temp_bytes(capacity) bytes.
int tile_depth_sort(int n_tiles, const int32_t* tile_offsets, const float* depths, uint32_t capacity,
                    uint32_t* flatten_ids, uint32_t* tile_ids_fill, void* temp, hipStream_t stream,
                    const uint32_t* staging, const int32_t* group_offsets, int group_shift) {
  if (n_tiles <= 0) return MGS_OK;
  const size_t stride = align_up((size_t)(capacity ? capacity : 1) * sizeof(uint32_t), 256) / sizeof(uint32_t);
  uint32_t* t = static_cast<uint32_t*>(temp);
  // average list length the capacity allows: short lists -> the fast path with the smaller LDS list
  const bool short_lists = (size_t)capacity <= (size_t)n_tiles * 640;
#define MGS_TS_LAUNCH(G, F, OFFS, STG, SH, OUT)                                                              \
  hipLaunchKernelGGL((tile_depth_sort_kernel<G, F>), dim3(n_tiles), dim3(kTS), 0, stream, n_tiles, OFFS,   \
                     depths, flatten_ids, tile_ids_fill, t, t + stride, t + 2 * stride, t + 3 * stride,    \
                     STG, SH, OUT)
  if (staging) {
    if (short_lists) MGS_TS_LAUNCH(true, kFastShort, group_offsets, staging, group_shift, const_cast<int32_t*>(tile_offsets));
    else MGS_TS_LAUNCH(true, kFastLong, group_offsets, staging, group_shift, const_cast<int32_t*>(tile_offsets));
  } else {
    if (short_lists) MGS_TS_LAUNCH(false, kFastShort, tile_offsets, (const uint32_t*)nullptr, 0, (int32_t*)nullptr);
    else MGS_TS_LAUNCH(false, kFastLong, tile_offsets, (const uint32_t*)nullptr, 0, (int32_t*)nullptr);
  }
#undef MGS_TS_LAUNCH
  return check_launch("tile_depth_sort");
}

}  // namespace mgs
