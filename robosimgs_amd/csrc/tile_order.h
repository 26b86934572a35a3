// tile_order.h -- launch order of the raster kernels' tiles: groups of four consecutive tiles by falling total list
// length.  A tile (or an 8x8 block of it) is one wave's serial job and the hardware starts workgroups in index
// order, so a long list that starts late sets the kernel's end while most of the chip idles; with the long
// lists first the short ones fill the end of the launch (longest-processing-time-first).  Groups of four
// neighbours rather than single tiles: neighbouring tiles share most of their Gaussians, and keeping them together
// keeps those records in the CU's cache (raster_fwd_q_kernel at config 2: 190.9 us in index order, 185.9 by tile
// length, 175.2 by group total).  The order changes the schedule, never a result.
#ifndef MGS_TILE_ORDER_H_
#define MGS_TILE_ORDER_H_

#include "mgs_common.h"

namespace mgs {

constexpr int kOrderClasses = 1024;      // length classes of the counting sort

// One workgroup of T threads (T = 512 or 1024): order[0 .. n_groups) = the groups by falling total(g); the
// order inside a class is whatever the LDS atomics give.  total(g) is evaluated three times per group.
template <int T, class Total>
__device__ __forceinline__ void order_groups_by_total(int n_groups, Total total, int32_t* __restrict__ order) {
  __shared__ uint32_t hist[kOrderClasses];
  __shared__ uint32_t wave_tot[T / 64];
  __shared__ uint32_t longest;
  constexpr int kPer = kOrderClasses / T;           // consecutive classes per thread in the scan
  constexpr int kBatch = 8;                         // totals in flight per thread
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < kPer; ++k) hist[t * kPer + k] = 0u;
  if (t == 0) longest = 1u;
  __syncthreads();
  uint32_t m = 1u;
  for (int g0 = 0; g0 < n_groups; g0 += T * kBatch) {
    uint32_t v[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) v[j] = g0 + j * T + t < n_groups ? total(g0 + j * T + t) : 0u;
#pragma unroll
    for (int j = 0; j < kBatch; ++j) m = max(m, v[j]);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
  if ((t & 63) == 0) atomicMax(&longest, m);
  __syncthreads();
  const float scale = (float)(kOrderClasses - 1) / (float)longest;
  auto cls = [&](uint32_t len) {
    return kOrderClasses - 1 - min(kOrderClasses - 1, (int)((float)len * scale));       // longest first
  };
  for (int g0 = 0; g0 < n_groups; g0 += T * kBatch) {
    uint32_t v[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) v[j] = g0 + j * T + t < n_groups ? total(g0 + j * T + t) : 0u;
#pragma unroll
    for (int j = 0; j < kBatch; ++j)
      if (g0 + j * T + t < n_groups) atomicAdd(&hist[cls(v[j])], 1u);
  }
  __syncthreads();
  // exclusive scan of the class counts
  uint32_t c[kPer], sum = 0;
#pragma unroll
  for (int k = 0; k < kPer; ++k) { c[k] = hist[t * kPer + k]; sum += c[k]; }
  uint32_t incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d);
    if ((t & 63) >= d) incl += up;
  }
  if ((t & 63) == 63) wave_tot[t >> 6] = incl;
  __syncthreads();
  uint32_t ex = incl - sum;
  for (int w = 0; w < (t >> 6); ++w) ex += wave_tot[w];
#pragma unroll
  for (int k = 0; k < kPer; ++k) { hist[t * kPer + k] = ex; ex += c[k]; }
  __syncthreads();
  for (int g0 = 0; g0 < n_groups; g0 += T * kBatch) {
    uint32_t v[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) v[j] = g0 + j * T + t < n_groups ? total(g0 + j * T + t) : 0u;
#pragma unroll
    for (int j = 0; j < kBatch; ++j)
      if (g0 + j * T + t < n_groups) order[atomicAdd(&hist[cls(v[j])], 1u)] = g0 + j * T + t;
  }
}

// Which tile a raster wave works on: unit u of the launch (tile slots in launch order) -> tile index, or -1 for
// the padding slots of the last group.  group_order == nullptr: index order.
__device__ __forceinline__ int tile_of_unit(int unit, int n_tiles, const int32_t* __restrict__ group_order) {
  if (!group_order) return unit < n_tiles ? unit : -1;
  if (unit >= ((n_tiles + 3) & ~3)) return -1;
  const int tile = group_order[unit >> 2] * 4 + (unit & 3);
  return tile < n_tiles ? tile : -1;
}

// Standalone: the order from finished tile offsets (radix path, or a caller that only has the lists).
int launch_tile_group_order(int n_tiles, const int32_t* tile_offsets, int32_t* order, hipStream_t stream);

}  // namespace mgs
#endif
