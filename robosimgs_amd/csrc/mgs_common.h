// mgs_common.h -- host-side plumbing shared by the libmgs.so translation units.
#ifndef MGS_COMMON_H_
#define MGS_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mgs.h"

namespace mgs {

// thread-local last-error text behind mgs_last_error_string()
char* error_buffer();
int set_error(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return set_error((int)e, "%s: launch failed: %s", what, hipGetErrorString(e));
  return MGS_OK;
}

inline unsigned div_up(unsigned a, unsigned b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#define MGS_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) return ::mgs::set_error(MGS_ERR_INVALID_ARGUMENT, __VA_ARGS__); \
  } while (0)

// ---- wave64 helpers -----------------------------------------------------------------
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// wave64 ballot of a predicate: the lane mask the compare already produced.  (HIP's __ballot(int) turns the predicate
// into 0 / 1 and compares that with zero again: a v_cndmask and a v_cmp per call that the compiler does not always fold
// away -- two of the per-tile sort's group filter's instructions per list entry were exactly these.)
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ unsigned mask_rank(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
// wave-wide OR / AND / sum of one 32-bit value per lane, returned wave-uniform.  Six DPP steps (quad_perm x2,
// row_half_mirror, row_mirror, row_bcast15, row_bcast31: the total lands in lane 63) and a readlane, instead of six
// ds_bpermute round trips with their lane-index arithmetic (__shfl_xor).
#define MGS_DPP_STEP(OP, IDENT, CTRL, RMASK) \
  v = v OP (uint32_t)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, CTRL, RMASK, 0xf, false);
#define MGS_WAVE_REDUCE(NAME, OP, IDENT)                                         \
  __device__ __forceinline__ uint32_t NAME(uint32_t v) {                         \
    MGS_DPP_STEP(OP, IDENT, 0xB1, 0xf)                                           \
    MGS_DPP_STEP(OP, IDENT, 0x4E, 0xf)                                           \
    MGS_DPP_STEP(OP, IDENT, 0x141, 0xf)                                          \
    MGS_DPP_STEP(OP, IDENT, 0x140, 0xf)                                          \
    MGS_DPP_STEP(OP, IDENT, 0x142, 0xa)                                          \
    MGS_DPP_STEP(OP, IDENT, 0x143, 0xc)                                          \
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);                      \
  }
MGS_WAVE_REDUCE(wave_or, |, 0u)
MGS_WAVE_REDUCE(wave_and, &, ~0u)
MGS_WAVE_REDUCE(wave_sum_u32, +, 0u)
#undef MGS_WAVE_REDUCE
#undef MGS_DPP_STEP
#endif

// ---- internal launchers shared between translation units ---------------------------------
// Radix sort of (key, value) uint32 pairs on bits [0, key_bits) with the element count read
// from device memory.  Buffers a/b alternate; the sorted result ends in (keys_out, vals_out).
// temp: radix_sort_temp_bytes(capacity).
size_t radix_sort_temp_bytes(uint32_t capacity);
int radix_sort_pairs(const uint32_t* n_dev, uint32_t capacity, int key_bits, uint32_t* keys_in,
                     uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, void* temp,
                     hipStream_t stream);

// Orders flatten_ids[offsets[t] .. offsets[t+1]) of every tile by (depth bits, Gaussian id)
// (tile_sort.hip).  temp: tile_depth_sort_temp_bytes(capacity).
size_t tile_depth_sort_temp_bytes(uint32_t capacity, int n_tiles);
// first word of the list of long tiles inside `temp` (a header of eight counters, 16-byte aligned: they must be zero when
// tile_depth_sort starts)
uint32_t* tile_depth_sort_long_list(void* temp, uint32_t capacity);
// tile_ids_fill (may be null): also stores every entry's tile id (the direct binning path has not written them).
// staging / group_offsets / group_shift (staging may be null): the lists arrive grouped, 2^group_shift tiles per
// segment of `staging`, and the kernel also WRITES tile_offsets (see tile_sort.hip).
// long_list_zeroed: an earlier kernel of the caller has stored the zeros (otherwise a 32-byte memset is enqueued).
// scratch_k / scratch_i: two more arrays of `capacity` words that nobody reads once the per-tile sort's first launch is
// done (scratch_i may be `staging`): scratch of the kernel that sorts the units of long lists; null: no list is deferred.
int tile_depth_sort(int n_tiles, const int32_t* tile_offsets, const float* depths, uint32_t capacity,
                    uint32_t* flatten_ids, uint32_t* tile_ids_fill, void* temp, hipStream_t stream,
                    uint32_t* scratch_k, uint32_t* scratch_i,
                    const uint32_t* staging = nullptr, const int32_t* group_offsets = nullptr, int group_shift = 0,
                    bool long_list_zeroed = false);

// mgs_debug_set_sort_opts() bits (sort.hip): 1 one-sweep radix passes, 2 ... also for small inputs,
// 4 binning partitions by tile with the radix passes even where the direct path applies
int sort_opts();

}  // namespace mgs
#endif
