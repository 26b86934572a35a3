// sh_staging.h -- moving 192-byte SH coefficient rows between HBM and lanes without a
// 192-byte-strided access.  A wave owns 64 consecutive rows (64 x 48 floats = 768 float4
// pieces).  Lane l of piece-round m handles piece f = 64 m + l, which belongs to row f / 12:
// the HBM side is lane-linear (1 KiB per instruction); rows whose bit in `wave_mask` is clear
// are neither fetched nor needed.  In LDS a row occupies 13 float4 slots (52-dword pitch): 13 is
// odd, so the 16 lanes of a ds_read_b128 service group land on 16 distinct 4-bank slots.
#ifndef MGS_SH_STAGING_H_
#define MGS_SH_STAGING_H_

#include <hip/hip_runtime.h>

namespace mgs {

constexpr int kShWave = 64;
constexpr int kShRowF4 = 12;      // 48 floats
constexpr int kShPitchF4 = 13;    // 52 dwords
constexpr int kShWaveSlots = kShWave * kShPitchF4;   // float4 slots per wave

// HBM -> this wave's LDS region.  g0 = first row of the wave; rows >= n do not exist.
__device__ __forceinline__ void sh_rows_to_lds(const float* __restrict__ coeffs, int g0, int n,
                                               unsigned long long wave_mask, float4* lds) {
  const unsigned lane = threadIdx.x & (kShWave - 1);
  const float4* src = reinterpret_cast<const float4*>(coeffs + (size_t)g0 * 48);
  float4 piece[kShRowF4];
#pragma unroll
  for (int m = 0; m < kShRowF4; ++m) {
    unsigned f = m * kShWave + lane, owner = f / kShRowF4;
    bool need = ((wave_mask >> owner) & 1ull) && (g0 + (int)owner < n);
    piece[m] = need ? src[f] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int m = 0; m < kShRowF4; ++m) {
    unsigned f = m * kShWave + lane, owner = f / kShRowF4;
    lds[owner * kShPitchF4 + (f - owner * kShRowF4)] = piece[m];
  }
}

// this wave's LDS region -> HBM, lane-linear; ACCUM adds to what is stored there
template <bool ACCUM>
__device__ __forceinline__ void sh_rows_from_lds(float* __restrict__ dst_base, int g0, int n,
                                                 const float4* lds) {
  const unsigned lane = threadIdx.x & (kShWave - 1);
  float4* dst = reinterpret_cast<float4*>(dst_base + (size_t)g0 * 48);
#pragma unroll
  for (int m = 0; m < kShRowF4; ++m) {
    unsigned f = m * kShWave + lane, owner = f / kShRowF4;
    if (g0 + (int)owner < n) {
      float4 v = lds[owner * kShPitchF4 + (f - owner * kShRowF4)];
      if (ACCUM) {
        float4 o = dst[f];
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
      dst[f] = v;
    }
  }
}

}  // namespace mgs
#endif
