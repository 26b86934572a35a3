// calibration of rocprofv3 FETCH_SIZE for the raster's access pattern: every thread gathers one
// 48-byte record (three 16-byte loads) through an index; the records are read exactly once, so
// the true byte count is known: n * (48 + 4).  Variants: identity index (coalesced), random
// permutation (scattered records), permutation within 64-record windows (local scatter).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
__device__ __forceinline__ void body(int n, const int* __restrict__ idx, const float4* __restrict__ rec, float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int g = idx[i];
  float4 a = rec[3 * (size_t)g], b = rec[3 * (size_t)g + 1], c = rec[3 * (size_t)g + 2];
  out[i] = a.x + a.w + b.y + b.z + c.x + c.w;
}
__global__ void gather_identity(int n, const int* idx, const float4* rec, float* out) { body(n, idx, rec, out); }
__global__ void gather_random(int n, const int* idx, const float4* rec, float* out) { body(n, idx, rec, out); }
__global__ void gather_window64(int n, const int* idx, const float4* rec, float* out) { body(n, idx, rec, out); }
int main() {
  const int n = 6 * 1024 * 1024;                       // 288 MB of records (> the 256 MB Infinity Cache)
  std::vector<int> id(n);
  std::iota(id.begin(), id.end(), 0);
  int *d_idx; float4* d_rec; float* d_out;
  hipMalloc(&d_idx, n * 4); hipMalloc(&d_rec, (size_t)n * 48); hipMalloc(&d_out, n * 4);
  hipMemset(d_rec, 0, (size_t)n * 48);
  std::mt19937 rng(1);
  for (int variant = 0; variant < 3; ++variant) {
    std::iota(id.begin(), id.end(), 0);
    if (variant == 1) std::shuffle(id.begin(), id.end(), rng);
    if (variant == 2) for (int b = 0; b + 64 <= n; b += 64) std::shuffle(id.begin() + b, id.begin() + b + 64, rng);
    hipMemcpy(d_idx, id.data(), n * 4, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    if (variant == 0) gather_identity<<<n / 256, 256>>>(n, d_idx, d_rec, d_out);
    else if (variant == 1) gather_random<<<n / 256, 256>>>(n, d_idx, d_rec, d_out);
    else gather_window64<<<n / 256, 256>>>(n, d_idx, d_rec, d_out);
    hipDeviceSynchronize();
  }
  printf("true bytes read per launch: %.1f MB (records %.1f + index %.1f), written %.1f MB\n",
         n * 52.0 / 1e6, n * 48.0 / 1e6, n * 4.0 / 1e6, n * 4.0 / 1e6);
  return 0;
}
