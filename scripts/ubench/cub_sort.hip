// reference point only: how fast does rocPRIM's tuned radix sort do the two sorts of the binning?
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <vector>
#include <random>
int main() {
  struct Case { size_t n; int bits; const char* name; } cases[] = {{1000000, 32, "depth sort 1M x 32 bits"}, {5019684, 13, "tile sort 5M x 13 bits"}, {5019684, 45, "textbook 5M x 45 bits (64-bit keys)"}};
  for (auto c : cases) {
    std::mt19937_64 rng(1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    if (c.bits <= 32) {
      std::vector<uint32_t> k(c.n), v(c.n);
      for (size_t i = 0; i < c.n; ++i) { k[i] = (uint32_t)(rng() & ((c.bits == 32) ? 0xffffffffull : ((1ull << c.bits) - 1))); v[i] = (uint32_t)i; }
      uint32_t *dk, *dv, *ok, *ov; hipMalloc(&dk, c.n * 4); hipMalloc(&dv, c.n * 4); hipMalloc(&ok, c.n * 4); hipMalloc(&ov, c.n * 4);
      hipMemcpy(dk, k.data(), c.n * 4, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), c.n * 4, hipMemcpyHostToDevice);
      size_t tb = 0; hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, ok, dv, ov, (int)c.n, 0, c.bits);
      void* tmp; hipMalloc(&tmp, tb);
      for (int it = 0; it < 10; ++it) {
        hipEventRecord(e0); hipcub::DeviceRadixSort::SortPairs(tmp, tb, dk, ok, dv, ov, (int)c.n, 0, c.bits); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
      }
    } else {
      std::vector<uint64_t> k(c.n); std::vector<uint32_t> v(c.n);
      for (size_t i = 0; i < c.n; ++i) { k[i] = rng() & ((1ull << c.bits) - 1); v[i] = (uint32_t)i; }
      uint64_t *dk, *ok; uint32_t *dv, *ov; hipMalloc(&dk, c.n * 8); hipMalloc(&ok, c.n * 8); hipMalloc(&dv, c.n * 4); hipMalloc(&ov, c.n * 4);
      hipMemcpy(dk, k.data(), c.n * 8, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), c.n * 4, hipMemcpyHostToDevice);
      size_t tb = 0; hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, ok, dv, ov, (int)c.n, 0, c.bits);
      void* tmp; hipMalloc(&tmp, tb);
      for (int it = 0; it < 10; ++it) {
        hipEventRecord(e0); hipcub::DeviceRadixSort::SortPairs(tmp, tb, dk, ok, dv, ov, (int)c.n, 0, c.bits); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
      }
    }
    printf("%-40s %.1f us\n", c.name, best * 1e3);
  }
  return 0;
}
