// atomic_rate.hip -- how fast are integer atomics on a few thousand hot counters (a tile histogram / per-tile
// cursors), from all XCDs at once?  hipcc --offload-arch=gfx950 -O3 atomic_rate.hip -o atomic_rate && ./atomic_rate
// Each thread stands for one Gaussian covering a w x h rectangle of a 120 x 68 tile grid (8160 counters), drawn
// so that the total is ~3.7 M (tile, Gaussian) pairs like configs[1].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void count_kernel(int n, const uint32_t* __restrict__ rect, unsigned* __restrict__ counters) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  uint32_t r = rect[g];
  int x0 = r & 0xff, y0 = (r >> 8) & 0xff, w = (r >> 16) & 0xff, h = r >> 24;
  for (int y = y0; y < y0 + h; ++y)
    for (int x = x0; x < x0 + w; ++x) atomicAdd(&counters[y * 120 + x], 1u);          // result unused: no return
}
__global__ void cursor_kernel(int n, const uint32_t* __restrict__ rect, unsigned* __restrict__ cursors,
                              const unsigned* __restrict__ base, uint2* __restrict__ out) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  uint32_t r = rect[g];
  int x0 = r & 0xff, y0 = (r >> 8) & 0xff, w = (r >> 16) & 0xff, h = r >> 24;
  for (int y = y0; y < y0 + h; ++y)
    for (int x = x0; x < x0 + w; ++x) {
      int t = y * 120 + x;
      unsigned pos = base[t] + atomicAdd(&cursors[t], 1u);                             // returning
      out[pos] = make_uint2((unsigned)g, 0x3f800000u);
    }
}
// the same work, but a wave first combines equal tiles of its 64 lanes?  (not done: neighbours are unrelated)

int main() {
  const int n = 765000;
  std::mt19937 rng(1);
  std::vector<uint32_t> rect(n);
  std::vector<unsigned> cnt(8160, 0);
  size_t total = 0;
  for (int i = 0; i < n; ++i) {
    int w = 1 + (rng() % 100 < 60 ? rng() % 2 : rng() % 4), h = 1 + (rng() % 100 < 60 ? rng() % 2 : rng() % 4);
    int x0 = rng() % (120 - w + 1), y0 = rng() % (68 - h + 1);
    rect[i] = x0 | (y0 << 8) | (w << 16) | (h << 24);
    total += (size_t)w * h;
    for (int y = y0; y < y0 + h; ++y) for (int x = x0; x < x0 + w; ++x) cnt[y * 120 + x]++;
  }
  std::vector<unsigned> base(8160);
  unsigned run = 0;
  for (int t = 0; t < 8160; ++t) { base[t] = run; run += cnt[t]; }
  printf("%d Gaussians, %zu pairs\n", n, total);
  uint32_t* d_rect; unsigned *d_cnt, *d_base; uint2* d_out;
  CK(hipMalloc(&d_rect, n * 4)); CK(hipMalloc(&d_cnt, 8160 * 4)); CK(hipMalloc(&d_base, 8160 * 4)); CK(hipMalloc(&d_out, total * 8));
  CK(hipMemcpy(d_rect, rect.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_base, base.data(), 8160 * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int which = 0; which < 2; ++which)
    for (int bs : {64, 256}) {
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(d_cnt, 0, 8160 * 4));
        CK(hipEventRecord(e0));
        if (which == 0) hipLaunchKernelGGL(count_kernel, dim3((n + bs - 1) / bs), dim3(bs), 0, 0, n, d_rect, d_cnt);
        else hipLaunchKernelGGL(cursor_kernel, dim3((n + bs - 1) / bs), dim3(bs), 0, 0, n, d_rect, d_cnt, d_base, d_out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("%s block %3d: %.1f us  (%.1f G atomics/s)\n", which ? "cursor + 8-byte scatter" : "count (no return)     ", bs,
             best * 1e3, total / (best * 1e-3) * 1e-9);
    }
  std::vector<unsigned> got(8160);
  CK(hipMemcpy(got.data(), d_cnt, 8160 * 4, hipMemcpyDeviceToHost));
  int bad = 0; for (int t = 0; t < 8160; ++t) bad += got[t] != cnt[t];
  printf("counters wrong: %d\n", bad);
  return 0;
}
