// micro-benchmark: per-tile counting / cursor atomics with 3DGS-like contention
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
__global__ void count_k(int n, const int4* rects, int tw, unsigned* cnt) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  int4 r = rects[g];
  for (int y = r.y; y < r.w; ++y) for (int x = r.x; x < r.z; ++x) atomicAdd(&cnt[y * tw + x], 1u);
}
__global__ void fill_k(int n, const int4* rects, int tw, unsigned* cur, const unsigned* off, unsigned long long* out) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  int4 r = rects[g];
  for (int y = r.y; y < r.w; ++y) for (int x = r.x; x < r.z; ++x) {
    int t = y * tw + x;
    unsigned p = atomicAdd(&cur[t], 1u);
    out[off[t] + p] = ((unsigned long long)g << 32) | g;
  }
}
__global__ void diff_k(int n, const int4* rects, int tw1, int* d) {   // 2D difference trick: 4 atomics per Gaussian
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  int4 r = rects[g];
  atomicAdd(&d[r.y * tw1 + r.x], 1); atomicAdd(&d[r.y * tw1 + r.z], -1);
  atomicAdd(&d[r.w * tw1 + r.x], -1); atomicAdd(&d[r.w * tw1 + r.z], 1);
}
int main() {
  const int n = 765000, tw = 120, th = 68, nt = tw * th;
  std::mt19937 rng(1);
  std::vector<int4> rects(n);
  std::vector<unsigned> cnt(nt, 0);
  size_t total = 0;
  for (auto& r : rects) {
    int w = 1 + rng() % 4, h = 1 + rng() % 4;
    int x0 = rng() % (tw - w + 1), y0 = rng() % (th - h + 1);
    r = make_int4(x0, y0, x0 + w, y0 + h);
    for (int y = y0; y < y0 + h; ++y) for (int x = x0; x < x0 + w; ++x) cnt[y * tw + x]++;
    total += w * h;
  }
  std::vector<unsigned> off(nt + 1, 0);
  for (int t = 0; t < nt; ++t) off[t + 1] = off[t] + cnt[t];
  int4* d_r; unsigned *d_c, *d_o; unsigned long long* d_out; int* d_d;
  hipMalloc(&d_r, n * sizeof(int4)); hipMalloc(&d_c, nt * 4); hipMalloc(&d_o, (nt + 1) * 4);
  hipMalloc(&d_out, total * 8); hipMalloc(&d_d, (tw + 1) * (th + 1) * 4);
  hipMemcpy(d_r, rects.data(), n * sizeof(int4), hipMemcpyHostToDevice);
  hipMemcpy(d_o, off.data(), (nt + 1) * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto fn) {
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
      hipMemset(d_c, 0, nt * 4); hipMemset(d_d, 0, (tw + 1) * (th + 1) * 4);
      hipDeviceSynchronize();
      hipEventRecord(e0); fn(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    printf("%-10s %.1f us  (pairs %zu)\n", name, best * 1e3, total);
  };
  dim3 grid((n + 255) / 256), block(256);
  timeit("count", [&] { count_k<<<grid, block>>>(n, d_r, tw, d_c); });
  timeit("diff2d", [&] { diff_k<<<grid, block>>>(n, d_r, tw + 1, d_d); });
  timeit("fill", [&] { fill_k<<<grid, block>>>(n, d_r, tw, d_c, d_o, d_out); });
  return 0;
}
