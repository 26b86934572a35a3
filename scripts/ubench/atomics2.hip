// micro-benchmark: XCD-private copies + workgroup-scope (L2-executed) atomics vs agent scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
template <int SCOPE, bool FLOAT>
__global__ void count_k(int n, const int4* rects, int tw, int nt, void* cnt) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  unsigned x = xcc_id();
  int4 r = rects[g];
  for (int y = r.y; y < r.w; ++y) for (int xx = r.x; xx < r.z; ++xx) {
    if (FLOAT) __hip_atomic_fetch_add((float*)cnt + x * nt + y * tw + xx, 1.0f, __ATOMIC_RELAXED, SCOPE);
    else __hip_atomic_fetch_add((unsigned*)cnt + x * nt + y * tw + xx, 1u, __ATOMIC_RELAXED, SCOPE);
  }
}
__global__ void xcc_hist(unsigned* h) { if (threadIdx.x == 0) atomicAdd(&h[xcc_id()], 1u); }
int main() {
  const int n = 765000, tw = 120, th = 68, nt = tw * th;
  std::mt19937 rng(1);
  std::vector<int4> rects(n);
  size_t total = 0;
  for (auto& r : rects) {
    int w = 1 + rng() % 4, h = 1 + rng() % 4;
    int x0 = rng() % (tw - w + 1), y0 = rng() % (th - h + 1);
    r = make_int4(x0, y0, x0 + w, y0 + h);
    total += w * h;
  }
  int4* d_r; unsigned* d_c; unsigned* d_h;
  hipMalloc(&d_r, n * sizeof(int4)); hipMalloc(&d_c, 16 * nt * 4); hipMalloc(&d_h, 64);
  hipMemcpy(d_r, rects.data(), n * sizeof(int4), hipMemcpyHostToDevice);
  hipMemset(d_h, 0, 64);
  xcc_hist<<<1024, 64>>>(d_h);
  unsigned hh[16]; hipMemcpy(hh, d_h, 64, hipMemcpyDeviceToHost);
  printf("xcc histogram of 1024 blocks:"); for (int i = 0; i < 16; ++i) printf(" %u", hh[i]); printf("\n");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto fn) {
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
      hipMemset(d_c, 0, 16 * nt * 4);
      hipDeviceSynchronize();
      hipEventRecord(e0); fn(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    // verify: sum over copies
    std::vector<unsigned> c(16 * nt); hipMemcpy(c.data(), d_c, 16 * nt * 4, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 16 * nt; ++i) s += name[0] == 'f' ? (double)((float*)c.data())[i] : (double)c[i];
    printf("%-22s %.1f us  (pairs %zu, sum of copies %.0f)\n", name, best * 1e3, total, s);
  };
  dim3 grid((n + 255) / 256), block(256);
  timeit("u32 agent", [&] { count_k<__HIP_MEMORY_SCOPE_AGENT, false><<<grid, block>>>(n, d_r, tw, nt, d_c); });
  timeit("u32 workgroup", [&] { count_k<__HIP_MEMORY_SCOPE_WORKGROUP, false><<<grid, block>>>(n, d_r, tw, nt, d_c); });
  timeit("u32 wavefront", [&] { count_k<__HIP_MEMORY_SCOPE_WAVEFRONT, false><<<grid, block>>>(n, d_r, tw, nt, d_c); });
  timeit("f32 agent", [&] { count_k<__HIP_MEMORY_SCOPE_AGENT, true><<<grid, block>>>(n, d_r, tw, nt, d_c); });
  timeit("f32 workgroup", [&] { count_k<__HIP_MEMORY_SCOPE_WORKGROUP, true><<<grid, block>>>(n, d_r, tw, nt, d_c); });
  return 0;
}
