// micro-benchmark: cost of a dependent kernel dispatch inside a HIP graph on gfx950, with 1..4
// graphs replayed concurrently on separate streams.  Each graph is a chain of `len` tiny kernels
// (stream capture, so every node depends on its predecessor, as in the render frame).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void tiny(float* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += 1.0f;
}
int main() {
  const int len = 25;
  float* buf[4];
  hipStream_t st[4];
  hipGraphExec_t ex[4];
  for (int s = 0; s < 4; ++s) {
    hipMalloc(&buf[s], 1 << 22);
    hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking);
  }
  for (int blocks : {1, 1000}) {
    for (int s = 0; s < 4; ++s) {
      hipGraph_t g;
      hipStreamBeginCapture(st[s], hipStreamCaptureModeGlobal);
      for (int k = 0; k < len; ++k) tiny<<<blocks, 256, 0, st[s]>>>(buf[s], blocks * 256);
      hipStreamEndCapture(st[s], &g);
      hipGraphInstantiate(&ex[s], g, nullptr, nullptr, 0);
    }
    for (int ns = 1; ns <= 4; ++ns) {
      const int iters = 200;
      for (int w = 0; w < 10; ++w) for (int s = 0; s < ns; ++s) hipGraphLaunch(ex[s], st[s]);
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < iters; ++i) for (int s = 0; s < ns; ++s) hipGraphLaunch(ex[s], st[s]);
      hipDeviceSynchronize();
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("%4d blocks/kernel, %d concurrent graphs of %d chained kernels: %.2f us per graph, %.2f us per kernel (aggregate)\n",
             blocks, ns, len, us / (iters * ns), us / (iters * ns * len));
    }
  }
  return 0;
}
