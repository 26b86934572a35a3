// loss.hip -- the photometric L1 term of the training step (BASELINE configs[2]: "L1 loss to
// random target"), forward and backward as two streaming HIP kernels instead of the six
// elementwise / reduction launches the same expression costs in eager PyTorch (72 -> ~20 us at
// 1920x1080x3).  HBM-bound: 8 B read per element forward, 8 B read + 4 B written backward.
// The sum is taken in a fixed order (per-thread strided partials, wave butterfly, per-block slots,
// one final block), so the loss is bit-reproducible.
#include <algorithm>

#include "mgs_common.h"

namespace mgs {
namespace {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 1024;

__device__ __forceinline__ float block_sum(float v, float* lds) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  return lds[0] + lds[1] + lds[2] + lds[3];
}

__global__ __launch_bounds__(kBlock) void l1_partial_kernel(size_t n, const float* __restrict__ a,
                                                            const float* __restrict__ b,
                                                            float* __restrict__ partial) {
  __shared__ float lds[kBlock / 64];
  float acc = 0.f;
  const size_t n4 = n / 4, stride = (size_t)gridDim.x * kBlock;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
    const float4 x = a4[i], y = b4[i];
    acc += (fabsf(x.x - y.x) + fabsf(x.y - y.y)) + (fabsf(x.z - y.z) + fabsf(x.w - y.w));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) acc += fabsf(a[4 * n4 + threadIdx.x] - b[4 * n4 + threadIdx.x]);
  const float s = block_sum(acc, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(kBlock) void l1_final_kernel(int n_partial, const float* __restrict__ partial,
                                                          float inv_n, float* __restrict__ loss) {
  __shared__ float lds[kBlock / 64];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n_partial; i += kBlock) acc += partial[i];
  const float s = block_sum(acc, lds);
  if (threadIdx.x == 0) *loss = s * inv_n;
}

__global__ __launch_bounds__(kBlock) void l1_bwd_kernel(size_t n, const float* __restrict__ a,
                                                        const float* __restrict__ b,
                                                        const float* __restrict__ v_loss, float inv_n,
                                                        float* __restrict__ v_a) {
  const float g = (v_loss ? *v_loss : 1.f) * inv_n;
  const size_t n4 = n / 4, stride = (size_t)gridDim.x * kBlock;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  float4* o4 = reinterpret_cast<float4*>(v_a);
  auto sg = [g](float d) { return d > 0.f ? g : (d < 0.f ? -g : 0.f); };   // torch.sign: 0 at 0
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
    const float4 x = a4[i], y = b4[i];
    o4[i] = make_float4(sg(x.x - y.x), sg(x.y - y.y), sg(x.z - y.z), sg(x.w - y.w));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = 4 * n4 + threadIdx.x;
    v_a[i] = sg(a[i] - b[i]);
  }
}

// Forward AND the gradient for v_loss = 1 in one pass over a and b: v_a = sign(a - b) / n beside the partial sums (12
// instead of 8 + 12 bytes per element for the pair of kernels above); l1_final_kernel adds the partials up as before.
// (Letting the block that finishes last do that -- __threadfence + a counter -- was measured: 51 us instead of ~18.  An
// agent-scope release on gfx950 writes the XCD's dirty L2 lines back, and here those are the 33 MB of v_a just stored.)
__global__ __launch_bounds__(kBlock) void l1_fwd_grad_kernel(size_t n, const float* __restrict__ a,
                                                             const float* __restrict__ b, float inv_n,
                                                             float* __restrict__ partial, float* __restrict__ v_a) {
  __shared__ float lds[kBlock / 64];
  float acc = 0.f;
  const size_t n4 = n / 4, stride = (size_t)gridDim.x * kBlock;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  float4* o4 = reinterpret_cast<float4*>(v_a);
  const float g = inv_n;
  auto sg = [g](float d) { return d > 0.f ? g : (d < 0.f ? -g : 0.f); };
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
    const float4 x = a4[i], y = b4[i];
    const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
    acc += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
    o4[i] = make_float4(sg(d0), sg(d1), sg(d2), sg(d3));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = 4 * n4 + threadIdx.x;
    const float d = a[i] - b[i];
    acc += fabsf(d);
    v_a[i] = sg(d);
  }
  const float s = block_sum(acc, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// v_a <- sign(v_a) * v_loss / n, in place (v_a as l1_fwd_grad_kernel left it, or this kernel).  v_loss == 1 -- the
// usual loss.backward() -- leaves at once: nothing to do, which is why the launch is kScaleBlocks workgroups only.
__global__ __launch_bounds__(kBlock) void l1_scale_kernel(size_t n, const float* __restrict__ v_loss, float inv_n,
                                                          float* __restrict__ v_a) {
  const float vl = *v_loss;
  if (vl == 1.f) return;
  const float g = vl * inv_n;
  const size_t n4 = n / 4, stride = (size_t)gridDim.x * kBlock;
  float4* o4 = reinterpret_cast<float4*>(v_a);
  auto sg = [g](float d) { return d > 0.f ? g : (d < 0.f ? -g : 0.f); };
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
    const float4 x = o4[i];
    o4[i] = make_float4(sg(x.x), sg(x.y), sg(x.z), sg(x.w));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = 4 * n4 + threadIdx.x;
    v_a[i] = sg(v_a[i]);
  }
}

constexpr unsigned kScaleBlocks = 128;

unsigned grid_for(size_t n) {
  size_t blocks = (n / 4 + kBlock - 1) / kBlock;
  return (unsigned)(blocks < 1 ? 1 : (blocks > kMaxBlocks ? kMaxBlocks : blocks));
}

}  // namespace
}  // namespace mgs

using namespace mgs;

extern "C" int mgs_l1_loss_fwd(size_t n, const float* a, const float* b, float* loss,
                               void* workspace, size_t* workspace_bytes, mgs_stream_t stream) {
  MGS_REQUIRE(workspace_bytes, "l1_loss_fwd: workspace_bytes is null");
  const size_t need = kMaxBlocks * sizeof(float);
  if (!workspace) {
    *workspace_bytes = need;
    return MGS_OK;
  }
  if (*workspace_bytes < need)
    return set_error(MGS_ERR_WORKSPACE_TOO_SMALL, "l1_loss_fwd: workspace %zu < %zu bytes",
                     *workspace_bytes, need);
  MGS_REQUIRE(n > 0 && a && b && loss, "l1_loss_fwd: empty input or null pointer");
  MGS_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0, "l1_loss_fwd: inputs must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  float* partial = static_cast<float*>(workspace);
  const unsigned grid = grid_for(n);
  hipLaunchKernelGGL(l1_partial_kernel, dim3(grid), dim3(kBlock), 0, s, n, a, b, partial);
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(kBlock), 0, s, (int)grid, partial,
                     (float)(1.0 / (double)n), loss);
  return check_launch("l1_loss_fwd");
}

extern "C" int mgs_l1_loss_bwd(size_t n, const float* a, const float* b, const float* v_loss,
                               float* v_a, mgs_stream_t stream) {
  MGS_REQUIRE(n > 0 && a && b && v_a, "l1_loss_bwd: empty input or null pointer");
  MGS_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0 && ((uintptr_t)v_a & 15) == 0,
              "l1_loss_bwd: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, a, b,
                     v_loss, (float)(1.0 / (double)n), v_a);
  return check_launch("l1_loss_bwd");
}

extern "C" int mgs_l1_loss_fwd_grad(size_t n, const float* a, const float* b, float* loss, float* v_a,
                                    void* workspace, size_t* workspace_bytes, mgs_stream_t stream) {
  MGS_REQUIRE(workspace_bytes, "l1_loss_fwd_grad: workspace_bytes is null");
  const size_t need = kMaxBlocks * sizeof(float);
  if (!workspace) {
    *workspace_bytes = need;
    return MGS_OK;
  }
  if (*workspace_bytes < need)
    return set_error(MGS_ERR_WORKSPACE_TOO_SMALL, "l1_loss_fwd_grad: workspace %zu < %zu bytes",
                     *workspace_bytes, need);
  MGS_REQUIRE(n > 0 && a && b && loss && v_a, "l1_loss_fwd_grad: empty input or null pointer");
  MGS_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0 && ((uintptr_t)v_a & 15) == 0,
              "l1_loss_fwd_grad: buffers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  float* partial = static_cast<float*>(workspace);
  const unsigned grid = grid_for(n);
  hipLaunchKernelGGL(l1_fwd_grad_kernel, dim3(grid), dim3(kBlock), 0, s, n, a, b, (float)(1.0 / (double)n), partial, v_a);
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(kBlock), 0, s, (int)grid, partial, (float)(1.0 / (double)n), loss);
  return check_launch("l1_loss_fwd_grad");
}

extern "C" int mgs_l1_loss_bwd_scale(size_t n, const float* v_loss, float* v_a, mgs_stream_t stream) {
  MGS_REQUIRE(n > 0 && v_loss && v_a, "l1_loss_bwd_scale: empty input or null pointer");
  MGS_REQUIRE(((uintptr_t)v_a & 15) == 0, "l1_loss_bwd_scale: v_a must be 16-byte aligned");
  hipLaunchKernelGGL(l1_scale_kernel, dim3(std::min(grid_for(n), kScaleBlocks)), dim3(kBlock), 0, (hipStream_t)stream, n, v_loss,
                     (float)(1.0 / (double)n), v_a);
  return check_launch("l1_loss_bwd_scale");
}
