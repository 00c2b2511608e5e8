// Micro-benchmark: fp32 MFMA shape vs sustained rate on random data (MI355X_MICROARCH.md, DVFS give-back item 7).
// Both loops read their A operand from LDS (as the colour-head kernels do) and keep B in registers.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int SHAPE>   // 0: 32x32x2, 1: 16x16x4
__global__ void __launch_bounds__(512) k(const float* __restrict__ w, const float* __restrict__ x, float* __restrict__ out, int iters) {
  __shared__ float lw[64 * 256];          // 64 KB of "weights"
  for (int i = threadIdx.x; i < 64 * 256; i += blockDim.x) lw[i] = w[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float b[16];
  for (int i = 0; i < 16; ++i) b[i] = x[(blockIdx.x * 512 + threadIdx.x) * 16 + i];
  if (SHAPE == 0) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(lw[((t * 16 + r) * 64 + lane + it * 64) & 16383], b[r], acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    f32x4 acc[16];
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 8; ++r)       // 16 tiles x 8 k-steps x 2048 flop = the same 64 x 4096 flop per iteration
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(lw[((t * 8 + r) * 64 + lane + it * 64) & 16383], b[(r + t) & 15], acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

int main() {
  const int blocks = 256, iters = 4000;
  std::vector<float> hw(64 * 256), hx(blocks * 512 * 16);
  for (auto& v : hw) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
  float *w, *x, *o;
  hipMalloc(&w, hw.size() * 4); hipMalloc(&x, hx.size() * 4); hipMalloc(&o, blocks * 512 * 4);
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep)
    for (int shape = 0; shape < 2; ++shape) {
      for (int warm = 0; warm < 2; ++warm) {
        if (shape == 0) k<0><<<blocks, 512>>>(w, x, o, iters); else k<1><<<blocks, 512>>>(w, x, o, iters);
      }
      hipEventRecord(a);
      for (int n = 0; n < 5; ++n) {
        if (shape == 0) k<0><<<blocks, 512>>>(w, x, o, iters); else k<1><<<blocks, 512>>>(w, x, o, iters);
      }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
      const double flop = (double)blocks * 8 * iters * 64 * 4096.0;
      printf("%s: %.3f ms  %.1f TFLOP/s\n", shape == 0 ? "32x32x2" : "16x16x4", ms, flop / ms / 1e9);
    }
  return 0;
}
