// Micro-benchmark: the 128 x 128 layer of the colour head as a 3-way bf16 split on v_mfma_f32_32x32x16_bf16
// (x = x0 + x1 + x2, w = w0 + w1 + w2 exactly; x*w ~ x0w0 + x0w1 + x1w0 + x0w2 + x1w1 + x2w0: 6 MFMAs at 16x the f32
// MFMA rate), weights (A operand, pre-split, 96 KB) in LDS, activations (B operand) in registers, R row tiles of 32
// samples per wave sharing every A operand read.  What does the loop sustain, and at which (waves, R)?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/bf16x3_loop.hip -o /tmp/bf16x3_loop && /tmp/bf16x3_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

template <int WAVES, int R>
__global__ void __launch_bounds__(WAVES * 64) k(const u32x4* __restrict__ w, const u32x4* __restrict__ x, float* __restrict__ out, int iters) {
  __shared__ u32x4 lw[4 * 8 * 3 * 64];          // [t2][ks][split][lane] 16 B: 96 KB
  for (int i = threadIdx.x; i < 4 * 8 * 3 * 64; i += blockDim.x) lw[i] = w[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  u32x4 b[R][8][3];
  for (int r = 0; r < R; ++r)
    for (int ks = 0; ks < 8; ++ks)
      for (int s = 0; s < 3; ++s) b[r][ks][s] = x[((blockIdx.x * WAVES * 64 + threadIdx.x) * R + r) * 24 + ks * 3 + s];
  float tot = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll 1
    for (int t2 = 0; t2 < 4; ++t2) {
      f32x16 acc[R];
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const u32x4 a0 = lw[((t2 * 8 + ks) * 3 + 0) * 64 + lane];
        const u32x4 a1 = lw[((t2 * 8 + ks) * 3 + 1) * 64 + lane];
        const u32x4 a2 = lw[((t2 * 8 + ks) * 3 + 2) * 64 + lane];
#pragma unroll
        for (int r = 0; r < R; ++r) {
#define MF(A, B) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc[r], 0, 0, 0)
          MF(a0, b[r][ks][0]); MF(a0, b[r][ks][1]); MF(a1, b[r][ks][0]);
          MF(a0, b[r][ks][2]); MF(a1, b[r][ks][1]); MF(a2, b[r][ks][0]);
#undef MF
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += acc[r][i];
    }
  }
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = tot;
}

template <int WAVES, int R>
void run(const u32x4* w, const u32x4* x, float* o) {
  const int blocks = 256, iters = 200;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int warm = 0; warm < 2; ++warm) k<WAVES, R><<<blocks, WAVES * 64>>>(w, x, o, iters);
  hipEventRecord(a);
  for (int n = 0; n < 5; ++n) k<WAVES, R><<<blocks, WAVES * 64>>>(w, x, o, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  const double tiles = (double)blocks * WAVES * R * iters;            // 32-row tiles through the 128x128 layer
  const double flop = tiles * 32 * 128 * 128 * 2;
  printf("waves %2d R %d: %.3f ms  %.1f ns per row tile per CU-wave  f32-equivalent %.1f TFLOP/s;  65536 tiles -> %.3f ms\n", WAVES, R, ms,
         ms * 1e6 / (iters * R), flop / ms / 1e9, ms / tiles * 65536);
}

int main() {
  const size_t nw = 4 * 8 * 3 * 64, nx = (size_t)256 * 12 * 64 * 3 * 24;
  std::vector<unsigned int> hw(nw * 4), hx(nx * 4);
  auto rnd_bf16_pair = [] {
    auto one = [] { float f = (float)rand() / RAND_MAX - 0.5f; unsigned int u; memcpy(&u, &f, 4); return u >> 16; };
    return one() | (one() << 16);
  };
  for (auto& v : hw) v = rnd_bf16_pair();
  for (auto& v : hx) v = rnd_bf16_pair();
  u32x4 *w, *x; float* o;
  hipMalloc(&w, hw.size() * 4); hipMalloc(&x, hx.size() * 4); hipMalloc(&o, 256 * 12 * 64 * 4);
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  run<4, 1>(w, x, o); run<4, 2>(w, x, o); run<4, 3>(w, x, o);
  run<8, 1>(w, x, o); run<8, 2>(w, x, o);
  run<12, 1>(w, x, o);
  return 0;
}
