// Micro-test: how many VALU instructions of the operand split does a wave issue under its own bf16 MFMAs for free?
// Loop body: 12 v_mfma_f32_32x32x16_bf16 on two alternating accumulators, K split-type VALU instructions (cvt_pk / lshl / and /
// sub on other registers) behind each MFMA; ns per MFMA for K = 0..12 at 1 and 2 waves per SIMD, every CU busy.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o tools/micro/mfma_valu_overlap && tools/micro/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
#define VALU4 "v_cvt_pk_bf16_f32 %4, %6, %7\n\tv_lshlrev_b32 %5, 16, %4\n\tv_and_b32 %4, 0xffff0000, %4\n\tv_sub_f32 %6, %6, %5\n\t"
#define VALU2B "v_sub_f32 %3, %3, %0\n\tv_cvt_pk_bf16_f32 %1, %2, %3\n\t"
template <int K>
__global__ void k(float* out, int iters) {
  f32x16 a, b;
  for (int r = 0; r < 16; ++r) { a[r] = r; b[r] = -r; }
  u32x4 fa = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, fb = fa;
  unsigned t0 = threadIdx.x, t1 = 1;
  float x = 1.5f + threadIdx.x, y = 0.25f;
  for (int it = 0; it < iters; ++it) {
#define STEP(ACC)                                                                                                         \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t" : "+v"(ACC) : "v"(fa), "v"(fb));                               \
    if (K >= 2) asm volatile(VALU2B : "+v"(t0), "+v"(t1), "+v"(x), "+v"(y));            \
    if (K >= 4) asm volatile(VALU2B : "+v"(t0), "+v"(t1), "+v"(x), "+v"(y));            \
    if (K >= 6) asm volatile(VALU2B : "+v"(t0), "+v"(t1), "+v"(x), "+v"(y));            \
    if (K >= 8) asm volatile(VALU2B : "+v"(t0), "+v"(t1), "+v"(x), "+v"(y));            \
    if (K >= 10) asm volatile(VALU2B : "+v"(t0), "+v"(t1), "+v"(x), "+v"(y));           \
    if (K >= 12) asm volatile(VALU2B : "+v"(t0), "+v"(t1), "+v"(x), "+v"(y));
    STEP(a) STEP(b) STEP(a) STEP(b) STEP(a) STEP(b) STEP(a) STEP(b) STEP(a) STEP(b) STEP(a) STEP(b)
  }
  float s = x + y + __uint_as_float(t0) + __uint_as_float(t1);
  for (int r = 0; r < 16; ++r) s += a[r] + b[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; if (hipMalloc(&out, 256 * 1024 * 4) != hipSuccess) return 1;
  const int iters = 2000;
  printf("ns per MFMA per SIMD (v_mfma_f32_32x32x16_bf16 = 32 cycles: 13.3 ns at 2.4 GHz), K VALU instructions behind each\n");
  printf("%4s %12s %12s\n", "K", "1 wave/SIMD", "2 waves/SIMD");
  for (int K = 0; K <= 12; K += 2) {
    printf("%4d", K);
    for (int wps : {1, 2}) {
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      auto launch = [&]() {
        switch (K) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * wps), 0, 0, out, iters); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(256 * wps), 0, 0, out, iters); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(256 * wps), 0, 0, out, iters); break;
          case 6: hipLaunchKernelGGL(k<6>, dim3(256), dim3(256 * wps), 0, 0, out, iters); break;
          case 8: hipLaunchKernelGGL(k<8>, dim3(256), dim3(256 * wps), 0, 0, out, iters); break;
          case 10: hipLaunchKernelGGL(k<10>, dim3(256), dim3(256 * wps), 0, 0, out, iters); break;
          case 12: hipLaunchKernelGGL(k<12>, dim3(256), dim3(256 * wps), 0, 0, out, iters); break;
        }
      };
      launch(); (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf(" %12.2f", ms * 1e6 / (12.0 * iters * wps));
    }
    printf("\n");
  }
  return 0;
}
