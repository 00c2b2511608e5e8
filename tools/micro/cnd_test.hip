#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(X) X X X X X X X X
#define R64(X) R8(R8(X))
template <int WHICH>
__global__ void k(float* out, int passes) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float x = 1.0f + threadIdx.x * 1e-3f, y = 0.5f;
  asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cmp_gt_f32_e64 s[20:21], %0, %1" :: "v"(x), "v"(y) : "vcc", "s20", "s21");
  for (int p = 0; p < passes; ++p) {
    // 0: cmp + 7 cndmask(vcc)   1: 8 cndmask (vcc written once before the loop)   2: 8 cndmask e64 vcc   3: cmp(e64 s) + 7 cndmask s
    // 4: 8 cndmask vcc, each followed by v_add   5: s_mov vcc + 8 cndmask   6: s_nop 0 between cndmasks
    if (WHICH == 0) asm volatile(R64("v_cmp_gt_f32 vcc, %8, %9\n\tv_cndmask_b32 %0, %8, %9, vcc\n\tv_cndmask_b32 %1, %8, %9, vcc\n\tv_cndmask_b32 %2, %8, %9, vcc\n\tv_cndmask_b32 %3, %8, %9, vcc\n\tv_cndmask_b32 %4, %8, %9, vcc\n\tv_cndmask_b32 %5, %8, %9, vcc\n\tv_cndmask_b32 %6, %8, %9, vcc\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc");
    if (WHICH == 1) asm volatile(R64("v_cndmask_b32 %0, %8, %9, vcc\n\tv_cndmask_b32 %1, %8, %9, vcc\n\tv_cndmask_b32 %2, %8, %9, vcc\n\tv_cndmask_b32 %3, %8, %9, vcc\n\tv_cndmask_b32 %4, %8, %9, vcc\n\tv_cndmask_b32 %5, %8, %9, vcc\n\tv_cndmask_b32 %6, %8, %9, vcc\n\tv_cndmask_b32 %7, %8, %9, vcc\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    if (WHICH == 2) asm volatile(R64("v_cndmask_b32_e64 %0, %8, %9, vcc\n\tv_cndmask_b32_e64 %1, %8, %9, vcc\n\tv_cndmask_b32_e64 %2, %8, %9, vcc\n\tv_cndmask_b32_e64 %3, %8, %9, vcc\n\tv_cndmask_b32_e64 %4, %8, %9, vcc\n\tv_cndmask_b32_e64 %5, %8, %9, vcc\n\tv_cndmask_b32_e64 %6, %8, %9, vcc\n\tv_cndmask_b32_e64 %7, %8, %9, vcc\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    if (WHICH == 3) asm volatile(R64("v_cmp_gt_f32_e64 s[20:21], %8, %9\n\tv_cndmask_b32_e64 %0, %8, %9, s[20:21]\n\tv_cndmask_b32_e64 %1, %8, %9, s[20:21]\n\tv_cndmask_b32_e64 %2, %8, %9, s[20:21]\n\tv_cndmask_b32_e64 %3, %8, %9, s[20:21]\n\tv_cndmask_b32_e64 %4, %8, %9, s[20:21]\n\tv_cndmask_b32_e64 %5, %8, %9, s[20:21]\n\tv_cndmask_b32_e64 %6, %8, %9, s[20:21]\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "s20", "s21");
    if (WHICH == 4) asm volatile(R64("v_cndmask_b32 %0, %8, %9, vcc\n\tv_add_f32 %1, %8, %9\n\tv_cndmask_b32 %2, %8, %9, vcc\n\tv_add_f32 %3, %8, %9\n\tv_cndmask_b32 %4, %8, %9, vcc\n\tv_add_f32 %5, %8, %9\n\tv_cndmask_b32 %6, %8, %9, vcc\n\tv_add_f32 %7, %8, %9\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    if (WHICH == 5) asm volatile(R64("v_cndmask_b32 %0, %8, %9, vcc\n\tv_cndmask_b32 %0, %8, %9, vcc\n\tv_cndmask_b32 %0, %8, %9, vcc\n\tv_cndmask_b32 %0, %8, %9, vcc\n\tv_cndmask_b32 %0, %8, %9, vcc\n\tv_cndmask_b32 %0, %8, %9, vcc\n\tv_cndmask_b32 %0, %8, %9, vcc\n\tv_cndmask_b32 %0, %8, %9, vcc\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    if (WHICH == 6) asm volatile(R64("v_cndmask_b32 %0, %8, %9, vcc\n\tv_cndmask_b32 %1, %9, %8, vcc\n\tv_cndmask_b32 %2, %8, %9, vcc\n\tv_cndmask_b32 %3, %9, %8, vcc\n\tv_cndmask_b32 %4, %8, %9, vcc\n\tv_cndmask_b32 %5, %9, %8, vcc\n\tv_cndmask_b32 %6, %8, %9, vcc\n\tv_cndmask_b32 %7, %9, %8, vcc\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
  float* out; if (hipMalloc(&out, 256 * 1024 * 4) != hipSuccess) return 1;
  const char* names[] = {"cmp + 7 cndmask vcc", "8 cndmask vcc (vcc old)", "8 cndmask_e64 vcc", "cmp_e64 + 7 cndmask s[20:21]", "cndmask vcc / add alternating", "8 cndmask same dst", "8 cndmask alternating srcs"};
  for (int w = 0; w < 7; ++w) {
    printf("%-32s", names[w]);
    for (int wps : {1, 2, 4}) {
      const int threads = 256 * wps;
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      auto launch = [&]() { switch (w) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, 100); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, 100); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, 100); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, 100); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 0, 0, out, 100); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(threads), 0, 0, out, 100); break;
        case 6: hipLaunchKernelGGL(k<6>, dim3(256), dim3(threads), 0, 0, out, 100); break; } };
      launch(); (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf(" %10.2f", ms * 1e6 / (512.0 * 100 * wps) * 2.4);
    }
    printf("\n");
  }
}
