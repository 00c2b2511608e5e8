// Micro-test: is it safe to overwrite the A operand register of a v_mfma_f32_32x32x16_bf16 with an LDS load issued right
// behind it?  hipcc (ROCm 7.2) does so (it treats MFMA A/B operands as read at issue); with a chain of DEPENDENT MFMAs
// queued in front, the last one may start -- and read its operands -- long after it was issued.
//   acc = 0;  Q x { acc = A_old . B + acc }  issued back to back;  ds_read A <- A_new  immediately behind the last one
//   expected: acc == Q * (A_old . B) exactly (all products equal, sums of equal terms of a power-of-two count are exact
//   for the data used here)
// One or two waves per SIMD (the second wave competes for the matrix pipe and delays the queue).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_war_lds.hip -o tools/micro/mfma_war_lds && tools/micro/mfma_war_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

template <int Q>
__global__ void __launch_bounds__(512) k(const u32x4* __restrict__ a_old, const u32x4* __restrict__ a_new, const u32x4* __restrict__ b,
                                          int* __restrict__ bad, int iters) {
  __shared__ u32x4 lds[2][64];
  __shared__ char pad[90 * 1024];                     // one workgroup per CU
  if (iters < 0) pad[threadIdx.x] = 0;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64) { lds[0][lane] = a_old[lane]; lds[1][lane] = a_new[lane]; }
  __syncthreads();
  const u32x4 B = b[lane];
  // reference: one product, waited for
  f32x16 one;
#pragma unroll
  for (int r = 0; r < 16; ++r) one[r] = 0.f;
  one = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, lds[0][lane]), __builtin_bit_cast(bf16x8, B), one, 0, 0, 0);
  const unsigned la = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)&lds[0][lane];
  int wrong = 0;
  for (int it = 0; it < iters; ++it) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    u32x4 A;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(A) : "v"(la));
    if (Q == 1)
      asm volatile("s_nop 4\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "ds_read_b128 %1, %3 offset:1024\n\t"
                   "s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15"
                   : "+v"(acc), "+v"(A) : "v"(B), "v"(la));
    else if (Q == 2)
      asm volatile("s_nop 4\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "ds_read_b128 %1, %3 offset:1024\n\t"
                   "s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15"
                   : "+v"(acc), "+v"(A) : "v"(B), "v"(la));
    else if (Q == 4)
      asm volatile("s_nop 4\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "ds_read_b128 %1, %3 offset:1024\n\t"
                   "s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15"
                   : "+v"(acc), "+v"(A) : "v"(B), "v"(la));
    else
      asm volatile("s_nop 4\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "ds_read_b128 %1, %3 offset:1024\n\t"
                   "s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15"
                   : "+v"(acc), "+v"(A) : "v"(B), "v"(la));
#pragma unroll
    for (int r = 0; r < 16; ++r) wrong += (acc[r] != (float)Q * one[r]);
    if (A[0] == 0x12345u) wrong += 1000;               // keep A alive
  }
  if (wrong) atomicAdd(bad, wrong);
}

template <int Q>
void run(const u32x4* ao, const u32x4* an, const u32x4* b, int* bad, int threads) {
  hipMemset(bad, 0, 4);
  k<Q><<<256, threads>>>(ao, an, b, bad, 20000);
  hipDeviceSynchronize();
  int h = 0;
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("%d waves/SIMD, %d dependent MFMAs queued, A overwritten by ds_read behind the last: %9d wrong elements of %lld\n", threads / 256, Q,
         h, 256LL * threads * 20000 * 16);
}

int main() {
  std::vector<unsigned> ao(64 * 4), an(64 * 4), hb(64 * 4);
  srand(3);
  // bf16 values 1.0 / 2.0 / 0.5 ... (exact products and sums): old A = 1.0, new A = 3.0, B = powers of two
  for (auto& v : ao) v = 0x3f803f80u;                 // {1.0, 1.0}
  for (auto& v : an) v = 0x40404040u;                 // {3.0, 3.0}
  for (auto& v : hb) { const unsigned short x = (rand() & 1) ? 0x3f80 : 0x4000, y = (rand() & 1) ? 0x3f00 : 0x3f80; v = x | ((unsigned)y << 16); }
  u32x4 *dao, *dan, *db; int* bad;
  hipMalloc(&dao, 1024); hipMalloc(&dan, 1024); hipMalloc(&db, 1024); hipMalloc(&bad, 4);
  hipMemcpy(dao, ao.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(dan, an.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
  for (int threads : {256, 512}) {
    run<1>(dao, dan, db, bad, threads); run<2>(dao, dan, db, bad, threads); run<4>(dao, dan, db, bad, threads); run<8>(dao, dan, db, bad, threads);
  }
  return 0;
}
