// Micro-test: does a v_mfma_f32_32x32x16_bf16 that accumulates into the result of the previous one read the right
// accumulator when K independent VALU instructions sit between the two?  (hipcc / ROCm 7.2 puts no wait states there:
// it relies on the hardware's dependency handling for same-type back-to-back accumulation.)
//   acc = A.B ; K x v_mov (independent) ; acc = A.B + acc   -> every element must equal 2 A.B exactly
// One wave per SIMD (256 threads, one workgroup per CU) so that nothing else fills the gap.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_dep_gap.hip -o tools/micro/mfma_dep_gap && tools/micro/mfma_dep_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

template <int K, bool F32>
__global__ void __launch_bounds__(512) k(const u32x4* __restrict__ a, const u32x4* __restrict__ b, int* __restrict__ bad, int iters) {
  __shared__ char pad[100 * 1024];                     // one workgroup per CU
  if (iters < 0) pad[threadIdx.x] = 0;
  const u32x4 A = a[threadIdx.x & 63], B = b[threadIdx.x & 63];   // (the f32 variant's check is not exact: ignore F32 = true)
  int wrong = 0;
  float d0 = 1.0f, d1 = 2.0f, d2 = 3.0f, d3 = 4.0f;
  for (int it = 0; it < iters; ++it) {
    f32x16 one, two;
#pragma unroll
    for (int r = 0; r < 16; ++r) one[r] = 0.f;
    if (F32) one = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(A[0]), __uint_as_float(B[0]), one, 0, 0, 0);
    else one = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), one, 0, 0, 0);
    two = one;                                         // reference: wait for it (the compiler pads this read)
    f32x16 acc;
    float z = 0.f;
    asm volatile("" : "+v"(z));                        // an opaque zero: keeps this chain apart from the reference
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = z;
    __builtin_amdgcn_sched_barrier(0);
    if (F32) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(A[0]), __uint_as_float(B[0]), acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < K; ++g) {                      // K independent VALU instructions
      if ((g & 3) == 0) asm volatile("v_add_f32 %0, %0, %0" : "+v"(d0));
      else if ((g & 3) == 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(d1));
      else if ((g & 3) == 2) asm volatile("v_add_f32 %0, %0, %0" : "+v"(d2));
      else asm volatile("v_add_f32 %0, %0, %0" : "+v"(d3));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (F32) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(A[0]), __uint_as_float(B[0]), acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) wrong += (acc[r] != 2.0f * two[r]);
  }
  if (wrong || d0 + d1 + d2 + d3 == 12345.f) atomicAdd(bad, wrong);
}

static int THREADS = 256;          // 256: one wave per SIMD; 512: two (the second competes for the matrix pipe)
template <int K, bool F32>
void run(const u32x4* a, const u32x4* b, int* bad) {
  hipMemset(bad, 0, 4);
  k<K, F32><<<256, THREADS>>>(a, b, bad, 2000);
  hipDeviceSynchronize();
  int h = 0;
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("%d waves/SIMD %s  K = %2d VALU between dependent MFMAs: %9d wrong elements of %lld\n", THREADS / 256, F32 ? "f32 32x32x2  " : "bf16 32x32x16", K, h,
         256LL * THREADS * 2000 * 16);
}

int main() {
  std::vector<unsigned> ha(64 * 4), hb(64 * 4);
  srand(3);
  for (auto& v : ha) { const unsigned short x = 0x3f00 + (rand() & 0xff), y = 0x3f00 + (rand() & 0xff); v = x | ((unsigned)y << 16); }
  for (auto& v : hb) { const unsigned short x = 0x3f00 + (rand() & 0xff), y = 0xbf00 + (rand() & 0xff); v = x | ((unsigned)y << 16); }
  u32x4 *a, *b; int* bad;
  hipMalloc(&a, 64 * 16); hipMalloc(&b, 64 * 16); hipMalloc(&bad, 4);
  hipMemcpy(a, ha.data(), 64 * 16, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), 64 * 16, hipMemcpyHostToDevice);
  for (int th : {256, 512}) {
    THREADS = th;
    run<0, false>(a, b, bad); run<1, false>(a, b, bad); run<2, false>(a, b, bad); run<3, false>(a, b, bad); run<4, false>(a, b, bad);
    run<6, false>(a, b, bad); run<8, false>(a, b, bad); run<10, false>(a, b, bad); run<12, false>(a, b, bad); run<16, false>(a, b, bad);
  }
  return 0;
}
