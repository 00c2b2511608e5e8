// Micro-test: may a VALU instruction overwrite an A-operand register of a v_mfma_f32_32x32x16_bf16 in the issue slot(s)
// right behind it?  hipcc (ROCm 7.2) schedules exactly that (the operand split of the next fragment reuses the register):
//     v_mfma_f32_32x32x16_bf16 v[0:15], v[134:137], v[24:27], v[0:15]
//     v_lshlrev_b32 v134, 16, v27
// Here: Q dependent MFMAs back to back (so that the last one may sit in the queue), K wait states, then v_mov_b32 into
// each of the four A registers; the accumulator must equal Q x (A . B).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_war_valu.hip -o tools/micro/mfma_war_valu && tools/micro/mfma_war_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

#define LOADA "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\ts_nop 7\n\t"
#define MF "v_mfma_f32_32x32x16_bf16 %0, v[100:103], %1, %0\n\t"
#define CLOB "v_mov_b32 v100, %6\n\tv_mov_b32 v101, %6\n\tv_mov_b32 v102, %6\n\tv_mov_b32 v103, %6\n\t"
#define TAIL "s_nop 15\n\ts_nop 15\n\ts_nop 15"
#define RUNQ(MFS, GAP) asm volatile(LOADA MFS GAP CLOB TAIL : "+v"(acc) : "v"(B), "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(junk) \
                                    : "v100", "v101", "v102", "v103")

template <int Q, int K>
__global__ void __launch_bounds__(512) k(const u32x4* __restrict__ a, const u32x4* __restrict__ b, int* __restrict__ bad, int iters) {
  __shared__ char pad[100 * 1024];                     // one workgroup per CU
  if (iters < 0) pad[threadIdx.x] = 0;
  const int lane = threadIdx.x & 63;
  const u32x4 A = a[lane], B = b[lane];
  f32x16 one;
#pragma unroll
  for (int r = 0; r < 16; ++r) one[r] = 0.f;
  asm volatile(LOADA MF TAIL : "+v"(one) : "v"(B), "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]) : "v100", "v101", "v102", "v103");
  int wrong = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned junk = 0x7f007f00u ^ (unsigned)it;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (Q == 1) { if (K == 0) RUNQ(MF, ""); else if (K == 1) RUNQ(MF, "s_nop 0\n\t"); else if (K == 2) RUNQ(MF, "s_nop 1\n\t"); else RUNQ(MF, "s_nop 3\n\t"); }
    else if (Q == 3) { if (K == 0) RUNQ(MF MF MF, ""); else if (K == 1) RUNQ(MF MF MF, "s_nop 0\n\t"); else if (K == 2) RUNQ(MF MF MF, "s_nop 1\n\t"); else RUNQ(MF MF MF, "s_nop 3\n\t"); }
    else { if (K == 0) RUNQ(MF MF MF MF MF MF, ""); else if (K == 1) RUNQ(MF MF MF MF MF MF, "s_nop 0\n\t"); else if (K == 2) RUNQ(MF MF MF MF MF MF, "s_nop 1\n\t"); else RUNQ(MF MF MF MF MF MF, "s_nop 3\n\t"); }
#pragma unroll
    for (int r = 0; r < 16; ++r) wrong += (acc[r] != (float)Q * one[r]);
  }
  if (wrong) atomicAdd(bad, wrong);
}

template <int Q, int K>
void run(const u32x4* a, const u32x4* b, int* bad, int threads) {
  hipMemset(bad, 0, 4);
  k<Q, K><<<256, threads>>>(a, b, bad, 5000);
  hipDeviceSynchronize();
  int h = 0;
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("%d waves/SIMD, %d dependent MFMAs, %d wait states, then VALU writes to the A operand registers: %9d wrong elements of %lld\n",
         threads / 256, Q, K, h, 256LL * threads * 5000 * 16);
}

int main() {
  std::vector<unsigned> ha(64 * 4), hb(64 * 4);
  for (auto& v : ha) v = 0x3f803f80u;                 // 1.0: Q x (A . B) is exact
  srand(3);
  for (auto& v : hb) { const unsigned short x = (rand() & 1) ? 0x3f80 : 0x4000, y = (rand() & 1) ? 0x3f00 : 0x3f80; v = x | ((unsigned)y << 16); }
  u32x4 *a, *b; int* bad;
  hipMalloc(&a, 1024); hipMalloc(&b, 1024); hipMalloc(&bad, 4);
  hipMemcpy(a, ha.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), 1024, hipMemcpyHostToDevice);
  for (int threads : {256, 512}) {
    run<1, 0>(a, b, bad, threads); run<1, 1>(a, b, bad, threads); run<1, 2>(a, b, bad, threads); run<1, 4>(a, b, bad, threads);
    run<3, 0>(a, b, bad, threads); run<3, 1>(a, b, bad, threads); run<3, 2>(a, b, bad, threads); run<3, 4>(a, b, bad, threads);
    run<6, 0>(a, b, bad, threads); run<6, 1>(a, b, bad, threads); run<6, 2>(a, b, bad, threads); run<6, 4>(a, b, bad, threads);
  }
  return 0;
}
