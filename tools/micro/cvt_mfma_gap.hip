// Micro-test: how many wait states does v_mfma_f32_32x32x16_bf16 need behind a VALU instruction that writes one of its
// B-operand registers?  hipcc (ROCm 7.2) leaves 2 (it pads with s_nop 0 when the producer is v_cvt_pk_bf16_f32 two
// instructions earlier).  Here: the last register of the B fragment is produced by v_cvt_pk_bf16_f32 (or v_mov_b32),
// K wait states later the MFMA reads it; the result is compared with the same MFMA issued after a long wait.
// The stale value the register held before is different, so a too-early read shows.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/cvt_mfma_gap.hip -o tools/micro/cvt_mfma_gap && tools/micro/cvt_mfma_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

#define GAP_0 ""
#define GAP_1 "s_nop 0\n\t"
#define GAP_2 "s_nop 1\n\t"
#define GAP_3 "s_nop 2\n\t"
#define GAP_4 "s_nop 3\n\t"
#define GAP_6 "s_nop 5\n\t"
#define GAP_8 "s_nop 7\n\t"
// the same two (three) wait states made of scalar ALU instructions, as hipcc counts them (K = 12: s_and + s_nop 0; 13: two SALU; 14: three)
#define GAP_12 "s_and_b64 s[40:41], exec, s[42:43]\n\ts_nop 0\n\t"
#define GAP_13 "s_and_b64 s[40:41], exec, s[42:43]\n\ts_mov_b32 s44, 1\n\t"
#define GAP_14 "s_and_b64 s[40:41], exec, s[42:43]\n\ts_mov_b32 s44, 1\n\ts_mov_b32 s45, 2\n\t"
#define BODY(PROD, GAP)                                                                                   \
  asm volatile("v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\t"   \
               "s_nop 7\n\t" PROD GAP "v_mfma_f32_32x32x16_bf16 %0, %1, v[100:103], %0\n\t"              \
               "s_nop 15\n\ts_nop 15"                                                                     \
               : "+v"(acc) : "v"(A), "v"(stale), "v"(B[1]), "v"(B[2]), "v"(B[3]), "v"(x0), "v"(x1)          \
               : "v100", "v101", "v102", "v103", "s40", "s41", "s42", "s43", "s44", "s45")
#define CVT "v_cvt_pk_bf16_f32 v100, %6, %7\n\t"       /* (re)writes the FIRST register of the B fragment */
#define MOV "v_mov_b32 v100, %6\n\t"

template <int K, bool USE_CVT>
__global__ void __launch_bounds__(512) k(const u32x4* __restrict__ a, const u32x4* __restrict__ b, int* __restrict__ bad, int iters) {
  __shared__ char pad[100 * 1024];                     // one workgroup per CU
  if (iters < 0) pad[threadIdx.x] = 0;
  const int lane = threadIdx.x & 63;
  const u32x4 A = a[lane];
  int wrong = 0;
  for (int it = 0; it < iters; ++it) {
    const float x0 = 1.0f + (float)((it + lane) & 7), x1 = 2.0f + (float)((it * 3 + lane) & 3);     // exact in bf16
    // reference: fragment completed long before the MFMA
    u32x4 Br = b[lane];
    if (USE_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(Br[0]) : "v"(x0), "v"(x1));
    else Br[0] = __float_as_uint(x0);
    f32x16 ref;
#pragma unroll
    for (int r = 0; r < 16; ++r) ref[r] = 0.f;
    asm volatile("s_nop 15\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %1, %0\n\ts_nop 15\n\ts_nop 15" : "+v"(ref) : "v"(Br), "v"(A));
    // test: the fragment's first register holds something else until K wait states before the MFMA
    const u32x4 B = b[lane];
    const unsigned stale = 0x7f007f00u ^ (unsigned)it; // what the register holds before (large: a stale read is unmissable)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (USE_CVT) {
      if (K == 0) BODY(CVT, GAP_0); else if (K == 1) BODY(CVT, GAP_1); else if (K == 2) BODY(CVT, GAP_2); else if (K == 3) BODY(CVT, GAP_3);
      else if (K == 4) BODY(CVT, GAP_4); else if (K == 6) BODY(CVT, GAP_6); else if (K == 12) BODY(CVT, GAP_12);
      else if (K == 13) BODY(CVT, GAP_13); else if (K == 14) BODY(CVT, GAP_14); else BODY(CVT, GAP_8);
    } else {
      if (K == 0) BODY(MOV, GAP_0); else if (K == 1) BODY(MOV, GAP_1); else if (K == 2) BODY(MOV, GAP_2); else if (K == 3) BODY(MOV, GAP_3);
      else if (K == 4) BODY(MOV, GAP_4); else if (K == 6) BODY(MOV, GAP_6); else BODY(MOV, GAP_8);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) wrong += (acc[r] != ref[r]);
  }
  if (wrong) atomicAdd(bad, wrong);
}

template <int K, bool USE_CVT>
void run(const u32x4* a, const u32x4* b, int* bad, int threads) {
  hipMemset(bad, 0, 4);
  k<K, USE_CVT><<<256, threads>>>(a, b, bad, 5000);
  hipDeviceSynchronize();
  int h = 0;
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("%d waves/SIMD  %s -> MFMA B operand, %d wait states between: %9d wrong elements of %lld\n", threads / 256,
         USE_CVT ? "v_cvt_pk_bf16_f32" : "v_mov_b32        ", K, h, 256LL * threads * 5000 * 16);
}

int main() {
  std::vector<unsigned> ha(64 * 4), hb(64 * 4);
  srand(3);
  for (auto& v : ha) { const unsigned short x = 0x3f80 + ((rand() & 3) << 7), y = 0x3f80 + ((rand() & 3) << 7); v = x | ((unsigned)y << 16); }
  for (auto& v : hb) { const unsigned short x = 0x3f80 + ((rand() & 3) << 7), y = 0x3f80; v = x | ((unsigned)y << 16); }
  u32x4 *a, *b; int* bad;
  hipMalloc(&a, 1024); hipMalloc(&b, 1024); hipMalloc(&bad, 4);
  hipMemcpy(a, ha.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), 1024, hipMemcpyHostToDevice);
  for (int threads : {256, 512}) {
    run<0, true>(a, b, bad, threads); run<1, true>(a, b, bad, threads); run<2, true>(a, b, bad, threads); run<3, true>(a, b, bad, threads);
    run<4, true>(a, b, bad, threads); run<6, true>(a, b, bad, threads); run<8, true>(a, b, bad, threads);
    run<12, true>(a, b, bad, threads); run<13, true>(a, b, bad, threads); run<14, true>(a, b, bad, threads);
    run<0, false>(a, b, bad, threads); run<1, false>(a, b, bad, threads); run<2, false>(a, b, bad, threads); run<4, false>(a, b, bad, threads);
  }
  return 0;
}
