// The exact 3-way bf16 split of fp32 operands and the six-product MFMA group built on it (see shade_x3.hip for the
// arithmetic); shared by shade_x3.hip and the split-operand weight-gradient kernel of shade.hip.
#pragma once
#include "common.h"

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

__device__ __forceinline__ unsigned int x3_pack(float a, float b) {          // two fp32 -> two bf16 (RNE), a in the low half
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 v; v[0] = a; v[1] = b;
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}

// 8 fp32 -> the three bf16 fragments (element j of the fragment = v[j])
__device__ __forceinline__ void x3_split8(const float (&v)[8], u32x4& s0, u32x4& s1, u32x4& s2) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = v[2 * p], b = v[2 * p + 1];
    const unsigned int h = x3_pack(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    const unsigned int m = x3_pack(ra, rb);
    const float qa = ra - __uint_as_float(m << 16), qb = rb - __uint_as_float(m & 0xffff0000u);
    s0[p] = h; s1[p] = m; s2[p] = x3_pack(qa, qb);
  }
}

// The same split with the subtractions hidden from the SLP vectoriser: packed (v_pk_add_f32) they want each pair of values
// in an aligned register pair, and values that arrive as the components of DIFFERENT wide loads then get copied into place
// -- behind a wait for those loads (shade_wgrad_c*).  Same instruction count, no constraint on where the values live.
__device__ __forceinline__ float x3_sub(float a, float b) {
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void x3_split8_free(const float (&v)[8], u32x4& s0, u32x4& s1, u32x4& s2) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = v[2 * p], b = v[2 * p + 1];
    const unsigned int h = x3_pack(a, b);
    const float ra = x3_sub(a, __uint_as_float(h << 16)), rb = x3_sub(b, __uint_as_float(h & 0xffff0000u));
    const unsigned int m = x3_pack(ra, rb);
    const float qa = x3_sub(ra, __uint_as_float(m << 16)), qb = x3_sub(rb, __uint_as_float(m & 0xffff0000u));
    s0[p] = h; s1[p] = m; s2[p] = x3_pack(qa, qb);
  }
}

// acc += A * B with the six partial products (A, B given as their three fragments)
__device__ __forceinline__ void x3_mfma6(f32x16& acc, const u32x4 a0, const u32x4 a1, const u32x4 a2, const u32x4 b0,
                                         const u32x4 b1, const u32x4 b2) {
#define X3_MF(A, B) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc, 0, 0, 0)
  X3_MF(a2, b0); X3_MF(a1, b1); X3_MF(a0, b2);       // smallest terms first
  X3_MF(a1, b0); X3_MF(a0, b1);
  X3_MF(a0, b0);
#undef X3_MF
}

