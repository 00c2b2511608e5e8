// "next" rows N1/N2 of SURVEY.md section 8f: MaskedAdam updates (K15-K17) and the
// total-variation gradient (K14).  Pure streaming kernels, HBM bound.
// Reference semantics: /root/reference/lib/cuda/adam_upd_kernel.cu:8-58,
//                      /root/reference/lib/cuda/total_variation_kernel.cu:13-35.
#include "common.h"

template <int MODE>
__global__ void __launch_bounds__(DVGO_BLOCK)
adam_kernel(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ exp_avg,
            float* __restrict__ exp_avg_sq, const float* __restrict__ perlr, int64_t n,
            float step_size, float beta1, float beta2, float eps, bool vec) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    const int64_t i = t * 4;
    if (i + 3 < n) {
      float4 p = *reinterpret_cast<float4*>(param + i);
      const float4 g = *reinterpret_cast<const float4*>(grad + i);
      if (MODE == 1 && g.x == 0.f && g.y == 0.f && g.z == 0.f && g.w == 0.f) return;   // nothing to write
      float4 m = *reinterpret_cast<float4*>(exp_avg + i);
      float4 v = *reinterpret_cast<float4*>(exp_avg_sq + i);
      float4 l = make_float4(0.f, 0.f, 0.f, 0.f);
      if (MODE == 2) l = *reinterpret_cast<const float4*>(perlr + i);
      adam_one<MODE>(p.x, g.x, m.x, v.x, l.x, step_size, beta1, beta2, eps);
      adam_one<MODE>(p.y, g.y, m.y, v.y, l.y, step_size, beta1, beta2, eps);
      adam_one<MODE>(p.z, g.z, m.z, v.z, l.z, step_size, beta1, beta2, eps);
      adam_one<MODE>(p.w, g.w, m.w, v.w, l.w, step_size, beta1, beta2, eps);
      *reinterpret_cast<float4*>(param + i) = p;
      *reinterpret_cast<float4*>(exp_avg + i) = m;
      *reinterpret_cast<float4*>(exp_avg_sq + i) = v;
      return;
    }
    for (int64_t k = i; k < n; ++k)
      adam_one<MODE>(param[k], grad[k], exp_avg[k], exp_avg_sq[k], MODE == 2 ? perlr[k] : 0.f, step_size,
                     beta1, beta2, eps);
  } else if (t < n) {
    adam_one<MODE>(param[t], grad[t], exp_avg[t], exp_avg_sq[t], MODE == 2 ? perlr[t] : 0.f, step_size,
                   beta1, beta2, eps);
  }
}

// Adam straight from the combined gradient rows of the fused march backward (march.hip: [n_vox][16] floats = 12 feature
// channels, the density gradient, 3 pad): updates the channels-last feature grid and the density grid in one pass and
// saves the split into two dense gradients plus their re-read (0.07 ms per step at 160^3).  Element-wise maths and the
// masked rule are those of adam_kernel<0/1>; 4 threads per row, one float4 each.
template <int MODE_K, int MODE_D>
__global__ void __launch_bounds__(DVGO_BLOCK)
adam_rows_kernel(const float* __restrict__ G, int64_t n_vox, float* __restrict__ pk, float* __restrict__ mk,
                 float* __restrict__ vk, float ss_k, float* __restrict__ pd, float* __restrict__ md,
                 float* __restrict__ vd, float ss_d, float beta1, float beta2, float eps) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = t >> 2;
  const int q = (int)(t & 3);
  if (row >= n_vox) return;
  const float4 g = reinterpret_cast<const float4*>(G)[row * 4 + q];
  if (q < 3) {
    if (MODE_K == 1 && g.x == 0.f && g.y == 0.f && g.z == 0.f && g.w == 0.f) return;
    const int64_t i = row * 3 + q;
    float4 p = reinterpret_cast<float4*>(pk)[i], m = reinterpret_cast<float4*>(mk)[i], v = reinterpret_cast<float4*>(vk)[i];
    adam_one<MODE_K>(p.x, g.x, m.x, v.x, 0.f, ss_k, beta1, beta2, eps);
    adam_one<MODE_K>(p.y, g.y, m.y, v.y, 0.f, ss_k, beta1, beta2, eps);
    adam_one<MODE_K>(p.z, g.z, m.z, v.z, 0.f, ss_k, beta1, beta2, eps);
    adam_one<MODE_K>(p.w, g.w, m.w, v.w, 0.f, ss_k, beta1, beta2, eps);
    reinterpret_cast<float4*>(pk)[i] = p;
    reinterpret_cast<float4*>(mk)[i] = m;
    reinterpret_cast<float4*>(vk)[i] = v;
  } else {
    adam_one<MODE_D>(pd[row], g.x, md[row], vd[row], 0.f, ss_d, beta1, beta2, eps);
  }
}

__device__ __forceinline__ float clamp1(float v) { return fminf(fmaxf(v, -1.f), 1.f); }

// One thread per grid element.  `cl` selects the thread -> element order so that consecutive
// lanes touch consecutive memory for both layouts.  The i axis uses wz (the reference's own
// quirk, total_variation_kernel.cu:31-32); wx is accepted and unused.
template <bool DENSE>
__global__ void __launch_bounds__(DVGO_BLOCK)
tv_kernel(const float* __restrict__ param, float* __restrict__ grad, float wy, float wz, int64_t C,
          int64_t I, int64_t J, int64_t K, int64_t sC, int64_t sI, int64_t sJ, int64_t sK, bool cl,
          int64_t N) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  int64_t c, i, j, k;
  if (cl) { c = t % C; k = t / C % K; j = t / C / K % J; i = t / C / K / J; }
  else    { k = t % K; j = t / K % J; i = t / K / J % I; c = t / K / J / I; }
  const int64_t o = c * sC + i * sI + j * sJ + k * sK;
  if (!DENSE && grad[o] == 0.0f) return;
  const float p = param[o];
  float g = 0.f;
  g += (k == 0     ? 0.f : wz * clamp1(p - param[o - sK]));
  g += (k == K - 1 ? 0.f : wz * clamp1(p - param[o + sK]));
  g += (j == 0     ? 0.f : wy * clamp1(p - param[o - sJ]));
  g += (j == J - 1 ? 0.f : wy * clamp1(p - param[o + sJ]));
  g += (i == 0     ? 0.f : wz * clamp1(p - param[o - sI]));
  g += (i == I - 1 ? 0.f : wz * clamp1(p - param[o + sI]));
  grad[o] += g;
}

// Row form of the same stencil for the two dense layouts (channels-last: a row = the K*C contiguous floats of one
// (i, j); channel-first: the K floats of one (c, i, j)).  One workgroup per row: the (c, i, j) decomposition and the
// j / i boundary tests are wave-uniform scalars, the k boundary is `e < sK` / `e >= R - sK`, so a thread does no
// integer division at all (the flat kernel spends ~300 instructions per element on 64-bit div/mod: 1.34 ms per
// step on the config-4 grids, 4x the traffic bound), and all seven accesses are contiguous across the wave.
template <bool DENSE>
__global__ void __launch_bounds__(DVGO_BLOCK)
tv_rows_kernel(const float* __restrict__ param, float* __restrict__ grad, float wy, float wz, int R, int sK,
               int I, int J, int64_t sC, int64_t sI, int64_t sJ, int row0) {
  const int row = row0 + blockIdx.x;
  const int ij = I * J;
  const int c = row / ij, rem = row - c * ij;
  const int i = rem / J, j = rem - i * J;
  const int64_t base = c * sC + i * sI + j * sJ;
  const float wjm = (j == 0) ? 0.f : wy, wjp = (j == J - 1) ? 0.f : wy;
  const float wim = (i == 0) ? 0.f : wz, wip = (i == I - 1) ? 0.f : wz;      // i axis weighted by wz (reference quirk)
  const int64_t ojm = (j == 0) ? 0 : -sJ, ojp = (j == J - 1) ? 0 : sJ;       // clamped: the loads stay in bounds
  const int64_t oim = (i == 0) ? 0 : -sI, oip = (i == I - 1) ? 0 : sI;
  const float* pr = param + base;
  float* gr = grad + base;
  for (int e = threadIdx.x; e < R; e += blockDim.x) {
    const float g0 = gr[e];
    if (!DENSE && g0 == 0.0f) continue;
    const float p = pr[e];
    float g = 0.f;
    g += (e < sK      ? 0.f : wz * clamp1(p - pr[e - sK]));
    g += (e >= R - sK ? 0.f : wz * clamp1(p - pr[e + sK]));
    g += wjm * clamp1(p - pr[e + ojm]);
    g += wjp * clamp1(p - pr[e + ojp]);
    g += wim * clamp1(p - pr[e + oim]);
    g += wip * clamp1(p - pr[e + oip]);
    gr[e] = g0 + g;
  }
}

extern "C" {

int dvgo_adam_upd(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                  const float* perlr, int64_t n, float step_size, float beta1, float beta2, float eps,
                  int mode, void* stream) {
  if (n < 0 || mode < 0 || mode > 2) return DVGO_EINVAL;
  if (n == 0) return 0;
  if (!param || !grad || !exp_avg || !exp_avg_sq || (mode == 2 && !perlr)) return DVGO_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const uintptr_t bits = (uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq |
                         (mode == 2 ? (uintptr_t)perlr : 0);
  const bool vec = (bits & 15) == 0;
  const int64_t threads = vec ? (n + 3) / 4 : n;
  if (!dvgo_fits(threads)) return DVGO_ERANGE;
  const int blocks = dvgo_blocks(threads, DVGO_BLOCK);
  if (mode == 0)
    adam_kernel<0><<<blocks, DVGO_BLOCK, 0, s>>>(param, grad, exp_avg, exp_avg_sq, perlr, n, step_size, beta1, beta2, eps, vec);
  else if (mode == 1)
    adam_kernel<1><<<blocks, DVGO_BLOCK, 0, s>>>(param, grad, exp_avg, exp_avg_sq, perlr, n, step_size, beta1, beta2, eps, vec);
  else
    adam_kernel<2><<<blocks, DVGO_BLOCK, 0, s>>>(param, grad, exp_avg, exp_avg_sq, perlr, n, step_size, beta1, beta2, eps, vec);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_adam_rows(const float* G, int64_t n_vox, int row_stride, int C, float* p_k0, float* m_k0, float* v_k0,
                   float step_size_k0, int mode_k0, float* p_density, float* m_density, float* v_density,
                   float step_size_density, int mode_density, float beta1, float beta2, float eps, void* stream) {
  if (n_vox < 0 || mode_k0 < 0 || mode_k0 > 1 || mode_density < 0 || mode_density > 1) return DVGO_EINVAL;
  if (n_vox == 0) return 0;
  if (!G || !p_k0 || !m_k0 || !v_k0 || !p_density || !m_density || !v_density) return DVGO_EINVAL;
  if (row_stride != 16 || C != 12) return DVGO_ERANGE;
  if ((((uintptr_t)G | (uintptr_t)p_k0 | (uintptr_t)m_k0 | (uintptr_t)v_k0) & 15) != 0) return DVGO_EINVAL;
  if (!dvgo_fits(n_vox * 4)) return DVGO_ERANGE;
  const int blocks = dvgo_blocks(n_vox * 4, DVGO_BLOCK);
  hipStream_t s = (hipStream_t)stream;
#define DVGO_ADAM_ROWS(MK, MD)                                                                                          \
  adam_rows_kernel<MK, MD><<<blocks, DVGO_BLOCK, 0, s>>>(G, n_vox, p_k0, m_k0, v_k0, step_size_k0, p_density, m_density, \
                                                         v_density, step_size_density, beta1, beta2, eps)
  if (mode_k0 == 1) { if (mode_density == 1) DVGO_ADAM_ROWS(1, 1); else DVGO_ADAM_ROWS(1, 0); }
  else              { if (mode_density == 1) DVGO_ADAM_ROWS(0, 1); else DVGO_ADAM_ROWS(0, 0); }
#undef DVGO_ADAM_ROWS
  DVGO_LAUNCH_CHECK();
  return 0;
}

// i_lo / i_hi: the planes [i_lo, i_hi) of the first spatial axis whose gradient is touched (the whole grid: 0, sz_i).  A
// data-parallel rank that owns one slab of the grid (train.py) adds the TV gradient of its slab only; the stencil still
// reads the neighbouring planes of `param`, which every rank holds.
int dvgo_total_variation_add_grad_slab(const float* param, float* grad, float wx, float wy, float wz,
                                       int64_t C, int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t sC,
                                       int64_t sI, int64_t sJ, int64_t sK, int dense_mode, int64_t i_lo, int64_t i_hi,
                                       void* stream) {
  (void)wx;
  if (C < 0 || sz_i < 0 || sz_j < 0 || sz_k < 0 || i_lo < 0 || i_hi > sz_i || i_lo > i_hi) return DVGO_EINVAL;
  const int64_t N = C * sz_i * sz_j * sz_k;
  if (N == 0 || i_lo == i_hi) return 0;
  if (!param || !grad) return DVGO_EINVAL;
  if (!dvgo_fits(N)) return DVGO_ERANGE;
  wy /= 6; wz /= 6;   // total_variation_kernel.cu:46-48
  const bool cl = (sC == 1 && C > 1);
  hipStream_t s = (hipStream_t)stream;
  const bool whole = (i_lo == 0 && i_hi == sz_i);
  // dense layouts: channels-last (sC == 1, sK == C) or channel-first (sK == 1), any C
  const bool rows_cl = (sC == 1 && sK == C && sJ == sz_k * C && sI == sz_j * sz_k * C);
  const bool rows_cf = (sK == 1 && sJ == sz_k && sI == sz_j * sz_k && (C == 1 || sC == sz_i * sz_j * sz_k));
  const int64_t n_rows = rows_cl ? sz_i * sz_j : C * sz_i * sz_j;
  const int64_t R = rows_cl ? sz_k * C : sz_k;
  if ((rows_cl || (rows_cf && (whole || C == 1))) && n_rows < ((int64_t)1 << 31) && R < ((int64_t)1 << 30)) {
    const int threads = R >= 256 ? 256 : (R > 128 ? 256 : (R > 64 ? 128 : 64));
    // rows are (i, j) pairs (channels-last, or C == 1), i outermost: a slab is a contiguous range of rows
    const int row0 = whole ? 0 : (int)(i_lo * sz_j);
    const int rows = whole ? (int)n_rows : (int)((i_hi - i_lo) * sz_j);
    if (dense_mode)
      tv_rows_kernel<true><<<rows, threads, 0, s>>>(param, grad, wy, wz, (int)R, (int)sK, (int)sz_i, (int)sz_j, sC, sI, sJ, row0);
    else
      tv_rows_kernel<false><<<rows, threads, 0, s>>>(param, grad, wy, wz, (int)R, (int)sK, (int)sz_i, (int)sz_j, sC, sI, sJ, row0);
    DVGO_LAUNCH_CHECK();
    return 0;
  }
  if (!whole) return DVGO_ERANGE;          // slabs are built for the row layouts above
  if (dense_mode)
    tv_kernel<true><<<dvgo_blocks(N, DVGO_BLOCK), DVGO_BLOCK, 0, s>>>(param, grad, wy, wz, C, sz_i, sz_j, sz_k, sC, sI, sJ, sK, cl, N);
  else
    tv_kernel<false><<<dvgo_blocks(N, DVGO_BLOCK), DVGO_BLOCK, 0, s>>>(param, grad, wy, wz, C, sz_i, sz_j, sz_k, sC, sI, sJ, sK, cl, N);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_total_variation_add_grad(const float* param, float* grad, float wx, float wy, float wz,
                                  int64_t C, int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t sC,
                                  int64_t sI, int64_t sJ, int64_t sK, int dense_mode, void* stream) {
  return dvgo_total_variation_add_grad_slab(param, grad, wx, wy, wz, C, sz_i, sz_j, sz_k, sC, sI, sJ, sK, dense_mode, 0,
                                            sz_i, stream);
}

}  // extern "C"
