// Small fused host-glue kernels around the march (harness row H3): the training loss of
// /root/reference/run.py:377-386 with its gradients in one pass, the view-direction embedding of
// lib/dvgo.py:524-525, and a multi-tensor Adam launch for the handful of small MLP tensors.
// They replace ~60 tiny framework launches per step; on sparse scenes the step is launch-bound.
#include "common.h"

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = v;
  __syncthreads();
  const float t = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return t;
}

// Per-ray terms.  loss_main = w_main * sum_{r,c} (rgb_marched - target)^2 / (3 N_global)
//                 loss_ent  = w_ent * sum_r -(p log p + (1-p) log(1-p)) / N_global,  p = clamp(T_last, 1e-6, 1-1e-6)
__global__ void __launch_bounds__(256)
loss_rays_kernel(const float* __restrict__ rgb_marched, const float* __restrict__ alphainv_last,
                 const float* __restrict__ target, int64_t N, float inv_n_global, float w_main, float w_ent,
                 float* __restrict__ g_marched, float* __restrict__ g_last, float* __restrict__ loss_out) {
  __shared__ float red[4];
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float l = 0.0f;
  if (r < N) {
    float se = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = rgb_marched[3 * r + c] - target[3 * r + c];
      se = fmaf(d, d, se);
      g_marched[3 * r + c] = 2.0f * w_main * d * inv_n_global * (1.0f / 3.0f);
    }
    l = w_main * se * inv_n_global * (1.0f / 3.0f);
    if (w_ent > 0.0f) {
      const float x = alphainv_last[r];
      const float p = fminf(fmaxf(x, 1e-6f), 1.0f - 1e-6f);
      const float lp = logf(p), lq = logf(1.0f - p);
      l += w_ent * (-(p * lp + (1.0f - p) * lq)) * inv_n_global;
      const bool inside = (x >= 1e-6f) && (x <= 1.0f - 1e-6f);        // clamp passes gradient on [min, max]
      g_last[r] = inside ? w_ent * (lq - lp) * inv_n_global : 0.0f;
    } else {
      g_last[r] = 0.0f;
    }
  }
  const float t = block_sum_256(l, red);
  if (threadIdx.x == 0) atomicAdd(loss_out, t);
}

// Per-sample term.  loss_per = w_per * sum_i weights_i * |raw_rgb_i - target[ray_id_i]|^2 / N_global
// (weights detached, run.py:385)
__global__ void __launch_bounds__(256)
loss_samples_kernel(const float* __restrict__ raw_rgb, const float* __restrict__ weights,
                    const int64_t* __restrict__ ray_id, const float* __restrict__ target, int64_t M_cap,
                    const int64_t* __restrict__ m_dev,
                    float inv_n_global, float w_per, float* __restrict__ g_raw_rgb, float* __restrict__ loss_out) {
  const int64_t M = m_dev ? (*m_dev < M_cap ? *m_dev : M_cap) : M_cap;      // sample count kept on the device (train.py)
  __shared__ float red[4];
  float l = 0.0f;
  // grid-stride: one same-address atomic per workgroup is the cost that matters here, so few, long-lived workgroups
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = ray_id[i];
    const float w = weights[i];
    float se = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = raw_rgb[3 * i + c] - target[3 * r + c];
      se = fmaf(d, d, se);
      g_raw_rgb[3 * i + c] = 2.0f * w_per * w * d * inv_n_global;
    }
    l += w_per * w * se * inv_n_global;
  }
  const float t = block_sum_256(l, red);
  if (threadIdx.x == 0) atomicAdd(loss_out, t);
}

// viewdirs_emb = cat([v, sin(v (x) freq), cos(v (x) freq)])  with (v (x) freq) flattened component-major
// (lib/dvgo.py:524-525) -> emb [N, 3 + 6F]
__global__ void __launch_bounds__(256)
viewdir_embed_kernel(const float* __restrict__ viewdirs, const float* __restrict__ freq, int F, int64_t N,
                     float* __restrict__ emb) {
  const int E = 3 + 6 * F;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * E) return;
  const int64_t r = t / E;
  const int e = (int)(t - r * E);
  float v;
  if (e < 3) v = viewdirs[3 * r + e];
  else {
    const int q = (e - 3) % (3 * F);
    const float a = viewdirs[3 * r + q / F] * freq[q % F];
    v = (e - 3 < 3 * F) ? sinf(a) : cosf(a);
  }
  emb[t] = v;
}

// Multi-tensor Adam (mode 0) for up to DVGO_MT_MAX small tensors in one launch.
#define DVGO_MT_MAX 16
struct AdamMulti {
  float* p[DVGO_MT_MAX];
  const float* g[DVGO_MT_MAX];
  float* m[DVGO_MT_MAX];
  float* v[DVGO_MT_MAX];
  int64_t start[DVGO_MT_MAX + 1];
  int n;
};
__global__ void __launch_bounds__(256)
adam_multi_kernel(AdamMulti A, float step_size, const float* __restrict__ step_size_dev, float beta1, float beta2, float eps) {
  if (step_size_dev != nullptr) step_size = *step_size_dev;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= A.start[A.n]) return;
  int k = 0;
  while (k + 1 < A.n && t >= A.start[k + 1]) ++k;
  const int64_t i = t - A.start[k];
  const float g = A.g[k][i];
  const float m = fmaf(beta1, A.m[k][i], (1.0f - beta1) * g);
  const float v = fmaf(beta2, A.v[k][i], ((1.0f - beta2) * g) * g);
  A.m[k][i] = m;
  A.v[k][i] = v;
  A.p[k][i] = A.p[k][i] - (step_size * m) / (sqrtf(v) + eps);
}

extern "C" {

int dvgo_loss_fwd_bwd(const float* rgb_marched, const float* alphainv_last, const float* target, int64_t N,
                      const float* raw_rgb, const float* weights, const int64_t* ray_id, int64_t M, const int64_t* m_dev,
                      int64_t n_rays_global, float w_main, float w_ent, float w_per, float* g_marched,
                      float* g_last, float* g_raw_rgb, float* loss_out, void* stream) {
  if (N < 0 || M < 0 || n_rays_global <= 0) return DVGO_EINVAL;
  if (!loss_out) return DVGO_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  DVGO_HIP_TRY(hipMemsetAsync(loss_out, 0, sizeof(float), s));
  const float inv = 1.0f / (float)n_rays_global;
  if (N > 0) {
    if (!rgb_marched || !alphainv_last || !target || !g_marched || !g_last) return DVGO_EINVAL;
    loss_rays_kernel<<<dvgo_blocks(N, 256), 256, 0, s>>>(rgb_marched, alphainv_last, target, N, inv, w_main, w_ent,
                                                         g_marched, g_last, loss_out);
    DVGO_LAUNCH_CHECK();
  }
  if (M > 0 && w_per > 0.0f) {
    if (!raw_rgb || !weights || !ray_id || !g_raw_rgb) return DVGO_EINVAL;
    const int64_t nb = dvgo_blocks(M, 256);
    loss_samples_kernel<<<(int)(nb < 1024 ? nb : 1024), 256, 0, s>>>(raw_rgb, weights, ray_id, target, M, m_dev, inv, w_per,
                                                            g_raw_rgb, loss_out);
    DVGO_LAUNCH_CHECK();
  }
  return 0;
}

int dvgo_viewdir_embed(const float* viewdirs, const float* freq, int n_freq, int64_t N, float* emb, void* stream) {
  if (N < 0 || n_freq < 0) return DVGO_EINVAL;
  if (N == 0) return 0;
  if (!viewdirs || !emb || (n_freq > 0 && !freq)) return DVGO_EINVAL;
  const int64_t total = N * (3 + 6 * n_freq);
  if (!dvgo_fits(total)) return DVGO_ERANGE;
  viewdir_embed_kernel<<<dvgo_blocks(total, 256), 256, 0, (hipStream_t)stream>>>(viewdirs, freq, n_freq, N, emb);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_adam_upd_multi(float* const* params, const float* const* grads, float* const* exp_avg,
                        float* const* exp_avg_sq, const int64_t* numel, int n_tensors, float step_size, float beta1,
                        float beta2, float eps, const float* step_size_dev, void* stream) {
  if (n_tensors < 0 || n_tensors > DVGO_MT_MAX) return DVGO_EINVAL;
  if (n_tensors == 0) return 0;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !numel) return DVGO_EINVAL;
  AdamMulti A;
  A.n = n_tensors;
  int64_t tot = 0;
  for (int k = 0; k < n_tensors; ++k) {
    if (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k] || numel[k] < 0) return DVGO_EINVAL;
    A.p[k] = params[k]; A.g[k] = grads[k]; A.m[k] = exp_avg[k]; A.v[k] = exp_avg_sq[k];
    A.start[k] = tot;
    tot += numel[k];
  }
  for (int k = n_tensors; k <= DVGO_MT_MAX; ++k) A.start[k] = tot;
  if (tot == 0) return 0;
  adam_multi_kernel<<<dvgo_blocks(tot, 256), 256, 0, (hipStream_t)stream>>>(A, step_size, step_size_dev, beta1, beta2, eps);
  DVGO_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
