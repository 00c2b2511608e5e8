// Front-to-back alpha compositing (K11-K13) and the per-ray segment sum (A9).
// Reference semantics: /root/reference/lib/cuda/render_utils_kernel.cu:430-561,
// torch_scatter.segment_coo(reduce='sum') as called at lib/dvgo.py:554-559.
//
// Mapping: the reference gives each RAY to one thread that walks its samples serially
// (8192 threads on the whole chip, lane stride = segment length).  Here each ray gets one
// 64-lane wavefront: the samples of a 64-chunk are loaded coalesced, the parts of the
// recurrence that do not depend on the carry are evaluated lane-parallel, and only the
// carry chain itself (T for the forward, back_cum for the backward) is walked in order with
// v_readlane.  Because the chain is evaluated in the reference's order and precision the
// outputs are bit-identical to the serial code.
#include "common.h"

// K11 :461-471 + :478-479 pre-fills + :489 (last segment end)
__global__ void __launch_bounds__(DVGO_BLOCK)
segment_bounds_kernel(const int64_t* __restrict__ ray_id, int64_t n_pts,
                      int64_t* __restrict__ i_start, int64_t* __restrict__ i_end) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pts) return;
  const int64_t r = ray_id[i];
  if (i > 0) {
    const int64_t rp = ray_id[i - 1];
    if (r != rp) { i_start[r] = i; i_end[rp] = i; }
  }
  if (i == n_pts - 1) i_end[r] = n_pts;
}

// K12 :440-458, one wavefront per ray.
__global__ void __launch_bounds__(DVGO_BLOCK)
alpha2weight_kernel(const float* __restrict__ alpha, int64_t n_rays, float* __restrict__ weight,
                    float* __restrict__ T, float* __restrict__ alphainv_last,
                    const int64_t* __restrict__ i_start, int64_t* __restrict__ i_end) {
  const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (ray >= n_rays) return;
  // the whole wave works on one ray: make the bounds provably wave-uniform (scalar loops)
  const int i_s = __builtin_amdgcn_readfirstlane((int)i_start[ray]);
  const int i_e_max = __builtin_amdgcn_readfirstlane((int)i_end[ray]);
  float Tc = 1.0f;          // wave-uniform carry
  int i_stop = i_e_max;
  for (int base = i_s; base < i_e_max; base += 64) {
    const int n = min(64, i_e_max - base);
    const bool act = lane < n;
    const float a = act ? alpha[base + lane] : 0.0f;
    // lane-parallel part: the double factor (1. - alpha + 1e-10)
    const double f = 1.0 - (double)a + 1e-10;
    float myT = 1.0f;       // T before this lane's sample
    float Tincl = 1.0f;     // T after this lane's sample
    for (int j = 0; j < n; ++j) {
      const double fj = dvgo_readlane_d(f, j);
      const float Tn = (float)((double)Tc * fj);
      if (lane == j) { myT = Tc; Tincl = Tn; }
      Tc = Tn;
    }
    // first lane whose running transmittance dropped below 1e-3 (double compare, :451)
    const unsigned long long stop = __ballot(act && ((double)Tincl < 1e-3));
    int cnt = n;
    if (stop) cnt = __ffsll((long long)stop);     // 1-based index of first set bit == j+1
    if (lane < cnt) {
      T[base + lane] = myT;
      weight[base + lane] = myT * a;
    } else if (act) {                              // after the break: pre-fill values (:478-479)
      T[base + lane] = 1.0f;
      weight[base + lane] = 0.0f;
    }
    if (stop) {
      i_stop = base + cnt;
      Tc = dvgo_readlane_f(Tincl, cnt - 1);
      // remaining chunks keep their pre-fill values
      for (int b2 = base + 64; b2 < i_e_max; b2 += 64) {
        if (b2 + lane < i_e_max) { T[b2 + lane] = 1.0f; weight[b2 + lane] = 0.0f; }
      }
      break;
    }
  }
  if (lane == 0) {
    i_end[ray] = i_stop;
    alphainv_last[ray] = Tc;
  }
}

// K13 :521-530, one wavefront per ray, chunks walked from the far end.
__global__ void __launch_bounds__(DVGO_BLOCK)
alpha2weight_backward_kernel(const float* __restrict__ alpha, const float* __restrict__ weight,
                             const float* __restrict__ T, const float* __restrict__ alphainv_last,
                             const int64_t* __restrict__ i_start, const int64_t* __restrict__ i_end,
                             int64_t n_rays, const float* __restrict__ grad_weights,
                             const float* __restrict__ grad_last, float* __restrict__ grad) {
  const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (ray >= n_rays) return;
  const int i_s = __builtin_amdgcn_readfirstlane((int)i_start[ray]);
  const int i_e = __builtin_amdgcn_readfirstlane((int)i_end[ray]);
  float acc = grad_last[ray] * alphainv_last[ray];   // wave-uniform back_cum
  for (int hi = i_e; hi > i_s; hi -= 64) {
    const int lo = max(i_s, hi - 64);
    const int n = hi - lo;
    const bool act = lane < n;
    const float a = act ? alpha[lo + lane] : 0.0f;
    const float w = act ? weight[lo + lane] : 0.0f;
    const float t = act ? T[lo + lane] : 0.0f;
    const float gw = act ? grad_weights[lo + lane] : 0.0f;
    float my_acc = 0.0f;    // back_cum as seen by this lane's sample
    for (int j = n - 1; j >= 0; --j) {
      const float gwj = dvgo_readlane_f(gw, j);
      const float wj = dvgo_readlane_f(w, j);
      if (lane == j) my_acc = acc;
      acc = fmaf(gwj, wj, acc);
    }
    if (act) {
      const float gt = gw * t;
      const float one_minus = 1.0f - a;
      grad[lo + lane] = (float)((double)gt - (double)my_acc / ((double)one_minus + 1e-10));
    }
  }
}

__global__ void __launch_bounds__(DVGO_BLOCK)
fill_f32_kernel(float* __restrict__ p, float v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ----------------------------------------------------------------------------------
// segment_coo(reduce='sum'): one lane per source row, segmented inclusive scan by key
// inside the wave, one atomic per (segment, wave, channel).  Rays spanning several waves
// are combined by the atomics (float add, order not reproducible -- torch_scatter's own
// order is unspecified as well).
// ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(DVGO_BLOCK)
segment_sum_kernel(const float* __restrict__ src, const int64_t* __restrict__ index, int64_t M,
                   int C, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool act = i < M;
  const int64_t key = act ? index[i] : -1;
  const int64_t key_next = __shfl_down(key, 1);
  const bool tail = act && (lane == 63 || key_next != key);
  for (int c = 0; c < C; ++c) {
    float v = act ? src[i * C + c] : 0.0f;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float o = __shfl_up(v, d);
      const int64_t ko = __shfl_up(key, d);
      if (lane >= d && ko == key) v += o;
    }
    if (tail) atomicAdd(out + key * C + c, v);
  }
}

extern "C" {

int dvgo_alpha2weight(const float* alpha, const int64_t* ray_id, int64_t n_pts, int64_t n_rays,
                      float* weight, float* T, float* alphainv_last, int64_t* i_start,
                      int64_t* i_end, void* stream) {
  if (n_pts < 0 || n_rays < 0) return DVGO_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (n_rays > 0) {
    if (!alphainv_last || !i_start || !i_end) return DVGO_EINVAL;
    if (!dvgo_fits(n_rays * 64)) return DVGO_ERANGE;
    DVGO_HIP_TRY(hipMemsetAsync(i_start, 0, sizeof(int64_t) * n_rays, s));
    DVGO_HIP_TRY(hipMemsetAsync(i_end, 0, sizeof(int64_t) * n_rays, s));
  }
  if (n_pts == 0) {   // :483 early return: weights empty, alphainv_last = 1
    if (n_rays > 0) {
      fill_f32_kernel<<<dvgo_blocks(n_rays, DVGO_BLOCK), DVGO_BLOCK, 0, s>>>(alphainv_last, 1.0f, n_rays);
      DVGO_LAUNCH_CHECK();
    }
    return 0;
  }
  if (n_rays == 0) return DVGO_EINVAL;   // points without rays
  if (!alpha || !ray_id || !weight || !T) return DVGO_EINVAL;
  if (!dvgo_fits(n_pts)) return DVGO_ERANGE;
  segment_bounds_kernel<<<dvgo_blocks(n_pts, DVGO_BLOCK), DVGO_BLOCK, 0, s>>>(ray_id, n_pts, i_start, i_end);
  DVGO_LAUNCH_CHECK();
  alpha2weight_kernel<<<dvgo_blocks(n_rays * 64, DVGO_BLOCK), DVGO_BLOCK, 0, s>>>(
      alpha, n_rays, weight, T, alphainv_last, i_start, i_end);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_alpha2weight_backward(const float* alpha, const float* weight, const float* T,
                               const float* alphainv_last, const int64_t* i_start,
                               const int64_t* i_end, int64_t n_rays, int64_t n_pts,
                               const float* grad_weights, const float* grad_last, float* grad,
                               void* stream) {
  if (n_pts < 0 || n_rays < 0) return DVGO_EINVAL;
  if (n_pts == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (!grad) return DVGO_EINVAL;
  DVGO_HIP_TRY(hipMemsetAsync(grad, 0, sizeof(float) * n_pts, s));   // :538 zeros_like
  if (n_rays == 0) return 0;                                           // :539
  if (!alpha || !weight || !T || !alphainv_last || !i_start || !i_end || !grad_weights || !grad_last)
    return DVGO_EINVAL;
  if (!dvgo_fits(n_rays * 64)) return DVGO_ERANGE;
  alpha2weight_backward_kernel<<<dvgo_blocks(n_rays * 64, DVGO_BLOCK), DVGO_BLOCK, 0, s>>>(
      alpha, weight, T, alphainv_last, i_start, i_end, n_rays, grad_weights, grad_last, grad);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_segment_sum(const float* src, const int64_t* index, int64_t M, int C, int64_t N,
                     float* out, void* stream) {
  if (M < 0 || C < 0 || N < 0) return DVGO_EINVAL;
  if (M == 0 || C == 0) return 0;
  if (!src || !index || !out || N == 0) return DVGO_EINVAL;
  if (!dvgo_fits(M)) return DVGO_ERANGE;
  segment_sum_kernel<<<dvgo_blocks(M, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(src, index, M, C, out);
  DVGO_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
