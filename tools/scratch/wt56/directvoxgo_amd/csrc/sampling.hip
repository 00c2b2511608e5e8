// Ray-AABB intersection and sample-point generation (K1-K7).
// Reference semantics: /root/reference/lib/cuda/render_utils_kernel.cu:11-287.
#include "common.h"

// ----------------------------------------------------------------------------------
// K1 + K2 + K3 in one pass over the rays (the reference launches three kernels).
// Any output pointer may be null.
// ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(DVGO_BLOCK)
ray_setup_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                 const float* __restrict__ xyz_min, const float* __restrict__ xyz_max,
                 const float* __restrict__ t_min_in,   // used when only K2/K3 are wanted
                 const float* __restrict__ t_max_in,
                 float near, float far, float stepdist, int64_t n_rays,
                 float* __restrict__ t_min, float* __restrict__ t_max,
                 int64_t* __restrict__ n_steps,
                 float* __restrict__ rays_start, float* __restrict__ rays_dir) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  float tmin, tmax;
  float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;
  if (rays_o) { ox = rays_o[3 * r]; oy = rays_o[3 * r + 1]; oz = rays_o[3 * r + 2]; }
  if (rays_d) { dx = rays_d[3 * r]; dy = rays_d[3 * r + 1]; dz = rays_d[3 * r + 2]; }
  if (t_min_in) {
    tmin = t_min_in[r];
    tmax = t_max_in ? t_max_in[r] : tmin;
  } else {
    // K1 :23-33
    const float vx = (dx == 0) ? (float)1e-6 : dx;
    const float vy = (dy == 0) ? (float)1e-6 : dy;
    const float vz = (dz == 0) ? (float)1e-6 : dz;
    const float ax = (xyz_max[0] - ox) / vx, ay = (xyz_max[1] - oy) / vy, az = (xyz_max[2] - oz) / vz;
    const float bx = (xyz_min[0] - ox) / vx, by = (xyz_min[1] - oy) / vy, bz = (xyz_min[2] - oz) / vz;
    tmin = fmaxf(fminf(fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz)), far), near);
    tmax = fmaxf(fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)), far), near);
  }
  if (t_min) t_min[r] = tmin;
  if (t_max) t_max[r] = tmax;
  if (n_steps) {
    // K2 :47  max(ceil((t_max - t_min) / stepdist), 1.) -> int64
    const float c = ceilf((tmax - tmin) / stepdist);
    n_steps[r] = (int64_t)fmax((double)c, 1.);
  }
  if (rays_start) {
    // K3 :62-71
    const float rnorm = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
    rays_start[3 * r + 0] = fmaf(dx, tmin, ox);
    rays_start[3 * r + 1] = fmaf(dy, tmin, oy);
    rays_start[3 * r + 2] = fmaf(dz, tmin, oz);
    rays_dir[3 * r + 0] = dx / rnorm;
    rays_dir[3 * r + 1] = dy / rnorm;
    rays_dir[3 * r + 2] = dz / rnorm;
  }
}

// ----------------------------------------------------------------------------------
// Single-workgroup scans.  N is the ray count (8192 per training step, <= ~1M when
// pre-filtering training rays), so one 1024-thread workgroup walking tiles with a carry
// is latency-trivial and needs no inter-workgroup protocol.
// ----------------------------------------------------------------------------------
template <typename TIn, bool EXCLUSIVE>
__global__ void __launch_bounds__(1024)
scan_kernel(const TIn* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
  constexpr int ITEMS = 8;
  __shared__ int64_t wave_sums[16];
  __shared__ int64_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024 * ITEMS) {
    int64_t v[ITEMS];
    int64_t tsum = 0;
    const int64_t i0 = base + (int64_t)tid * ITEMS;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      v[k] = (i0 + k < n) ? (int64_t)in[i0 + k] : 0;
      tsum += v[k];
    }
    // inclusive scan of tsum across the wave
    int64_t inc = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int64_t o = __shfl_up(inc, d);
      if (lane >= d) inc += o;
    }
    if (lane == 63) wave_sums[wid] = inc;
    __syncthreads();
    int64_t wave_off = 0;
    for (int w = 0; w < wid; ++w) wave_off += wave_sums[w];
    const int64_t carry = carry_s;
    int64_t run = carry + wave_off + inc - tsum;   // exclusive prefix of this thread's first item
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      if (EXCLUSIVE) { if (i0 + k < n) out[i0 + k] = run; run += v[k]; }
      else           { run += v[k]; if (i0 + k < n) out[i0 + k] = run; }
    }
    __syncthreads();
    if (tid == 1023) carry_s = run;
    __syncthreads();
  }
  if (EXCLUSIVE && tid == 0) out[n] = carry_s;
}

// Long inputs (full-image ray chunks: 65536 rays and more) in three short launches instead of one workgroup walking
// the array (65 us at 65536): block totals -> scan of the totals -> per-block scan with its carry.  The totals live in
// the first output slot of each block's own range (read back as the carry before that range is written), so no
// scratch buffer is needed.
#define DVGO_SCAN_TILE 8192      // items per workgroup: 1024 threads x 8
template <typename TIn>
__global__ void __launch_bounds__(1024)
scan_block_totals_kernel(const TIn* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
  __shared__ int64_t wave_sums[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * DVGO_SCAN_TILE + (int64_t)tid * 8;
  int64_t t = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) t += (i0 + k < n) ? (int64_t)in[i0 + k] : 0;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
  if (lane == 0) wave_sums[wid] = t;
  __syncthreads();
  if (tid == 0) {
    int64_t tot = 0;
    for (int w = 0; w < 16; ++w) tot += wave_sums[w];
    out[(int64_t)blockIdx.x * DVGO_SCAN_TILE] = tot;
  }
}

// exclusive scan, in place, of the nb strided totals out[b * TILE]; the grand total goes to *total when given
__global__ void __launch_bounds__(1024)
scan_totals_kernel(int64_t* __restrict__ out, int64_t nb, int64_t* __restrict__ total) {
  __shared__ int64_t wave_sums[16];
  __shared__ int64_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < nb; base += 1024) {
    const int64_t b = base + tid;
    const int64_t v = (b < nb) ? out[b * DVGO_SCAN_TILE] : 0;
    int64_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int64_t o = __shfl_up(inc, d);
      if (lane >= d) inc += o;
    }
    if (lane == 63) wave_sums[wid] = inc;
    __syncthreads();
    int64_t wave_off = 0;
    for (int w = 0; w < wid; ++w) wave_off += wave_sums[w];
    const int64_t excl = carry_s + wave_off + inc - v;
    if (b < nb) out[b * DVGO_SCAN_TILE] = excl;
    __syncthreads();
    if (tid == 1023) carry_s = excl + v;
    __syncthreads();
  }
  if (total != nullptr && tid == 0) *total = carry_s;
}

template <typename TIn, bool EXCLUSIVE>
__global__ void __launch_bounds__(1024)
scan_blocks_kernel(const TIn* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
  __shared__ int64_t wave_sums[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * DVGO_SCAN_TILE;
  const int64_t carry = out[base];                  // this block's exclusive prefix, left there by scan_totals_kernel
  const int64_t i0 = base + (int64_t)tid * 8;
  int64_t v[8];
  int64_t tsum = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v[k] = (i0 + k < n) ? (int64_t)in[i0 + k] : 0;
    tsum += v[k];
  }
  int64_t inc = tsum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int64_t o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wave_sums[wid] = inc;
  __syncthreads();                                  // also: every thread has read `carry` before anyone writes out[base]
  int64_t wave_off = 0;
  for (int w = 0; w < wid; ++w) wave_off += wave_sums[w];
  int64_t run = carry + wave_off + inc - tsum;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (EXCLUSIVE) { if (i0 + k < n) out[i0 + k] = run; run += v[k]; }
    else           { run += v[k]; if (i0 + k < n) out[i0 + k] = run; }
  }
}

template <typename TIn, bool EXCLUSIVE>
static void launch_scan(const TIn* in, int64_t n, int64_t* out, hipStream_t s) {
  if (n <= 2 * DVGO_SCAN_TILE || (const void*)in == (const void*)out) {     // short (or in place): one workgroup
    scan_kernel<TIn, EXCLUSIVE><<<1, 1024, 0, s>>>(in, n, out);
    return;
  }
  const int64_t nb = (n + DVGO_SCAN_TILE - 1) / DVGO_SCAN_TILE;
  scan_block_totals_kernel<TIn><<<(int)nb, 1024, 0, s>>>(in, n, out);
  scan_totals_kernel<<<1, 1024, 0, s>>>(out, nb, EXCLUSIVE ? out + n : nullptr);
  scan_blocks_kernel<TIn, EXCLUSIVE><<<(int)nb, 1024, 0, s>>>(in, n, out);
}

// ----------------------------------------------------------------------------------
// K4 + K5 + K6 as one flat pass over the M0 samples: each sample finds its ray by
// binary search in the inclusive cumsum (lanes of a wave mostly share a ray, so the
// search loads are broadcasts), then writes ids, position and the out-of-box flag.
// ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(DVGO_BLOCK)
sample_fill_kernel(const float* __restrict__ rays_start, const float* __restrict__ rays_dir,
                   const float* __restrict__ xyz_min, const float* __restrict__ xyz_max,
                   const int64_t* __restrict__ cum, int64_t n_rays, float stepdist,
                   int64_t total, float* __restrict__ rays_pts, uint8_t* __restrict__ mask_outbbox,
                   int64_t* __restrict__ ray_id, int64_t* __restrict__ step_id) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t r = dvgo_upper_bound(cum, n_rays, idx);
  const int64_t s = idx - (r ? cum[r - 1] : 0);
  float px, py, pz;
  dvgo_sample_pos(rays_start, rays_dir, r, stepdist, (int)s, px, py, pz);
  ray_id[idx] = r;
  step_id[idx] = s;
  rays_pts[3 * idx + 0] = px;
  rays_pts[3 * idx + 1] = py;
  rays_pts[3 * idx + 2] = pz;
  mask_outbbox[idx] = (uint8_t)((xyz_min[0] > px) | (xyz_min[1] > py) | (xyz_min[2] > pz) |
                                (xyz_max[0] < px) | (xyz_max[1] < py) | (xyz_max[2] < pz));
}

// K7 :248-263
__global__ void __launch_bounds__(DVGO_BLOCK)
sample_ndc_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                  const float* __restrict__ xyz_min, const float* __restrict__ xyz_max,
                  int n_samples, int64_t total, float* __restrict__ rays_pts,
                  uint8_t* __restrict__ mask_outbbox) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t r = idx / n_samples;
  const int s = (int)(idx % n_samples);
  const float dist = ((float)s) / (float)(n_samples - 1);
  const float px = fmaf(rays_d[3 * r + 0], dist, rays_o[3 * r + 0]);
  const float py = fmaf(rays_d[3 * r + 1], dist, rays_o[3 * r + 1]);
  const float pz = fmaf(rays_d[3 * r + 2], dist, rays_o[3 * r + 2]);
  rays_pts[3 * idx + 0] = px;
  rays_pts[3 * idx + 1] = py;
  rays_pts[3 * idx + 2] = pz;
  mask_outbbox[idx] = (uint8_t)((xyz_min[0] > px) | (xyz_min[1] > py) | (xyz_min[2] > pz) |
                                (xyz_max[0] < px) | (xyz_max[1] < py) | (xyz_max[2] < pz));
}

// ----------------------------------------------------------------------------------
extern "C" {

int dvgo_abi_version(void) { return 2; }

int dvgo_infer_t_minmax(const float* rays_o, const float* rays_d, const float* xyz_min,
                        const float* xyz_max, float near, float far, int64_t n_rays,
                        float* t_min, float* t_max, void* stream) {
  if (n_rays < 0) return DVGO_EINVAL;
  if (n_rays == 0) return 0;
  if (!rays_o || !rays_d || !xyz_min || !xyz_max || !t_min || !t_max) return DVGO_EINVAL;
  if (!dvgo_fits(n_rays)) return DVGO_ERANGE;
  ray_setup_kernel<<<dvgo_blocks(n_rays, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(
      rays_o, rays_d, xyz_min, xyz_max, nullptr, nullptr, near, far, 1.f, n_rays, t_min, t_max,
      nullptr, nullptr, nullptr);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_infer_n_samples(const float* t_min, const float* t_max, float stepdist, int64_t n_rays,
                         int64_t* n_samples, void* stream) {
  if (n_rays < 0) return DVGO_EINVAL;
  if (n_rays == 0) return 0;
  if (!t_min || !t_max || !n_samples) return DVGO_EINVAL;
  if (!dvgo_fits(n_rays)) return DVGO_ERANGE;
  ray_setup_kernel<<<dvgo_blocks(n_rays, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(
      nullptr, nullptr, nullptr, nullptr, t_min, t_max, 0.f, 0.f, stepdist, n_rays, nullptr, nullptr,
      n_samples, nullptr, nullptr);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_infer_ray_start_dir(const float* rays_o, const float* rays_d, const float* t_min,
                             int64_t n_rays, float* rays_start, float* rays_dir, void* stream) {
  if (n_rays < 0) return DVGO_EINVAL;
  if (n_rays == 0) return 0;
  if (!rays_o || !rays_d || !t_min || !rays_start || !rays_dir) return DVGO_EINVAL;
  if (!dvgo_fits(n_rays)) return DVGO_ERANGE;
  ray_setup_kernel<<<dvgo_blocks(n_rays, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(
      rays_o, rays_d, nullptr, nullptr, t_min, nullptr, 0.f, 0.f, 1.f, n_rays, nullptr, nullptr,
      nullptr, rays_start, rays_dir);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_sample_pts_prepare(const float* rays_o, const float* rays_d, const float* xyz_min,
                            const float* xyz_max, float near, float far, float stepdist,
                            int64_t n_rays, float* t_min, float* t_max, int64_t* n_steps,
                            int64_t* n_steps_cumsum, float* rays_start, float* rays_dir,
                            void* stream) {
  if (n_rays < 0) return DVGO_EINVAL;
  if (n_rays == 0) return 0;
  if (!rays_o || !rays_d || !xyz_min || !xyz_max || !t_min || !t_max || !n_steps ||
      !rays_start || !rays_dir)
    return DVGO_EINVAL;
  if (!dvgo_fits(n_rays)) return DVGO_ERANGE;
  hipStream_t s = (hipStream_t)stream;
  ray_setup_kernel<<<dvgo_blocks(n_rays, DVGO_BLOCK), DVGO_BLOCK, 0, s>>>(
      rays_o, rays_d, xyz_min, xyz_max, nullptr, nullptr, near, far, stepdist, n_rays, t_min, t_max,
      n_steps, rays_start, rays_dir);
  DVGO_LAUNCH_CHECK();
  if (n_steps_cumsum) {   // NULL: caller uses fixed-stride scratch and does not need M0
    launch_scan<int64_t, false>(n_steps, n_rays, n_steps_cumsum, s);
    DVGO_LAUNCH_CHECK();
  }
  return 0;
}

int dvgo_sample_pts_fill(const float* rays_start, const float* rays_dir, const float* xyz_min,
                         const float* xyz_max, const int64_t* n_steps_cumsum, int64_t n_rays,
                         float stepdist, int64_t total_len, float* rays_pts, uint8_t* mask_outbbox,
                         int64_t* ray_id, int64_t* step_id, void* stream) {
  if (n_rays < 0 || total_len < 0) return DVGO_EINVAL;
  if (n_rays == 0 || total_len == 0) return 0;
  if (!rays_start || !rays_dir || !xyz_min || !xyz_max || !n_steps_cumsum || !rays_pts ||
      !mask_outbbox || !ray_id || !step_id)
    return DVGO_EINVAL;
  if (!dvgo_fits(total_len)) return DVGO_ERANGE;
  sample_fill_kernel<<<dvgo_blocks(total_len, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(
      rays_start, rays_dir, xyz_min, xyz_max, n_steps_cumsum, n_rays, stepdist, total_len, rays_pts,
      mask_outbbox, ray_id, step_id);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_sample_ndc_pts_on_rays(const float* rays_o, const float* rays_d, const float* xyz_min,
                                const float* xyz_max, int n_samples, int64_t n_rays,
                                float* rays_pts, uint8_t* mask_outbbox, void* stream) {
  if (n_rays < 0 || n_samples < 0) return DVGO_EINVAL;
  const int64_t total = n_rays * (int64_t)n_samples;
  if (total == 0) return 0;
  if (!rays_o || !rays_d || !xyz_min || !xyz_max || !rays_pts || !mask_outbbox) return DVGO_EINVAL;
  if (!dvgo_fits(total)) return DVGO_ERANGE;
  sample_ndc_kernel<<<dvgo_blocks(total, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(
      rays_o, rays_d, xyz_min, xyz_max, n_samples, total, rays_pts, mask_outbbox);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_exclusive_scan_i32(const int32_t* counts, int64_t n, int64_t* offsets, void* stream) {
  if (n < 0) return DVGO_EINVAL;
  if (!offsets) return DVGO_EINVAL;
  if (n == 0) { DVGO_HIP_TRY(hipMemsetAsync(offsets, 0, sizeof(int64_t), (hipStream_t)stream)); return 0; }
  if (!counts) return DVGO_EINVAL;
  launch_scan<int32_t, true>(counts, n, offsets, (hipStream_t)stream);
  DVGO_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
