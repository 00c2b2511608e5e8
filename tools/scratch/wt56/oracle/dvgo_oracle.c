/*
 * dvgo_oracle.c -- CPU oracle for the DirectVoxGO volumetric ray-marching hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker.  The product path (directvoxgo_amd/) never links or
 * imports this file and fails loudly when its HIP library is missing.
 *
 * What it is: a scalar, single-threaded C restatement of the arithmetic the
 * reference specifies in /root/reference (paths below are relative to that tree):
 *
 *   lib/cuda/render_utils_kernel.cu   K1..K13 (sampling, mask lookup, activation,
 *                                     compositing and their backward passes)
 *   lib/dvgo.py:312-328               grid_sampler (coordinate normalisation + flip)
 *   torch F.grid_sample               trilinear, align_corners=True, zero padding
 *                                     (third-party: PyTorch, un-pinned by the reference;
 *                                     restated from ATen/native/GridSampler.{h,cpp}
 *                                     as shipped with torch 2.10 in this image)
 *   torch_scatter.segment_coo         reduce='sum' (third-party, absent from the image;
 *                                     restated as out[index[i]] += src[i])
 *   lib/cuda/adam_upd_kernel.cu       K15..K17
 *   lib/cuda/total_variation_kernel.cu K14
 *
 * Precision rules: every operation is written with the float/double promotion the
 * CUDA source specifies through its literal and variable types (see each function).
 * The file must be compiled with -ffp-contract=off; a*b+c is fused only where an
 * explicit fmaf() is written, which is where nvcc (-fmad=true, its default) contracts
 * the reference expression.  The HIP kernels make the same explicit choices, so
 * everything that does not go through libm (expf/powf) is expected to be bit-identical
 * between this oracle and the device.
 *
 * Pinning status (see DESIGN.md "Oracle"): the reference holds no tests or golden
 * vectors and its native path is CUDA-only (not buildable/runnable here), so this
 * oracle is pinned against (a) the reference's own importable PyTorch fragments
 * (grid_sampler, the PyTorch slab-test/sampler restatements, the softplus form of the
 * activation, MaskCache's affine map) run in the build container, (b) torch's CPU
 * F.grid_sample forward/backward, and (c) analytic identities.  The committed
 * fixtures in tests/golden/ hold those outputs.  Ulp-level agreement with the CUDA
 * binary itself is "parity unpinned".
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORA_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* K1  infer_t_minmax        lib/cuda/render_utils_kernel.cu:11-35           */
/* float locals; zero direction component replaced by (float)1e-6;           */
/* clamp order max(min(x, far), near).                                       */
/* ------------------------------------------------------------------------- */
ORA_API void ora_infer_t_minmax(const float* rays_o, const float* rays_d,
                                const float* xyz_min, const float* xyz_max,
                                float near, float far, int64_t n_rays,
                                float* t_min, float* t_max) {
  for (int64_t r = 0; r < n_rays; ++r) {
    const float* o = rays_o + 3 * r;
    const float* d = rays_d + 3 * r;
    float vx = (d[0] == 0) ? (float)1e-6 : d[0];
    float vy = (d[1] == 0) ? (float)1e-6 : d[1];
    float vz = (d[2] == 0) ? (float)1e-6 : d[2];
    float ax = (xyz_max[0] - o[0]) / vx;
    float ay = (xyz_max[1] - o[1]) / vy;
    float az = (xyz_max[2] - o[2]) / vz;
    float bx = (xyz_min[0] - o[0]) / vx;
    float by = (xyz_min[1] - o[1]) / vy;
    float bz = (xyz_min[2] - o[2]) / vz;
    t_min[r] = fmaxf(fminf(fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz)), far), near);
    t_max[r] = fmaxf(fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)), far), near);
  }
}

/* ------------------------------------------------------------------------- */
/* K2  infer_n_samples       render_utils_kernel.cu:37-49                    */
/* n = (int64) max((double) ceilf((t_max - t_min) / stepdist), 1.)           */
/* ------------------------------------------------------------------------- */
ORA_API void ora_infer_n_samples(const float* t_min, const float* t_max, float stepdist,
                                 int64_t n_rays, int64_t* n_samples) {
  for (int64_t r = 0; r < n_rays; ++r) {
    float c = ceilf((t_max[r] - t_min[r]) / stepdist);
    double m = fmax((double)c, 1.);
    n_samples[r] = (int64_t)m;
  }
}

/* ------------------------------------------------------------------------- */
/* K3  infer_ray_start_dir   render_utils_kernel.cu:51-73                    */
/* start = o + d*t_min (contracted by nvcc -> fmaf); dir = d / |d|           */
/* |d| = sqrtf(dx*dx + dy*dy + dz*dz), sum contracted left to right.         */
/* ------------------------------------------------------------------------- */
ORA_API void ora_infer_ray_start_dir(const float* rays_o, const float* rays_d, const float* t_min,
                                     int64_t n_rays, float* rays_start, float* rays_dir) {
  for (int64_t r = 0; r < n_rays; ++r) {
    const float* o = rays_o + 3 * r;
    const float* d = rays_d + 3 * r;
    float rnorm = sqrtf(fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0])));
    for (int a = 0; a < 3; ++a) {
      rays_start[3 * r + a] = fmaf(d[a], t_min[r], o[a]);
      rays_dir[3 * r + a] = d[a] / rnorm;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* A1  sample_pts_on_rays    render_utils_kernel.cu:138-236 (K4, K5, K6)     */
/* Two entry points because the output length is data dependent:             */
/*   ora_sample_pts_prepare : K1+K2+K3 + inclusive cumsum, returns M0        */
/*   ora_sample_pts_fill    : ids + points + out-of-box mask                 */
/* ------------------------------------------------------------------------- */
ORA_API int64_t ora_sample_pts_prepare(const float* rays_o, const float* rays_d,
                                       const float* xyz_min, const float* xyz_max,
                                       float near, float far, float stepdist, int64_t n_rays,
                                       float* t_min, float* t_max, int64_t* n_steps,
                                       int64_t* n_steps_cumsum, float* rays_start, float* rays_dir) {
  ora_infer_t_minmax(rays_o, rays_d, xyz_min, xyz_max, near, far, n_rays, t_min, t_max);
  ora_infer_n_samples(t_min, t_max, stepdist, n_rays, n_steps);
  int64_t acc = 0;
  for (int64_t r = 0; r < n_rays; ++r) { acc += n_steps[r]; n_steps_cumsum[r] = acc; }
  ora_infer_ray_start_dir(rays_o, rays_d, t_min, n_rays, rays_start, rays_dir);
  return acc;
}

ORA_API void ora_sample_pts_fill(const float* rays_start, const float* rays_dir,
                                 const float* xyz_min, const float* xyz_max,
                                 const int64_t* n_steps, int64_t n_rays, float stepdist,
                                 float* rays_pts, uint8_t* mask_outbbox,
                                 int64_t* ray_id, int64_t* step_id) {
  int64_t idx = 0;
  for (int64_t r = 0; r < n_rays; ++r) {
    for (int64_t s = 0; s < n_steps[r]; ++s, ++idx) {
      ray_id[idx] = r;
      step_id[idx] = s;
      /* K6 :172-187  dist = stepdist * i_step (int -> float), p = start + dir*dist (fma) */
      const int i_step = (int)s;
      const float dist = stepdist * (float)i_step;
      const float px = fmaf(rays_dir[3 * r + 0], dist, rays_start[3 * r + 0]);
      const float py = fmaf(rays_dir[3 * r + 1], dist, rays_start[3 * r + 1]);
      const float pz = fmaf(rays_dir[3 * r + 2], dist, rays_start[3 * r + 2]);
      rays_pts[3 * idx + 0] = px;
      rays_pts[3 * idx + 1] = py;
      rays_pts[3 * idx + 2] = pz;
      mask_outbbox[idx] = (uint8_t)((xyz_min[0] > px) | (xyz_min[1] > py) | (xyz_min[2] > pz) |
                                    (xyz_max[0] < px) | (xyz_max[1] < py) | (xyz_max[2] < pz));
    }
  }
}

/* ------------------------------------------------------------------------- */
/* A2  sample_ndc_pts_on_rays   render_utils_kernel.cu:238-287 (K7)          */
/* dist = (float)step / (N_samples-1); p = o + d*dist (fma); un-normalised d */
/* ------------------------------------------------------------------------- */
ORA_API void ora_sample_ndc_pts_on_rays(const float* rays_o, const float* rays_d,
                                        const float* xyz_min, const float* xyz_max,
                                        int n_samples, int64_t n_rays,
                                        float* rays_pts, uint8_t* mask_outbbox) {
  for (int64_t r = 0; r < n_rays; ++r) {
    for (int s = 0; s < n_samples; ++s) {
      const int64_t idx = r * n_samples + s;
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
  }
}

/* ------------------------------------------------------------------------- */
/* A3  maskcache_lookup      render_utils_kernel.cu:300-351 (K8)             */
/* ijk = (int) roundf(xyz*scale + shift)  (fma; half away from zero);        */
/* out of range -> false (output is zero-initialised, :332)                  */
/* ------------------------------------------------------------------------- */
ORA_API void ora_maskcache_lookup(const uint8_t* world, const float* xyz,
                                  const float* scale, const float* shift,
                                  int sz_i, int sz_j, int sz_k, int64_t n_pts, uint8_t* out) {
  for (int64_t p = 0; p < n_pts; ++p) {
    const int i = (int)roundf(fmaf(xyz[3 * p + 0], scale[0], shift[0]));
    const int j = (int)roundf(fmaf(xyz[3 * p + 1], scale[1], shift[1]));
    const int k = (int)roundf(fmaf(xyz[3 * p + 2], scale[2], shift[2]));
    uint8_t v = 0;
    if (0 <= i && i < sz_i && 0 <= j && j < sz_j && 0 <= k && k < sz_k)
      v = world[(int64_t)i * sz_j * sz_k + (int64_t)j * sz_k + k];
    out[p] = v;
  }
}

/* ------------------------------------------------------------------------- */
/* A5  raw2alpha / raw2alpha_backward  render_utils_kernel.cu:357-428        */
/* e = expf(d + shift) (may be inf); alpha = 1 - powf(1 + e, -interval)      */
/* bwd: (float)( min((double)e, 1e10) * (double)powf(1+e, -interval-1)       */
/*               * interval * g )                                            */
/* ------------------------------------------------------------------------- */
ORA_API void ora_raw2alpha(const float* density, float shift, float interval, int64_t n,
                           float* exp_d, float* alpha) {
  for (int64_t i = 0; i < n; ++i) {
    const float e = expf(density[i] + shift);
    exp_d[i] = e;
    alpha[i] = 1 - powf(1 + e, -interval);
  }
}

ORA_API void ora_raw2alpha_backward(const float* exp_d, const float* grad_back, float interval,
                                    int64_t n, float* grad) {
  for (int64_t i = 0; i < n; ++i) {
    const float e = exp_d[i];
    double v = fmin((double)e, 1e10) * (double)powf(1 + e, -interval - 1);
    v = v * (double)interval;
    v = v * (double)grad_back[i];
    grad[i] = (float)v;
  }
}

/* ------------------------------------------------------------------------- */
/* A6  alpha2weight          render_utils_kernel.cu:430-505 (K11, K12)       */
/* ------------------------------------------------------------------------- */
ORA_API void ora_alpha2weight(const float* alpha, const int64_t* ray_id, int64_t n_pts,
                              int64_t n_rays, float* weight, float* T, float* alphainv_last,
                              int64_t* i_start, int64_t* i_end) {
  for (int64_t i = 0; i < n_pts; ++i) { weight[i] = 0.f; T[i] = 1.f; }          /* :478-479 */
  for (int64_t r = 0; r < n_rays; ++r) { alphainv_last[r] = 1.f; i_start[r] = 0; i_end[r] = 0; }
  if (n_pts == 0) return;                                                        /* :483 */
  for (int64_t i = 1; i < n_pts; ++i) {                                          /* K11 :461-471 */
    if (ray_id[i] != ray_id[i - 1]) { i_start[ray_id[i]] = i; i_end[ray_id[i - 1]] = i; }
  }
  i_end[ray_id[n_pts - 1]] = n_pts;                                              /* :489 */
  for (int64_t r = 0; r < n_rays; ++r) {                                         /* K12 :440-458 */
    const int i_s = (int)i_start[r];
    const int i_e_max = (int)i_end[r];
    float T_cum = 1.f;
    int i;
    for (i = i_s; i < i_e_max; ++i) {
      T[i] = T_cum;
      weight[i] = T_cum * alpha[i];
      T_cum = (float)((double)T_cum * (1. - (double)alpha[i] + 1e-10));
      if ((double)T_cum < 1e-3) { i += 1; break; }
    }
    i_end[r] = i;
    alphainv_last[r] = T_cum;
  }
}

/* ------------------------------------------------------------------------- */
/* A7  alpha2weight_backward  render_utils_kernel.cu:507-561 (K13)           */
/* grad[i] = (float)( (double)(g_w[i]*T[i]) - (double)acc /                  */
/*                    ((double)(1 - alpha[i]) + 1e-10) );  acc += g_w[i]*w[i]*/
/* (1 - alpha[i]) is evaluated in float: int 1 promotes to scalar_t.         */
/* ------------------------------------------------------------------------- */
ORA_API void ora_alpha2weight_backward(const float* alpha, const float* weight, const float* T,
                                       const float* alphainv_last, const int64_t* i_start,
                                       const int64_t* i_end, int64_t n_rays, int64_t n_pts,
                                       const float* grad_weights, const float* grad_last,
                                       float* grad) {
  for (int64_t i = 0; i < n_pts; ++i) grad[i] = 0.f;                             /* :538 */
  for (int64_t r = 0; r < n_rays; ++r) {
    const int i_s = (int)i_start[r];
    const int i_e = (int)i_end[r];
    float back_cum = grad_last[r] * alphainv_last[r];
    for (int i = i_e - 1; i >= i_s; --i) {
      const float gt = grad_weights[i] * T[i];
      const float one_minus = 1 - alpha[i];
      grad[i] = (float)((double)gt - (double)back_cum / ((double)one_minus + 1e-10));
      back_cum = back_cum + grad_weights[i] * weight[i];   /* nvcc: fma */
    }
  }
}
/* note on the last line: `back_cum += g*w` is a float a*b+c which nvcc contracts; the
 * explicit-fma variant is used by ora_alpha2weight_backward_fma below and is the one the
 * HIP kernel matches.  Both are exported so tests can bound the difference. */
ORA_API void ora_alpha2weight_backward_fma(const float* alpha, const float* weight, const float* T,
                                           const float* alphainv_last, const int64_t* i_start,
                                           const int64_t* i_end, int64_t n_rays, int64_t n_pts,
                                           const float* grad_weights, const float* grad_last,
                                           float* grad) {
  for (int64_t i = 0; i < n_pts; ++i) grad[i] = 0.f;
  for (int64_t r = 0; r < n_rays; ++r) {
    const int i_s = (int)i_start[r];
    const int i_e = (int)i_end[r];
    float back_cum = grad_last[r] * alphainv_last[r];
    for (int i = i_e - 1; i >= i_s; --i) {
      const float gt = grad_weights[i] * T[i];
      const float one_minus = 1 - alpha[i];
      grad[i] = (float)((double)gt - (double)back_cum / ((double)one_minus + 1e-10));
      back_cum = fmaf(grad_weights[i], weight[i], back_cum);
    }
  }
}

/* ------------------------------------------------------------------------- */
/* A4/A8  grid_sampler  lib/dvgo.py:312-328 + F.grid_sample(bilinear,        */
/*        align_corners=True, padding zeros) -- ATen GridSampler.h:27-36     */
/*        (unnormalise) and GridSampler.cpp grid_sampler_3d_cpu_impl         */
/*        (corner order tnw,tne,tsw,tse,bnw,bne,bsw,bse; weights as          */
/*        differences to the opposite corner).                               */
/* grid is addressed with element strides so that both the reference layout  */
/* [1,C,X,Y,Z] and a channels-last layout can be checked.                    */
/* xyz -> u = (p - min)/(max - min); c = u*2 - 1; g = ((c + 1)/2)*(size-1).  */
/* The .flip(-1) of the reference only re-orders (x,y,z) into grid_sample's  */
/* (W,H,D) argument order: x indexes X (dim 2), y -> Y, z -> Z (innermost).  */
/* ------------------------------------------------------------------------- */
static inline float ora_src_index(float p, float mn, float mx, int size) {
  const float u = (p - mn) / (mx - mn);
  const float c = u * 2 - 1;
  return ((c + 1) / 2) * (float)(size - 1);
}

typedef struct { int64_t off[8]; float w[8]; int ok[8]; } ora_corners_t;

static inline void ora_corners(const float* p, const float* xyz_min, const float* xyz_max,
                               int X, int Y, int Z, int64_t sX, int64_t sY, int64_t sZ,
                               ora_corners_t* c) {
  const float gx = ora_src_index(p[0], xyz_min[0], xyz_max[0], X);
  const float gy = ora_src_index(p[1], xyz_min[1], xyz_max[1], Y);
  const float gz = ora_src_index(p[2], xyz_min[2], xyz_max[2], Z);
  const int64_t i0 = (int64_t)floorf(gx), j0 = (int64_t)floorf(gy), k0 = (int64_t)floorf(gz);
  const int64_t i1 = i0 + 1, j1 = j0 + 1, k1 = k0 + 1;
  /* ATen names: ix <-> gz (W), iy <-> gy (H), iz <-> gx (D) */
  const float wx0 = (float)i1 - gx, wx1 = gx - (float)i0;
  const float wy0 = (float)j1 - gy, wy1 = gy - (float)j0;
  const float wz0 = (float)k1 - gz, wz1 = gz - (float)k0;
  /* weight = (ix term) * (iy term) * (iz term), evaluated left to right */
  const float wz[2] = {wz0, wz1}, wy[2] = {wy0, wy1}, wx[2] = {wx0, wx1};
  const int64_t ii[2] = {i0, i1}, jj[2] = {j0, j1}, kk[2] = {k0, k1};
  int n = 0;
  for (int a = 0; a < 2; ++a)        /* D / X : t,b  */
    for (int b = 0; b < 2; ++b)      /* H / Y : n,s  */
      for (int d = 0; d < 2; ++d) {  /* W / Z : w,e  */
        c->w[n] = (wz[d] * wy[b]) * wx[a];
        c->ok[n] = (ii[a] >= 0 && ii[a] < X && jj[b] >= 0 && jj[b] < Y && kk[d] >= 0 && kk[d] < Z);
        /* out-of-range corners (incl. NaN / inf coordinates) are never dereferenced: keep their offset defined */
        c->off[n] = c->ok[n] ? ii[a] * sX + jj[b] * sY + kk[d] * sZ : 0;
        ++n;
      }
}

/* use_fma=1: out += v*w contracted (what nvcc emits for the CUDA grid_sampler, and what the
 * HIP kernels do); use_fma=0: separate multiply and add (bit-identical to torch's CPU
 * kernel in this image, used to pin the corner/weight logic exactly). */
ORA_API void ora_grid_sample_fwd(const float* grid, int C, int X, int Y, int Z,
                                 int64_t sC, int64_t sX, int64_t sY, int64_t sZ,
                                 const float* xyz, const float* xyz_min, const float* xyz_max,
                                 int64_t M, int use_fma, float* out /* [M,C] */) {
  ora_corners_t c;
  for (int64_t m = 0; m < M; ++m) {
    ora_corners(xyz + 3 * m, xyz_min, xyz_max, X, Y, Z, sX, sY, sZ, &c);
    for (int ch = 0; ch < C; ++ch) {
      float acc = 0.f;
      for (int n = 0; n < 8; ++n) {
        if (!c.ok[n]) continue;
        const float v = grid[ch * sC + c.off[n]];
        if (use_fma) acc = fmaf(v, c.w[n], acc);
        else { const float t = v * c.w[n]; acc = acc + t; }
      }
      out[m * C + ch] = acc;
    }
  }
}

ORA_API void ora_grid_sample_bwd(const float* grad_out /* [M,C] */, int C, int X, int Y, int Z,
                                 int64_t sC, int64_t sX, int64_t sY, int64_t sZ,
                                 const float* xyz, const float* xyz_min, const float* xyz_max,
                                 int64_t M, float* grad_grid /* accumulated into */) {
  ora_corners_t c;
  for (int64_t m = 0; m < M; ++m) {
    ora_corners(xyz + 3 * m, xyz_min, xyz_max, X, Y, Z, sX, sY, sZ, &c);
    for (int ch = 0; ch < C; ++ch) {
      const float g = grad_out[m * C + ch];
      for (int n = 0; n < 8; ++n)
        if (c.ok[n]) grad_grid[ch * sC + c.off[n]] += c.w[n] * g;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* A9  segment_coo(reduce='sum')  lib/dvgo.py:554-559,571-575                */
/* ------------------------------------------------------------------------- */
ORA_API void ora_segment_sum(const float* src, const int64_t* index, int64_t M, int C,
                             int64_t N, float* out /* [N,C], accumulated into */) {
  (void)N;
  for (int64_t i = 0; i < M; ++i)
    for (int c = 0; c < C; ++c) out[index[i] * C + c] += src[i * C + c];
}

/* ------------------------------------------------------------------------- */
/* K15-K17  Adam updates     lib/cuda/adam_upd_kernel.cu:8-58, host :60-132  */
/* step_size = lr * sqrt(1 - pow(beta2,(float)step)) / (1 - pow(beta1,...))  */
/* on the host with float arguments -> C++ float overloads (:72,:96,:121).   */
/* mode 0 = adam_upd, 1 = masked_adam_upd (skip grad==0), 2 = with per-lr.   */
/* ------------------------------------------------------------------------- */
ORA_API void ora_adam_upd(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                          const float* perlr, int64_t n, int step, float beta1, float beta2,
                          float lr, float eps, int mode) {
  const float step_size = lr * sqrtf(1 - powf(beta2, (float)step)) / (1 - powf(beta1, (float)step));
  for (int64_t i = 0; i < n; ++i) {
    if (mode == 1 && grad[i] == 0) continue;
    exp_avg[i] = fmaf(beta1, exp_avg[i], (1 - beta1) * grad[i]);
    exp_avg_sq[i] = fmaf(beta2, exp_avg_sq[i], (1 - beta2) * grad[i] * grad[i]);
    const float ss = (mode == 2) ? step_size * perlr[i] : step_size;
    param[i] -= ss * exp_avg[i] / (sqrtf(exp_avg_sq[i]) + eps);
  }
}

/* ------------------------------------------------------------------------- */
/* K14  total_variation_add_grad  lib/cuda/total_variation_kernel.cu:13-67   */
/* Faithful to the reference including its use of wz on the i axis (:31-32); */
/* wx is accepted and unused.  Index math assumes the contiguous             */
/* [1,C,X,Y,Z] layout of the reference.                                      */
/* ------------------------------------------------------------------------- */
static inline float ora_clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

ORA_API void ora_total_variation_add_grad(const float* param, float* grad, float wx, float wy,
                                          float wz, int64_t sz_i, int64_t sz_j, int64_t sz_k,
                                          int64_t N, int dense_mode) {
  wx /= 6; wy /= 6; wz /= 6;
  (void)wx;
  for (int64_t index = 0; index < N; ++index) {
    if (!(dense_mode || grad[index] != 0)) continue;
    const int64_t k = index % sz_k;
    const int64_t j = index / sz_k % sz_j;
    const int64_t i = index / sz_k / sz_j % sz_i;
    float g = 0;
    g += (k == 0        ? 0 : wz * ora_clampf(param[index] - param[index - 1], -1.f, 1.f));
    g += (k == sz_k - 1 ? 0 : wz * ora_clampf(param[index] - param[index + 1], -1.f, 1.f));
    g += (j == 0        ? 0 : wy * ora_clampf(param[index] - param[index - sz_k], -1.f, 1.f));
    g += (j == sz_j - 1 ? 0 : wy * ora_clampf(param[index] - param[index + sz_k], -1.f, 1.f));
    g += (i == 0        ? 0 : wz * ora_clampf(param[index] - param[index - sz_k * sz_j], -1.f, 1.f));
    g += (i == sz_i - 1 ? 0 : wz * ora_clampf(param[index] - param[index + sz_k * sz_j], -1.f, 1.f));
    grad[index] += g;
  }
}
