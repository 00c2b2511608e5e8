// Shared device/host helpers for libdvgo_hip.so (gfx950 only; wavefront = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dvgo_hip.h"

#define DVGO_WAVE 64
#define DVGO_BLOCK 256

#define DVGO_LAUNCH_CHECK()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return (int)e__;                   \
  } while (0)

#define DVGO_HIP_TRY(expr)                                    \
  do {                                                        \
    hipError_t e__ = (expr);                                  \
    if (e__ != hipSuccess) return (int)e__;                   \
  } while (0)

static inline int dvgo_blocks(int64_t work, int per_block) {
  return (int)((work + per_block - 1) / per_block);
}

// 1-D launches are limited to 2^31-1 blocks; all flat kernels index with int64 but the
// sizes the path sees (M0 < 2^31, as in the reference: render_utils_kernel.cu:206) fit.
static inline bool dvgo_fits(int64_t work) { return work >= 0 && work < ((int64_t)1 << 31); }

// --------------------------------------------------------------------------------------
// Trilinear setup shared by every interpolation kernel.  Restates
//   lib/dvgo.py:316      ind_norm = ((xyz - min) / (max - min)).flip(-1) * 2 - 1
//   ATen GridSampler.h   unnormalise (align_corners=True): ((c + 1) / 2) * (size - 1)
//   ATen grid_sampler_3d corner weights as differences to the opposite corner.
// Kept operation for operation identical to oracle/dvgo_oracle.c (ora_corners) so that
// floor() decisions and weights are bit-identical; compiled with -ffp-contract=off.
// --------------------------------------------------------------------------------------
struct TriSetup {
  int i0, j0, k0;        // floor corner (may be -1 or size-1 at the faces)
  float wx0, wx1, wy0, wy1, wz0, wz1;
  float gx, gy, gz;      // continuous voxel coordinates (what the weights are derived from)
};

__device__ __forceinline__ float dvgo_src_index(float p, float mn, float mx, int size) {
  const float u = (p - mn) / (mx - mn);
  const float c = u * 2.0f - 1.0f;
  return ((c + 1.0f) / 2.0f) * (float)(size - 1);
}

// floor corner and weights from the continuous voxel coordinates (also used where a kernel hands g over to another
// one instead of the position: same expressions, hence the same bits)
__device__ __forceinline__ TriSetup dvgo_tri_from_g(float gx, float gy, float gz) {
  TriSetup t;
  const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
  t.i0 = (int)fx; t.j0 = (int)fy; t.k0 = (int)fz;
  t.wx0 = (fx + 1.0f) - gx; t.wx1 = gx - fx;
  t.wy0 = (fy + 1.0f) - gy; t.wy1 = gy - fy;
  t.wz0 = (fz + 1.0f) - gz; t.wz1 = gz - fz;
  t.gx = gx; t.gy = gy; t.gz = gz;
  return t;
}

__device__ __forceinline__ TriSetup dvgo_tri_setup(float px, float py, float pz,
                                                   float mnx, float mny, float mnz,
                                                   float mxx, float mxy, float mxz,
                                                   int X, int Y, int Z) {
  const float gx = dvgo_src_index(px, mnx, mxx, X);
  const float gy = dvgo_src_index(py, mny, mxy, Y);
  const float gz = dvgo_src_index(pz, mnz, mxz, Z);
  return dvgo_tri_from_g(gx, gy, gz);
}

// weight of corner n (bit2 = +X, bit1 = +Y, bit0 = +Z): (wz * wy) * wx, left to right
__device__ __forceinline__ float dvgo_tri_weight(const TriSetup& t, int n) {
  const float wz = (n & 1) ? t.wz1 : t.wz0;
  const float wy = (n & 2) ? t.wy1 : t.wy0;
  const float wx = (n & 4) ? t.wx1 : t.wx0;
  return (wz * wy) * wx;
}

__device__ __forceinline__ bool dvgo_tri_inb(const TriSetup& t, int n, int X, int Y, int Z) {
  const int i = t.i0 + ((n >> 2) & 1), j = t.j0 + ((n >> 1) & 1), k = t.k0 + (n & 1);
  return (i >= 0) & (i < X) & (j >= 0) & (j < Y) & (k >= 0) & (k < Z);
}

// Sample position on a ray: K6 (render_utils_kernel.cu:178-181): dist = stepdist * step,
// p = start + dir * dist (contracted)
__device__ __forceinline__ void dvgo_sample_pos(const float* __restrict__ start,
                                                const float* __restrict__ dir, int64_t r,
                                                float stepdist, int step,
                                                float& px, float& py, float& pz) {
  const float dist = stepdist * (float)step;
  px = fmaf(dir[3 * r + 0], dist, start[3 * r + 0]);
  py = fmaf(dir[3 * r + 1], dist, start[3 * r + 1]);
  pz = fmaf(dir[3 * r + 2], dist, start[3 * r + 2]);
}

// distance of sample `step` along its ray in the fused march:
//   K6 (render_utils_kernel.cu:178)  stepdist * i_step                -- rays_start / unit rays_dir   (stepdist > 0)
//   K7 (render_utils_kernel.cu:254)  (float)i_step / (N_samples - 1)  -- rays_o / un-normalised rays_d (stepdist < 0)
__device__ __forceinline__ float march_dist(float stepdist, int step) {
  return (stepdist > 0.0f) ? stepdist * (float)step : ((float)step) / (-stepdist);
}

__device__ __forceinline__ void march_pos(const float* __restrict__ start, const float* __restrict__ dir, int64_t r,
                                          float stepdist, int step, float& px, float& py, float& pz) {
  const float dist = march_dist(stepdist, step);
  px = fmaf(dir[3 * r + 0], dist, start[3 * r + 0]);
  py = fmaf(dir[3 * r + 1], dist, start[3 * r + 1]);
  pz = fmaf(dir[3 * r + 2], dist, start[3 * r + 2]);
}

// first index r in [0,n) with cum[r] > idx  (cum inclusive, non-decreasing)
__device__ __forceinline__ int64_t dvgo_upper_bound(const int64_t* __restrict__ cum, int64_t n,
                                                    int64_t idx) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (cum[mid] > idx) hi = mid; else lo = mid + 1;
  }
  return lo;
}

__device__ __forceinline__ float dvgo_readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double dvgo_readlane_d(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// --------------------------------------------------------------------------------------
// Bricks: the voxel lattice cut into 8x8x8 blocks, the ownership unit of the gradient scatter (brick.hip).
// A sample contributes to the (up to 2 per axis) bricks that hold its 8 corner voxels.
// --------------------------------------------------------------------------------------
#define DVGO_BRICK_LOG 3
#define DVGO_BRICK (1 << DVGO_BRICK_LOG)

// distinct brick coordinates of the in-range voxels {i0, i0+1} on one axis; returns how many (0..2)
__device__ __forceinline__ int dvgo_brick_axis(int i0, int n, int& b0, int& b1) {
  int c = 0;
  b0 = b1 = 0;
  if (i0 >= 0 && i0 < n) { b0 = i0 >> DVGO_BRICK_LOG; c = 1; }
  const int i1 = i0 + 1;
  if (i1 >= 0 && i1 < n) {
    const int v = i1 >> DVGO_BRICK_LOG;
    if (c == 0) { b0 = v; c = 1; }
    else if (v != b0) { b1 = v; c = 2; }
  }
  return c;
}

__device__ __forceinline__ unsigned long long dvgo_lanemask_le(int lane) { return ~0ull >> (63 - lane); }

// One Adam element update (adam_upd_kernel.cu:8-58): MODE 0 plain, 1 masked (skip grad == 0), 2 per-voxel lr.
template <int MODE>
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float perlr,
                                         float step_size, float beta1, float beta2, float eps) {
  if (MODE == 1 && g == 0.0f) return;
  m = fmaf(beta1, m, (1.0f - beta1) * g);
  v = fmaf(beta2, v, ((1.0f - beta2) * g) * g);
  const float ss = (MODE == 2) ? step_size * perlr : step_size;
  p = p - (ss * m) / (sqrtf(v) + eps);
}
