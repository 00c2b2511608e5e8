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

// --------------------------------------------------------------------------------------
// Single-channel trilinear value with every corner load in flight at once.  The straightforward form -- `if (in
// bounds) d = fmaf(grid[off], w, d)` per corner -- compiles to eight dependent round trips (load, s_waitcnt
// vmcnt(0), fma, next corner); here the loads are unconditional (out-of-range corners read a clamped, valid
// address and are dropped by a select), the two z-neighbours of a corner pair arrive as ONE 8-byte load (Z is the
// innermost axis: they are adjacent dwords; 4-byte aligned, which global loads allow), and the fma chain runs
// afterwards in the reference's corner order, skipping out-of-range corners exactly as before: same bits.
// Needs Z >= 2.
// --------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) dvgo_f2u { float x, y; };

__device__ __forceinline__ float dvgo_tri_value_c1(const float* __restrict__ grid, const TriSetup& t, int X, int Y, int Z) {
  const int i1 = t.i0 + 1, j1 = t.j0 + 1, k1 = t.k0 + 1;
  const bool x0 = (t.i0 >= 0) & (t.i0 < X), x1 = (i1 >= 0) & (i1 < X);
  const bool y0 = (t.j0 >= 0) & (t.j0 < Y), y1 = (j1 >= 0) & (j1 < Y);
  const bool z0 = (t.k0 >= 0) & (t.k0 < Z), z1 = (k1 >= 0) & (k1 < Z);
  const int ia = min(max(t.i0, 0), X - 1), ib = min(max(i1, 0), X - 1);
  const int ja = min(max(t.j0, 0), Y - 1), jb = min(max(j1, 0), Y - 1);
  const int kb = min(max(t.k0, 0), Z - 2);
  const dvgo_f2u q00 = *reinterpret_cast<const dvgo_f2u*>(grid + ((int64_t)(ia * Y + ja) * Z + kb));
  const dvgo_f2u q01 = *reinterpret_cast<const dvgo_f2u*>(grid + ((int64_t)(ia * Y + jb) * Z + kb));
  const dvgo_f2u q10 = *reinterpret_cast<const dvgo_f2u*>(grid + ((int64_t)(ib * Y + ja) * Z + kb));
  const dvgo_f2u q11 = *reinterpret_cast<const dvgo_f2u*>(grid + ((int64_t)(ib * Y + jb) * Z + kb));
  // kb == k0 except at the faces: k0 == -1 -> kb = 0 (voxel k1 = 0 is .x), k0 == Z-1 -> kb = Z-2 (voxel k0 is .y)
  const bool lo = t.k0 < kb, hi = t.k0 > kb;
#define DVGO_ZSEL(q, v0, v1) const float v0 = hi ? q.y : q.x, v1 = lo ? q.x : q.y
  DVGO_ZSEL(q00, v000, v001); DVGO_ZSEL(q01, v010, v011); DVGO_ZSEL(q10, v100, v101); DVGO_ZSEL(q11, v110, v111);
#undef DVGO_ZSEL
  float d = 0.f;
  d = (x0 & y0 & z0) ? fmaf(v000, (t.wz0 * t.wy0) * t.wx0, d) : d;
  d = (x0 & y0 & z1) ? fmaf(v001, (t.wz1 * t.wy0) * t.wx0, d) : d;
  d = (x0 & y1 & z0) ? fmaf(v010, (t.wz0 * t.wy1) * t.wx0, d) : d;
  d = (x0 & y1 & z1) ? fmaf(v011, (t.wz1 * t.wy1) * t.wx0, d) : d;
  d = (x1 & y0 & z0) ? fmaf(v100, (t.wz0 * t.wy0) * t.wx1, d) : d;
  d = (x1 & y0 & z1) ? fmaf(v101, (t.wz1 * t.wy0) * t.wx1, d) : d;
  d = (x1 & y1 & z0) ? fmaf(v110, (t.wz0 * t.wy1) * t.wx1, d) : d;
  d = (x1 & y1 & z1) ? fmaf(v111, (t.wz1 * t.wy1) * t.wx1, d) : d;
  return d;
}

// x^y for x >= 1 (or +inf) and y < 0: the power of K9 / K10 (render_utils_kernel.cu:367,404: `pow(1 + exp_d, -interval)`),
// as exp2(y * log2 x) on the hardware's v_log_f32 / v_exp_f32.  The C library's powf is ~200 VALU instructions of
// special-case handling (negative bases, integer exponents, denormals) that this argument range never needs -- a
// seventh of the instructions of the VALU-bound march_density kernel.  x = 1 -> 1, x = +inf -> 0, as pow.  Every kernel
// that evaluates the activation calls this one function, so the fused and the op-by-op paths agree bit for bit; against
// the CPU oracle (glibc powf) alpha stays inside the activation's stated tolerance (rtol 1e-5 / atol 1e-6: the result
// p is within a few ulp, and alpha = 1 - p inherits p's ABSOLUTE error, <= ~2e-7).
__device__ __forceinline__ float dvgo_pow_neg(float x, float y) {
  return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
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

// K1 + K2 + K3 of one ray (render_utils_kernel.cu:11-73), expression for expression what sampling.hip's ray_setup_kernel
// computes (same flags, hence the same bits): slab test -> t_min / t_max, step count, start point and unit direction.
struct RaySetup { float tmin, tmax, sx, sy, sz, dx, dy, dz; int64_t n; };
__device__ __forceinline__ RaySetup dvgo_ray_setup(float ox, float oy, float oz, float dx, float dy, float dz,
                                                   float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                                                   float near, float far, float stepdist) {
  RaySetup R;
  // K1 :23-33
  const float vx = (dx == 0) ? (float)1e-6 : dx;
  const float vy = (dy == 0) ? (float)1e-6 : dy;
  const float vz = (dz == 0) ? (float)1e-6 : dz;
  const float ax = (mxx - ox) / vx, ay = (mxy - oy) / vy, az = (mxz - oz) / vz;
  const float bx = (mnx - ox) / vx, by = (mny - oy) / vy, bz = (mnz - oz) / vz;
  R.tmin = fmaxf(fminf(fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz)), far), near);
  R.tmax = fmaxf(fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)), far), near);
  // K2 :47  max(ceil((t_max - t_min) / stepdist), 1.) -> int64
  const float c = ceilf((R.tmax - R.tmin) / stepdist);
  R.n = (int64_t)fmax((double)c, 1.);
  // K3 :62-71
  const float rnorm = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
  R.sx = fmaf(dx, R.tmin, ox); R.sy = fmaf(dy, R.tmin, oy); R.sz = fmaf(dz, R.tmin, oz);
  R.dx = dx / rnorm; R.dy = dy / rnorm; R.dz = dz / rnorm;
  return R;
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
// v with lane `lane` (wave-uniform) replaced by the wave-uniform value s: one v_writelane_b32 (this hipcc has no
// __builtin_amdgcn_writelane; the declaration below binds the LLVM intrinsic by its name)
extern "C" __device__ int dvgo_llvm_writelane_i32(int, int, int) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ float dvgo_writelane_f(float s, int lane, float v) {
  return __int_as_float(dvgo_llvm_writelane_i32(__float_as_int(s), lane, __float_as_int(v)));
}
// v_readlane_b32 the optimiser cannot see through: in the transmittance walk it would otherwise read the two halves of
// the double product and narrow on the scalar side (two more vector instructions per step)
__device__ __forceinline__ float dvgo_readlane_opaque_f(float v, int lane) {
  int o;
  asm("v_readlane_b32 %0, %1, %2" : "=s"(o) : "v"(v), "s"(lane));
  return __int_as_float(o);
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
// (selects only: with branches and by-reference outputs hipcc turned the six brick coordinates of a sample into a
// dynamically indexed private array -- 28 B of scratch per lane in both march kernels)
__device__ __forceinline__ int dvgo_brick_axis(int i0, int n, int& b0, int& b1) {
  const int i1 = i0 + 1;
  const bool in0 = (i0 >= 0) & (i0 < n), in1 = (i1 >= 0) & (i1 < n);
  const int v0 = i0 >> DVGO_BRICK_LOG, v1 = i1 >> DVGO_BRICK_LOG;
  const bool two = in0 & in1 & (v1 != v0);
  b0 = in0 ? v0 : (in1 ? v1 : 0);
  b1 = two ? v1 : 0;
  return two ? 2 : ((in0 | in1) ? 1 : 0);
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
