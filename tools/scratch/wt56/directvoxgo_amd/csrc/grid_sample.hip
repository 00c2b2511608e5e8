// Trilinear interpolation of a dense voxel grid and its gradient scatter ("DenseGrid").
// Replaces lib/dvgo.py:312-328 grid_sampler -> torch F.grid_sample(mode='bilinear',
// align_corners=True, padding zeros) and grid_sampler_3d_backward w.r.t. the grid.
//
// The grid is addressed through element strides.  Two layouts matter:
//   channel-first  [C,X,Y,Z]  (sZ == 1) -- the reference's parameter layout: one sample touches
//                  8*C scattered dwords from C planes that are X*Y*Z*4 bytes apart;
//   channels-last  [X,Y,Z,C]  (sC == 1) -- this library's preferred layout: the C values of a
//                  corner are contiguous (48 B for C = 12), fetched as 16-byte vectors, and the
//                  two z-neighbours of a corner pair are adjacent in memory.
#include "common.h"

template <int VEC>   // VEC = 4: channels-last with C % 4 == 0 and 16-B aligned base; 1: generic
__global__ void __launch_bounds__(DVGO_BLOCK)
grid_sample_fwd_kernel(const float* __restrict__ grid, int C, int X, int Y, int Z,
                       int64_t sC, int64_t sX, int64_t sY, int64_t sZ,
                       const float* __restrict__ xyz, const float* __restrict__ xyz_min,
                       const float* __restrict__ xyz_max, int64_t M, float* __restrict__ out) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const TriSetup t = dvgo_tri_setup(xyz[3 * m], xyz[3 * m + 1], xyz[3 * m + 2], xyz_min[0], xyz_min[1],
                                    xyz_min[2], xyz_max[0], xyz_max[1], xyz_max[2], X, Y, Z);
  float w[8];
  int64_t off[8];
  bool ok[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    w[n] = dvgo_tri_weight(t, n);
    ok[n] = dvgo_tri_inb(t, n, X, Y, Z);
    off[n] = (int64_t)(t.i0 + ((n >> 2) & 1)) * sX + (int64_t)(t.j0 + ((n >> 1) & 1)) * sY +
             (int64_t)(t.k0 + (n & 1)) * sZ;
  }
  if (VEC == 4) {
    for (int c = 0; c < C; c += 4) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        if (ok[n]) {
          const float4 v = *reinterpret_cast<const float4*>(grid + off[n] + c);
          acc.x = fmaf(v.x, w[n], acc.x);
          acc.y = fmaf(v.y, w[n], acc.y);
          acc.z = fmaf(v.z, w[n], acc.z);
          acc.w = fmaf(v.w, w[n], acc.w);
        }
      }
      *reinterpret_cast<float4*>(out + m * C + c) = acc;
    }
  } else {
    for (int c = 0; c < C; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int n = 0; n < 8; ++n)
        if (ok[n]) acc = fmaf(grid[c * sC + off[n]], w[n], acc);
      out[m * C + c] = acc;
    }
  }
}

// Gradient scatter: grad_grid[corner, c] += w * grad_out[m, c]  (float atomics).
// Channels-last maps (sample, channel) onto adjacent lanes so that one atomic wave-instruction
// covers contiguous 4*C-byte runs; channel-first falls back to one lane per sample.
__global__ void __launch_bounds__(DVGO_BLOCK)
grid_sample_bwd_kernel(const float* __restrict__ grad_out, int C, int X, int Y, int Z,
                       int64_t sC, int64_t sX, int64_t sY, int64_t sZ,
                       const float* __restrict__ xyz, const float* __restrict__ xyz_min,
                       const float* __restrict__ xyz_max, int64_t M, float* __restrict__ grad_grid) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const TriSetup t = dvgo_tri_setup(xyz[3 * m], xyz[3 * m + 1], xyz[3 * m + 2], xyz_min[0], xyz_min[1],
                                    xyz_min[2], xyz_max[0], xyz_max[1], xyz_max[2], X, Y, Z);
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    if (!dvgo_tri_inb(t, n, X, Y, Z)) continue;
    const float w = dvgo_tri_weight(t, n);
    const int64_t off = (int64_t)(t.i0 + ((n >> 2) & 1)) * sX + (int64_t)(t.j0 + ((n >> 1) & 1)) * sY +
                        (int64_t)(t.k0 + (n & 1)) * sZ;
    for (int c = 0; c < C; ++c) atomicAdd(grad_grid + c * sC + off, w * grad_out[m * C + c]);
  }
}

// channels-last: thread = (sample, channel)
__global__ void __launch_bounds__(DVGO_BLOCK)
grid_sample_bwd_cl_kernel(const float* __restrict__ grad_out, int C, int X, int Y, int Z,
                          int64_t sX, int64_t sY, int64_t sZ,
                          const float* __restrict__ xyz, const float* __restrict__ xyz_min,
                          const float* __restrict__ xyz_max, int64_t M, float* __restrict__ grad_grid) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= M * C) return;
  const int64_t m = tid / C;
  const int c = (int)(tid - m * C);
  const TriSetup t = dvgo_tri_setup(xyz[3 * m], xyz[3 * m + 1], xyz[3 * m + 2], xyz_min[0], xyz_min[1],
                                    xyz_min[2], xyz_max[0], xyz_max[1], xyz_max[2], X, Y, Z);
  const float g = grad_out[tid];
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    if (!dvgo_tri_inb(t, n, X, Y, Z)) continue;
    const float w = dvgo_tri_weight(t, n);
    const int64_t off = (int64_t)(t.i0 + ((n >> 2) & 1)) * sX + (int64_t)(t.j0 + ((n >> 1) & 1)) * sY +
                        (int64_t)(t.k0 + (n & 1)) * sZ;
    atomicAdd(grad_grid + off + c, w * g);
  }
}

static int check_grid_args(const void* grid, int C, int X, int Y, int Z, const void* xyz,
                           const void* mn, const void* mx, int64_t M, const void* out) {
  if (M < 0 || C < 0 || X <= 0 || Y <= 0 || Z <= 0) return DVGO_EINVAL;
  if (M == 0 || C == 0) return 1;   // nothing to do
  if (!grid || !xyz || !mn || !mx || !out) return DVGO_EINVAL;
  if (!dvgo_fits(M * (int64_t)(C > 0 ? C : 1))) return DVGO_ERANGE;
  return 0;
}

extern "C" {

int dvgo_grid_sample_fwd(const float* grid, int C, int X, int Y, int Z, int64_t sC, int64_t sX,
                         int64_t sY, int64_t sZ, const float* xyz, const float* xyz_min,
                         const float* xyz_max, int64_t M, float* out, void* stream) {
  const int rc = check_grid_args(grid, C, X, Y, Z, xyz, xyz_min, xyz_max, M, out);
  if (rc < 0) return rc;
  if (rc == 1) return 0;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (sC == 1) && (C % 4 == 0) && (sX % 4 == 0) && (sY % 4 == 0) && (sZ % 4 == 0) &&
                   ((((uintptr_t)grid) & 15) == 0) && ((((uintptr_t)out) & 15) == 0);
  if (vec)
    grid_sample_fwd_kernel<4><<<dvgo_blocks(M, DVGO_BLOCK), DVGO_BLOCK, 0, s>>>(
        grid, C, X, Y, Z, sC, sX, sY, sZ, xyz, xyz_min, xyz_max, M, out);
  else
    grid_sample_fwd_kernel<1><<<dvgo_blocks(M, DVGO_BLOCK), DVGO_BLOCK, 0, s>>>(
        grid, C, X, Y, Z, sC, sX, sY, sZ, xyz, xyz_min, xyz_max, M, out);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_grid_sample_bwd(const float* grad_out, int C, int X, int Y, int Z, int64_t sC, int64_t sX,
                         int64_t sY, int64_t sZ, const float* xyz, const float* xyz_min,
                         const float* xyz_max, int64_t M, float* grad_grid, void* stream) {
  const int rc = check_grid_args(grad_out, C, X, Y, Z, xyz, xyz_min, xyz_max, M, grad_grid);
  if (rc < 0) return rc;
  if (rc == 1) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (sC == 1 && C > 1)
    grid_sample_bwd_cl_kernel<<<dvgo_blocks(M * C, DVGO_BLOCK), DVGO_BLOCK, 0, s>>>(
        grad_out, C, X, Y, Z, sX, sY, sZ, xyz, xyz_min, xyz_max, M, grad_grid);
  else
    grid_sample_bwd_kernel<<<dvgo_blocks(M, DVGO_BLOCK), DVGO_BLOCK, 0, s>>>(
        grad_out, C, X, Y, Z, sC, sX, sY, sZ, xyz, xyz_min, xyz_max, M, grad_grid);
  DVGO_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
