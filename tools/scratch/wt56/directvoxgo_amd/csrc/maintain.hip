// Grid maintenance of the coarse stage ("next" row N4 of SURVEY.md section 8f), as kernels instead of the reference's
// autograd round trip:
//
//   voxel_count_views     /root/reference/lib/dvgo.py:265-295.  The reference counts, per voxel, the training views that
//                         "see" it by pushing ones through grid_sample, summing, calling backward() and testing
//                         ones.grad > 1 -- i.e. per view: acc[v] = sum of the trilinear weights voxel v receives from the
//                         view's sample points, count[v] += acc[v] > 1.  Here: one wavefront per ray, lanes = sample
//                         points, the 8 x 64 corner weights of a chunk merged in a per-wave LDS table before they go out
//                         as float atomics into the view's accumulator; a second streaming kernel commits
//                         count += (acc > 1) and clears the accumulator for the next view.
//   maskout_near_cam_vox  lib/dvgo.py:215-226: density = -100 where the nearest camera is within `near`.
#include "common.h"

struct ViewParams {
  float mnx, mny, mnz, mxx, mxy, mxz;
  float near, far, step;     // step = stepsize * voxel_size (float32, as the reference's tensor product)
  int n_samples;
  int X, Y, Z;
};

__global__ void __launch_bounds__(DVGO_BLOCK)
view_weight_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, int64_t n_rays, ViewParams P,
                   float* __restrict__ acc) {
  constexpr int H = 512;
  __shared__ int s_keys[4][H];
  __shared__ float s_vals[4][H];
  const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (ray >= n_rays) return;
  int* keys = s_keys[threadIdx.x >> 6];
  float* vals = s_vals[threadIdx.x >> 6];
  for (int s = lane; s < H; s += 64) { keys[s] = -1; vals[s] = 0.0f; }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  const float ox = rays_o[3 * ray], oy = rays_o[3 * ray + 1], oz = rays_o[3 * ray + 2];
  const float dx = rays_d[3 * ray], dy = rays_d[3 * ray + 1], dz = rays_d[3 * ray + 2];
  // lib/dvgo.py:281-285: slab entry with the 1e-6 substitution, clamp(min=near, max=far)
  const float vx = (dx == 0.f) ? 1e-6f : dx, vy = (dy == 0.f) ? 1e-6f : dy, vz = (dz == 0.f) ? 1e-6f : dz;
  const float ax = (P.mxx - ox) / vx, bx = (P.mnx - ox) / vx;
  const float ay = (P.mxy - oy) / vy, by = (P.mny - oy) / vy;
  const float az = (P.mxz - oz) / vz, bz = (P.mnz - oz) / vz;
  float t_min = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
  t_min = fminf(fmaxf(t_min, P.near), P.far);
  const float norm = sqrtf(dx * dx + dy * dy + dz * dz);
  const int64_t YZ = (int64_t)P.Y * P.Z;
  for (int base = 0; base < P.n_samples; base += 64) {
    const int k = base + lane;
    if (k < P.n_samples) {
      const float t = t_min + (P.step * (float)k) / norm;            // interpx (:287)
      const float px = ox + dx * t, py = oy + dy * t, pz = oz + dz * t;
      const TriSetup tr = dvgo_tri_setup(px, py, pz, P.mnx, P.mny, P.mnz, P.mxx, P.mxy, P.mxz, P.X, P.Y, P.Z);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (!dvgo_tri_inb(tr, c, P.X, P.Y, P.Z)) continue;
        const float w = dvgo_tri_weight(tr, c);
        const int key = (int)((int64_t)(tr.i0 + ((c >> 2) & 1)) * YZ + (int64_t)(tr.j0 + ((c >> 1) & 1)) * P.Z + (tr.k0 + (c & 1)));
        int sidx = (int)((unsigned)key * 2654435761u >> 23);     // H = 512
        bool placed = false;
        for (int probes = 0; probes < 16; ++probes) {
          const int prev = atomicCAS(&keys[sidx], -1, key);
          if (prev == -1 || prev == key) { placed = true; break; }
          sidx = (sidx + 1) & (H - 1);
        }
        if (placed) atomicAdd(&vals[sidx], w);
        else atomicAdd(acc + key, w);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#pragma unroll
    for (int it = 0; it < H / 64; ++it) {
      const int s = it * 64 + lane;
      const int kk = keys[s];
      if (kk != -1) {
        atomicAdd(acc + kk, vals[s]);
        keys[s] = -1;
        vals[s] = 0.0f;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  }
}

__global__ void __launch_bounds__(DVGO_BLOCK)
view_commit_kernel(float* __restrict__ acc, float* __restrict__ count, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = acc[i];
  if (a > 1.0f) count[i] += 1.0f;          // count += (ones.grad > 1)   (:291)
  if (a != 0.0f) acc[i] = 0.0f;
}

// one thread per voxel; gx/gy/gz are the torch.linspace coordinate vectors of the three axes (so that the voxel centres
// carry the reference's own rounding), cams [n_cam,3]
__global__ void __launch_bounds__(DVGO_BLOCK)
maskout_near_cam_kernel(float* __restrict__ density, const float* __restrict__ gx, const float* __restrict__ gy,
                        const float* __restrict__ gz, int X, int Y, int Z, const float* __restrict__ cams, int n_cam,
                        float near, float value) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= (int64_t)X * Y * Z) return;
  const int k = (int)(v % Z), j = (int)((v / Z) % Y), i = (int)(v / ((int64_t)Y * Z));
  const float x = gx[i], y = gy[j], z = gz[k];
  float best = INFINITY;
  for (int c = 0; c < n_cam; ++c) {
    const float ex = x - cams[3 * c], ey = y - cams[3 * c + 1], ez = z - cams[3 * c + 2];
    best = fminf(best, sqrtf((ex * ex + ey * ey) + ez * ez));
  }
  if (best <= near) density[v] = value;
}

extern "C" {

int dvgo_view_weight_accumulate(const float* rays_o, const float* rays_d, int64_t n_rays, const float* xyz_min,
                                const float* xyz_max, float near, float far, float step, int n_samples, int X, int Y,
                                int Z, float* acc, void* stream) {
  if (n_rays < 0 || n_samples < 0 || X <= 0 || Y <= 0 || Z <= 0) return DVGO_EINVAL;
  if (n_rays == 0 || n_samples == 0) return 0;
  if (!rays_o || !rays_d || !xyz_min || !xyz_max || !acc) return DVGO_EINVAL;
  if (!dvgo_fits(n_rays * 64) || (int64_t)X * Y * Z >= ((int64_t)1 << 31)) return DVGO_ERANGE;
  ViewParams P;
  P.mnx = xyz_min[0]; P.mny = xyz_min[1]; P.mnz = xyz_min[2];
  P.mxx = xyz_max[0]; P.mxy = xyz_max[1]; P.mxz = xyz_max[2];
  P.near = near; P.far = far; P.step = step; P.n_samples = n_samples;
  P.X = X; P.Y = Y; P.Z = Z;
  view_weight_kernel<<<dvgo_blocks(n_rays * 64, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(rays_o, rays_d, n_rays, P, acc);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_view_count_commit(float* acc, float* count, int64_t n_vox, void* stream) {
  if (n_vox < 0) return DVGO_EINVAL;
  if (n_vox == 0) return 0;
  if (!acc || !count) return DVGO_EINVAL;
  if (!dvgo_fits(n_vox)) return DVGO_ERANGE;
  view_commit_kernel<<<dvgo_blocks(n_vox, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(acc, count, n_vox);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_maskout_near_cam(float* density, const float* grid_x, const float* grid_y, const float* grid_z, int X, int Y, int Z,
                          const float* cam_o, int n_cam, float near, float value, void* stream) {
  if (X <= 0 || Y <= 0 || Z <= 0 || n_cam < 0) return DVGO_EINVAL;
  if (n_cam == 0) return 0;
  if (!density || !grid_x || !grid_y || !grid_z || !cam_o) return DVGO_EINVAL;
  const int64_t n = (int64_t)X * Y * Z;
  if (!dvgo_fits(n)) return DVGO_ERANGE;
  maskout_near_cam_kernel<<<dvgo_blocks(n, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(density, grid_x, grid_y, grid_z, X, Y, Z,
                                                                                            cam_o, n_cam, near, value);
  DVGO_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
