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

// ----------------------------------------------------------------------------------
// rays_of_view: rays_o / rays_d / viewdirs of the pixels [p0, p0 + n) of one view, row-major (lib/ray_utils.py:9-85:
// get_rays + the viewdirs of get_rays_of_a_view + ndc_rays), one thread per pixel -- instead of ~10 framework launches and
// 23 MB of intermediate [H, W, 3] tensors per 800 x 800 view in front of a 12 ms render.  The arithmetic is the
// reference's, operation for operation in fp32 without contraction: pixel coordinate (+ 0.5 for mode 'center'), flips,
// dirs = ((i - cx) / fx, -(j - cy) / fy, -1) (inverse_y: (.., (j - cy) / fy, 1)), rays_d[k] = sum_m dirs[m] * c2w[k][m]
// (three products, summed left to right), rays_o = c2w[:3, 3], viewdirs = rays_d / |rays_d|.
// ----------------------------------------------------------------------------------
struct ViewCam {
  float R[3][3], t[3];
  float fx, fy, cx, cy;
  int H, W;
  int inverse_y, flip_x, flip_y, center, ndc;
  float ndc_focal, ndc_near;
};

__global__ void __launch_bounds__(DVGO_BLOCK)
rays_of_view_kernel(ViewCam V, int64_t p0, int64_t n, float* __restrict__ rays_o, float* __restrict__ rays_d,
                    float* __restrict__ viewdirs) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int64_t p = p0 + q;
  int row = (int)(p / V.W), col = (int)(p - (int64_t)row * V.W);
  if (V.flip_x) col = V.W - 1 - col;
  if (V.flip_y) row = V.H - 1 - row;
  const float i = (float)col + (V.center ? 0.5f : 0.0f), j = (float)row + (V.center ? 0.5f : 0.0f);
  float d0 = (i - V.cx) / V.fx, d1, d2;
  if (V.inverse_y) { d1 = (j - V.cy) / V.fy; d2 = 1.0f; }
  else { d1 = -(j - V.cy) / V.fy; d2 = -1.0f; }
  float rd[3], ro[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    rd[k] = (d0 * V.R[k][0] + d1 * V.R[k][1]) + d2 * V.R[k][2];
    ro[k] = V.t[k];
  }
  const float nrm = sqrtf((rd[0] * rd[0] + rd[1] * rd[1]) + rd[2] * rd[2]);
  if (viewdirs != nullptr) {
#pragma unroll
    for (int k = 0; k < 3; ++k) viewdirs[3 * q + k] = rd[k] / nrm;
  }
  if (V.ndc) {                                  // lib/ray_utils.py:60-77
    const float near = V.ndc_near, f = V.ndc_focal;
    const float t = -(near + ro[2]) / rd[2];
    const float ox = ro[0] + t * rd[0], oy = ro[1] + t * rd[1], oz = ro[2] + t * rd[2];
    const float sw = -1.0f / ((float)V.W / (2.0f * f)), sh = -1.0f / ((float)V.H / (2.0f * f));
    const float o0 = sw * ox / oz, o1 = sh * oy / oz, o2 = 1.0f + 2.0f * near / oz;
    const float e0 = sw * (rd[0] / rd[2] - ox / oz), e1 = sh * (rd[1] / rd[2] - oy / oz), e2 = -2.0f * near / oz;
    ro[0] = o0; ro[1] = o1; ro[2] = o2; rd[0] = e0; rd[1] = e1; rd[2] = e2;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) { rays_o[3 * q + k] = ro[k]; rays_d[3 * q + k] = rd[k]; }
}

extern "C" {

int dvgo_rays_of_view(int H, int W, const float* K4 /* host: fx, fy, cx, cy */, const float* c2w /* host: 3 x 4 row-major */,
                      int inverse_y, int flip_x, int flip_y, int center, int ndc, float ndc_focal, float ndc_near, int64_t p0,
                      int64_t n, float* rays_o, float* rays_d, float* viewdirs, void* stream) {
  if (H <= 0 || W <= 0 || p0 < 0 || n < 0 || p0 + n > (int64_t)H * W) return DVGO_EINVAL;
  if (n == 0) return 0;
  if (!K4 || !c2w || !rays_o || !rays_d) return DVGO_EINVAL;
  if (!dvgo_fits(n)) return DVGO_ERANGE;
  ViewCam V;
  for (int k = 0; k < 3; ++k) {
    for (int m = 0; m < 3; ++m) V.R[k][m] = c2w[4 * k + m];
    V.t[k] = c2w[4 * k + 3];
  }
  V.fx = K4[0]; V.fy = K4[1]; V.cx = K4[2]; V.cy = K4[3];
  V.H = H; V.W = W; V.inverse_y = inverse_y; V.flip_x = flip_x; V.flip_y = flip_y; V.center = center; V.ndc = ndc;
  V.ndc_focal = ndc_focal; V.ndc_near = ndc_near;
  rays_of_view_kernel<<<dvgo_blocks(n, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(V, p0, n, rays_o, rays_d, viewdirs);
  DVGO_LAUNCH_CHECK();
  return 0;
}


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
