/*
 * dvgo_hip.h -- C ABI of libdvgo_hip.so, the MI355X (gfx950) implementation of
 * DirectVoxGO's volumetric ray-marching hot path.
 *
 * This is the drop-in boundary.  The reference binds this path through a pybind11
 * module (`render_utils_cuda`, /root/reference/lib/cuda/render_utils.cpp:144-155) plus
 * two third-party calls (torch F.grid_sample, torch_scatter.segment_coo).  Every entry
 * point below replaces one of those callables; the reference file:line is cited on each.
 * The ctypes binding a maintainer would add is shown in INTEGRATION.md and implemented in
 * directvoxgo_amd/_lib.py + directvoxgo_amd/render_utils.py.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it is
 *     documented as host; no torch / ATen types.
 *   - fp32 tensors (`float`), int64 indices (`int64_t`), bool tensors as 1 byte/element
 *     (`uint8_t`, 0/1) -- the dtypes the reference uses in practice.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *     synchronises unless stated.  (The reference uses the legacy default stream and
 *     the current device; callers here pass torch's current stream.)
 *   - return value: 0 on success, otherwise the hipError_t of the failed launch /
 *     a negative DVGO_E* code for invalid arguments.  Empty inputs (0 rays / 0 points)
 *     are valid everywhere and return 0 without launching.
 *   - outputs are caller-allocated; sizes are stated per function.
 *   - `m_dev` (where an entry point has it, right after a sample count M): NULL, or a DEVICE pointer to the actual
 *     sample count.  The kernels then process min(M, *m_dev) samples and M is only the capacity of the arrays: the
 *     training step keeps the data-dependent count of surviving samples on the device and never reads it back
 *     (train.py, fused.py `capacity` mode), which removes the forward's host synchronisation.
 */
#ifndef DVGO_HIP_H
#define DVGO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVGO_EINVAL (-1)   /* invalid argument (null pointer, negative size, bad stride) */
#define DVGO_ERANGE (-2)   /* size exceeds what the kernel's 32-bit indexing supports    */

/* version of this ABI; bumped on any signature change */
int dvgo_abi_version(void);

/* Kernel-variant selection for A/B measurements (process-global; defaults are the fastest
 * measured variants).  key: one of DVGO_TUNE_*; returns DVGO_EINVAL for an unknown key. */
#define DVGO_TUNE_FEAT_BWD    0   /* 0 = one atomic per (sample,corner,channel); 1 = LDS de-duplicated rows */
#define DVGO_TUNE_DENSITY_BWD 1   /* 0 = direct atomics; 1 = LDS de-duplicated */
#define DVGO_TUNE_COUNT       8
int dvgo_set_tuning(int key, int value);

/* ---------------------------------------------------------------------------------
 * Sampling helpers.  render_utils.cpp:44-65 / render_utils_kernel.cu:11-132 (K1-K3)
 * --------------------------------------------------------------------------------- */
/* infer_t_minmax(rays_o, rays_d, xyz_min, xyz_max, near, far) -> [t_min, t_max]   [n_rays] each */
int dvgo_infer_t_minmax(const float* rays_o, const float* rays_d,
                        const float* xyz_min, const float* xyz_max,
                        float near, float far, int64_t n_rays,
                        float* t_min, float* t_max, void* stream);

/* infer_n_samples(t_min, t_max, stepdist) -> n_samples [n_rays] int64 */
int dvgo_infer_n_samples(const float* t_min, const float* t_max, float stepdist,
                         int64_t n_rays, int64_t* n_samples, void* stream);

/* infer_ray_start_dir(rays_o, rays_d, t_min) -> [rays_start, rays_dir]   [n_rays,3] each */
int dvgo_infer_ray_start_dir(const float* rays_o, const float* rays_d, const float* t_min,
                             int64_t n_rays, float* rays_start, float* rays_dir, void* stream);

/* ---------------------------------------------------------------------------------
 * sample_pts_on_rays.  render_utils.cpp:67-78 / render_utils_kernel.cu:138-236 (K4-K6)
 * The output length M0 = sum(N_steps) is data dependent, so the op is two-phase:
 *   1. dvgo_sample_pts_prepare : K1+K2+K3 and the inclusive cumsum of N_steps.
 *      The caller reads n_steps_cumsum[n_rays-1] (one D2H copy: the reference's own
 *      `N_steps.sum().item()`, :206) and allocates the M0-sized outputs.
 *   2. dvgo_sample_pts_fill    : ray_id / step_id / rays_pts / mask_outbbox.
 * --------------------------------------------------------------------------------- */
int dvgo_sample_pts_prepare(const float* rays_o, const float* rays_d,
                            const float* xyz_min, const float* xyz_max,
                            float near, float far, float stepdist, int64_t n_rays,
                            float* t_min, float* t_max,            /* [n_rays]         */
                            int64_t* n_steps,                      /* [n_rays]         */
                            int64_t* n_steps_cumsum,               /* [n_rays] inclusive */
                            float* rays_start, float* rays_dir,    /* [n_rays,3]       */
                            void* stream);

int dvgo_sample_pts_fill(const float* rays_start, const float* rays_dir,
                         const float* xyz_min, const float* xyz_max,
                         const int64_t* n_steps_cumsum, int64_t n_rays,
                         float stepdist, int64_t total_len,        /* M0               */
                         float* rays_pts,                          /* [M0,3]           */
                         uint8_t* mask_outbbox,                    /* [M0]             */
                         int64_t* ray_id, int64_t* step_id,        /* [M0]             */
                         void* stream);

/* sample_ndc_pts_on_rays.  render_utils.cpp:80-91 / render_utils_kernel.cu:238-287 (K7)
 * -> rays_pts [n_rays,N_samples,3], mask_outbbox [n_rays,N_samples] */
int dvgo_sample_ndc_pts_on_rays(const float* rays_o, const float* rays_d,
                                const float* xyz_min, const float* xyz_max,
                                int n_samples, int64_t n_rays,
                                float* rays_pts, uint8_t* mask_outbbox, void* stream);

/* ---------------------------------------------------------------------------------
 * maskcache_lookup.  render_utils.cpp:93-102 / render_utils_kernel.cu:300-351 (K8)
 * world [sz_i,sz_j,sz_k] bool, xyz [n_pts,3] -> out [n_pts] bool (fully written)
 * --------------------------------------------------------------------------------- */
int dvgo_maskcache_lookup(const uint8_t* world, const float* xyz,
                          const float* xyz2ijk_scale, const float* xyz2ijk_shift,
                          int sz_i, int sz_j, int sz_k, int64_t n_pts,
                          uint8_t* out, void* stream);

/* ---------------------------------------------------------------------------------
 * raw2alpha / raw2alpha_backward.  render_utils.cpp:104-114 / _kernel.cu:357-428 (K9,K10)
 * --------------------------------------------------------------------------------- */
int dvgo_raw2alpha(const float* density, float shift, float interval, int64_t n_pts,
                   float* exp_d, float* alpha, void* stream);
int dvgo_raw2alpha_backward(const float* exp_d, const float* grad_back, float interval,
                            int64_t n_pts, float* grad, void* stream);

/* ---------------------------------------------------------------------------------
 * alpha2weight / alpha2weight_backward.  render_utils.cpp:116-141 / _kernel.cu:430-561
 * (K11-K13).  ray_id must be segment-contiguous (as the reference requires).
 * Outputs of alpha2weight: weight,T [n_pts]; alphainv_last [n_rays]; i_start,i_end [n_rays].
 * All outputs are fully written (the pre-fills of :478-482 are done here).
 * The transmittance recurrence is evaluated in the reference's serial order and
 * precision, so results are bit-identical to the oracle for identical inputs.
 * --------------------------------------------------------------------------------- */
int dvgo_alpha2weight(const float* alpha, const int64_t* ray_id, int64_t n_pts, int64_t n_rays,
                      float* weight, float* T, float* alphainv_last,
                      int64_t* i_start, int64_t* i_end, void* stream);
int dvgo_alpha2weight_backward(const float* alpha, const float* weight, const float* T,
                               const float* alphainv_last, const int64_t* i_start,
                               const int64_t* i_end, int64_t n_rays, int64_t n_pts,
                               const float* grad_weights, const float* grad_last,
                               float* grad /* [n_pts], fully written */, void* stream);

/* ---------------------------------------------------------------------------------
 * Trilinear grid interpolation ("DenseGrid").  Replaces
 *   lib/dvgo.py:312-328 grid_sampler -> F.grid_sample(grid[1,C,X,Y,Z], ind_norm,
 *   mode='bilinear', align_corners=True)  and its autograd backward w.r.t. the grid.
 * The grid is addressed through ELEMENT strides (sC,sX,sY,sZ) so the reference layout
 * [1,C,X,Y,Z] (sZ=1) and the channels-last layout this library prefers (sC=1) are both
 * accepted.  xyz [M,3] world coordinates; out / grad_out are [M,C] row-major.
 * grad_grid must be zero-initialised (or hold a partial sum) -- values are accumulated
 * with float atomics, so the summation order is not reproducible (neither is the
 * reference's, run.py:147-149).
 * --------------------------------------------------------------------------------- */
int dvgo_grid_sample_fwd(const float* grid, int C, int X, int Y, int Z,
                         int64_t sC, int64_t sX, int64_t sY, int64_t sZ,
                         const float* xyz, const float* xyz_min, const float* xyz_max,
                         int64_t M, float* out, void* stream);
int dvgo_grid_sample_bwd(const float* grad_out, int C, int X, int Y, int Z,
                         int64_t sC, int64_t sX, int64_t sY, int64_t sZ,
                         const float* xyz, const float* xyz_min, const float* xyz_max,
                         int64_t M, float* grad_grid, void* stream);

/* ---------------------------------------------------------------------------------
 * segment sum.  Replaces torch_scatter.segment_coo(src, index, out, reduce='sum')
 * (lib/dvgo.py:554-559,571-575).  src [M,C], index [M] int64 sorted/segment-contiguous,
 * out [N,C] accumulated into (caller zero-fills, as the reference passes out=zeros).
 * --------------------------------------------------------------------------------- */
int dvgo_segment_sum(const float* src, const int64_t* index, int64_t M, int C,
                     int64_t N, float* out, void* stream);

/* ---------------------------------------------------------------------------------
 * Fused march (the MI355X-native fast path behind DirectVoxGO.forward, lib/dvgo.py:450-577).
 * See DESIGN.md "Fused pipeline" for the data flow.  N rays; the per-ray inputs n_steps /
 * rays_start / rays_dir come from dvgo_sample_pts_prepare (n_steps_cumsum may be NULL there).
 *
 * In the dvgo_march_* entry points xyz_min, xyz_max, xyz2ijk_scale and xyz2ijk_shift are
 * HOST pointers to 3 floats (model constants; they travel as kernel arguments).
 *
 * `stepdist` > 0 : metric spacing, sample s of ray r sits at rays_start[r] + rays_dir[r]*(stepdist*s)
 *                  (K6, render_utils_kernel.cu:178-181; rays_start/rays_dir from dvgo_sample_pts_prepare).
 * `stepdist` < 0 : NDC / MPI spacing with N_samples-1 = -stepdist: rays_o[r] + rays_d[r]*((float)s/(N_samples-1))
 *                  (K7, render_utils_kernel.cu:254-257; pass rays_o / rays_d and n_steps[r] = N_samples).
 *
 * Scratch records.  Each ray owns a slice of the rec2/rec3 scratch arrays that starts at
 *   n_steps_cumsum[r] - n_steps[r]   when n_steps_cumsum != NULL (exact, M0 records in total), or
 *   r * rec_stride                   when n_steps_cumsum == NULL (rec_stride >= max n_steps; since
 *                                    t is clamped to [near,far], ceil((far-near)/stepdist)+1 is a
 *                                    bound -- saves the scan and the host read of M0).
 *
 * dvgo_march_density: one wavefront per ray.  sample -> bbox test -> mask lookup ->
 *   density trilinear -> raw2alpha -> alpha>thres filter -> transmittance scan with early
 *   stop (T<1e-3) -> weight>thres filter (both filters only when fast_color_thres > 0,
 *   lib/dvgo.py:478,488).  Writes per ray r:
 *     rec2[base+j] (j < n2[r]) : {step, exp_d, alpha, T_before} of the j-th sample that entered
 *                                compositing (post alpha filter, up to and including the early-stop
 *                                sample); step has bit 31 set when the sample also passed the
 *                                weight filter
 *   and n2[r] (records of the ray), n3[r] (how many of them passed the weight filter: the reference's final
 *   sample set), alphainv_last[r].  mask may be NULL (no occupancy skipping).
 * --------------------------------------------------------------------------------- */
typedef struct { int32_t step; float exp_d; float alpha; float T; } dvgo_rec2_t;     /* 16 B */

int dvgo_march_density(const float* rays_start, const float* rays_dir,
                       const int64_t* n_steps, const int64_t* n_steps_cumsum, int64_t rec_stride,
                       int64_t n_rays,
                       const float* xyz_min, const float* xyz_max, float stepdist,
                       const uint8_t* mask, int mX, int mY, int mZ,
                       const float* xyz2ijk_scale, const float* xyz2ijk_shift,
                       const float* density, int X, int Y, int Z,            /* [X,Y,Z] contiguous */
                       float act_shift, float interval, float fast_color_thres,
                       dvgo_rec2_t* rec2,
                       int32_t* n2, int32_t* n3, float* alphainv_last,
                       int32_t* brick_cnt /* NULL, or [dvgo_n_bricks(X,Y,Z)] zero-initialised counters: the samples
                                             that entered compositing are counted per brick for the backward's
                                             owner-computes scatter (see "Brick scatter" below) */,
                       void* stream);

/* dvgo_march_hit: hit[r] = 1 iff ray r has an in-box sample whose nearest occupancy voxel is set -- the fused
 *   form of DirectVoxGO.hit_coarse_geo (lib/dvgo.py:412-423), one wavefront per ray.  Host pointers as above. */
int dvgo_march_hit(const float* rays_start, const float* rays_dir, const int64_t* n_steps, int64_t n_rays,
                   const float* xyz_min, const float* xyz_max, float stepdist,
                   const uint8_t* mask, int mX, int mY, int mZ,
                   const float* xyz2ijk_scale, const float* xyz2ijk_shift, uint8_t* hit, void* stream);

/* exclusive scan of int32 counts -> int64 offsets [n+1] (offsets[n] = total) */
int dvgo_exclusive_scan_i32(const int32_t* counts, int64_t n, int64_t* offsets, void* stream);

/* dvgo_march_gather: one wavefront per ray over its rec2 records; the records flagged as kept are compacted to
 *   [off3[r], off3[r+1]) (ray-major, the reference's order; off3 = exclusive scan of n3, M3 = off3[N]).  Writes
 *   ray_id, step_id [M3] int64; weights (= T * alpha), alpha [M3]; k0 features [M3,C] trilinearly interpolated from
 *   the feature grid (element strides as dvgo_grid_sample_fwd). */
int dvgo_march_gather(const dvgo_rec2_t* rec2, const int32_t* n2, const int64_t* n_steps, const int64_t* n_steps_cumsum,
                      int64_t rec_stride, const int64_t* off3, int64_t n_rays, int64_t M3,
                      const float* rays_start, const float* rays_dir, float stepdist,
                      const float* xyz_min, const float* xyz_max,
                      const float* k0, int C, int X, int Y, int Z,
                      int64_t sC, int64_t sX, int64_t sY, int64_t sZ,
                      int64_t* ray_id, int64_t* step_id, float* weights, float* alpha,
                      float* feat, void* stream);

/* dvgo_march_composite: rgb_marched[r] = sum_k w*rgb + alphainv_last[r]*bg ; optional depth
 *   = sum_k w*step_id (lib/dvgo.py:554-576).  One wavefront per ray over [off3[r], off3[r+1]). */
int dvgo_march_composite(const float* weights, const float* rgb /* [M3,3] */,
                         const int64_t* step_id, const int64_t* off3, int64_t n_rays,
                         const float* alphainv_last, float bg,
                         float* rgb_marched /* [N,3] */, float* depth /* [N] or NULL */,
                         void* stream);

/* Backward of the composite w.r.t. weights and rgb (either output may be NULL):
 *   grad_weights[i] = sum_c g[r,c]*rgb[i,c] ; grad_rgb[i,c] = g[r,c]*weights[i], r = ray_id[i].
 *   grad_last (NULL or [n_rays]) = bg * sum_c g[r,c], the gradient w.r.t. alphainv_last (lib/dvgo.py:559),
 *   written by the same launch. */
int dvgo_march_composite_bwd(const float* grad_rgb_marched /* [N,3] */, const float* weights,
                             const float* rgb, const int64_t* ray_id, int64_t M3, const int64_t* m_dev, int64_t n_rays,
                             float bg, float* grad_weights /* [M3] */, float* grad_rgb /* [M3,3] */,
                             float* grad_last, void* stream);

/* dvgo_march_feat_bwd: scatter grad_feat [M3,C] into grad_k0 (float atomics, de-duplicated per wavefront in LDS;
 *   element strides sC,sX,sY,sZ of the destination).
 *   grad_extra (NULL or [M3]): one more per-sample scalar scattered with the same trilinear weights as channel C
 *   of the same voxel row -- used with a combined gradient buffer of 64-byte rows (sC = 1, sZ = 16 floats: 12
 *   feature channels, the density gradient, 3 pad): float atomics are bound by 64-B requests, an aligned row is
 *   exactly one, and the density gradient of the kept samples then costs no request of its own.  Needs
 *   C == 12 and a row layout with sZ > C, else DVGO_ERANGE. */
int dvgo_march_feat_bwd(const float* grad_feat, const float* grad_extra, const int64_t* ray_id, const int64_t* step_id,
                        int64_t M3, const float* rays_start, const float* rays_dir, float stepdist,
                        const float* xyz_min, const float* xyz_max,
                        int C, int X, int Y, int Z, int64_t sC, int64_t sX, int64_t sY, int64_t sZ,
                        float* grad_k0, void* stream);

/* dvgo_march_density_bwd: one wavefront per ray.  alpha2weight backward (K13) over the rec2
 *   samples, raw2alpha backward (K10), then the trilinear scatter into grad_density (voxel v at
 *   grad_density[v * grad_stride]; 1 for the plain [X,Y,Z] grid).
 *   grad_weights is indexed in the M3 order of dvgo_march_gather; grad_last may be NULL (= 0).
 *   grad_kept (NULL or [M3]): when given, the samples kept by dvgo_march_gather write their density gradient there
 *   (for dvgo_march_feat_bwd's grad_extra) instead of scattering it; only the dropped ones are scattered here.
 *   brick_cursor (NULL or [n_bricks], from dvgo_brick_scan): when given nothing is scattered here (grad_density and
 *   grad_kept are ignored): every sample is appended to the list of each brick it touches -- brick_recs[slot] =
 *   {int32 kept index in the M3 order or -1, int32 ray, int32 step, float density gradient} (16 B) -- for
 *   dvgo_brick_accumulate. */
int dvgo_march_density_bwd(const dvgo_rec2_t* rec2, const int32_t* n2, const int64_t* n_steps,
                           const int64_t* n_steps_cumsum, int64_t rec_stride, const int64_t* off3,
                           int64_t n_rays,
                           const float* rays_start, const float* rays_dir, float stepdist,
                           const float* xyz_min, const float* xyz_max,
                           const float* alphainv_last, float interval,
                           const float* grad_weights /* [M3] */, const float* grad_last /* [N] */,
                           int X, int Y, int Z, float* grad_density, int64_t grad_stride, float* grad_kept,
                           int32_t* brick_cursor, void* brick_recs,
                           void* stream);

/* ---------------------------------------------------------------------------------
 * Brick scatter: the grid-gradient sums of grid_sampler_3d_backward (behind lib/dvgo.py:321; 8*C float atomics per
 * sample in the reference) without atomics.  The lattice is cut into 8x8x8-voxel bricks; a workgroup owns one brick,
 * sums the contributions of the samples listed for it and either writes the dense gradients with plain stores
 * (every voxel of the grid is written: zeros where no sample came by) or applies the Adam update in place.
 *   dvgo_n_bricks        number of bricks of an [X,Y,Z] lattice
 *   dvgo_brick_scan      counts [n] -> offsets [n+1] (offsets[n] = number of list entries E) and fill cursors [n]
 *   dvgo_brick_accumulate
 *     brick_off [n+1]; recs [E] 16-byte records as written by dvgo_march_density_bwd; rays_start / rays_dir /
 *     stepdist / xyz_min / xyz_max (host) as given to dvgo_march_density (sample positions are rebuilt from the
 *     records' ray and step); grad_feat [M3,C] (the gradient w.r.t. the gathered features); density and k0 share the
 *     lattice, k0 channels-last.
 *     p_k0 == NULL : writes grad_k0 [X,Y,Z,C] (channels-last) and grad_density [X,Y,Z], all of both.
 *     p_k0 != NULL : MaskedAdam in place on (p, exp_avg, exp_avg_sq) of k0 and density
 *                    (adam_upd_cuda.adam_upd / masked_adam_upd, lib/cuda/adam_upd.cpp:36-68; step sizes bias-corrected
 *                    on the host as adam_upd_kernel.cu:72; masked_* = skip elements whose gradient is exactly 0);
 *                    voxels of bricks no sample touched are left alone either way (their gradient is zero: the masked
 *                    update skips them; for an unmasked group use the dense path).
 *     Built for C in {3, 4, 9, 12}; DVGO_ERANGE otherwise.
 * --------------------------------------------------------------------------------- */
int dvgo_n_bricks(int X, int Y, int Z);
int dvgo_brick_slice(void);     /* default slice_len: entries per work item of dvgo_brick_accumulate (heavy bricks are cut
                                 * into slices); the scan and the accumulation of one step must be given the same value (>= 256) */
/* counts [n] -> offsets [n+1] + fill cursors [n].  With extra_off / active [n+1] and extra_brick (all or none; then
 * n < 2^18 and the total must stay below 2^28 entries) also the work tables of dvgo_brick_accumulate:
 *   - a brick with more than slice_len entries runs as ceil(count / slice_len) work items -- its own plus EXTRA ones
 *     appended behind the bricks (extra_off[n] = their number <= E / slice_len, extra_brick[x] = the brick of extra item x);
 *   - active[0 .. active[n]) = the non-empty bricks in brick order;
 *   - brick_cnt is CLEARED: it becomes the arrival counter array of the slices. */
int dvgo_brick_scan(int32_t* brick_cnt, int n_bricks, int32_t* brick_off, int32_t* brick_cursor, int32_t* extra_off,
                    int32_t* active, int32_t* extra_brick /* [n_extra_max] */, int n_extra_max, int slice_len, void* stream);
/* the ray scan (n3 [n_rays] -> off3 [n_rays + 1], int64) and dvgo_brick_scan in one launch; brick_cnt == NULL: rays only */
int dvgo_march_scans(const int32_t* n3, int64_t n_rays, int64_t* off3, int32_t* brick_cnt, int n_bricks,
                     int32_t* brick_off, int32_t* brick_cursor, int32_t* extra_off, int32_t* active, int32_t* extra_brick,
                     int n_extra_max, int slice_len, void* stream);
/* extra_off == NULL: one workgroup per brick, whatever its list length.  Else the work tables of dvgo_brick_scan,
 * `arrive` = the cleared counters, `scratch` = 2 * n_extra_max tiles of 512 * round_up(C + 1, 4) floats, n_extra_max = a
 * host-side upper bound of extra_off[n] (the launch is ceil8(n_bricks) + n_extra_max workgroups). */
int dvgo_brick_accumulate(const int32_t* brick_off, const int32_t* extra_off, const int32_t* active,
                          const int32_t* extra_brick, int32_t* arrive, float* scratch, int64_t n_extra_max, int slice_len,
                          const void* recs, const float* rays_start, const float* rays_dir,
                          float stepdist, const float* xyz_min, const float* xyz_max, const float* grad_feat,
                          int C, int X, int Y, int Z, float* grad_k0, float* grad_density,
                          float* p_k0, float* m_k0, float* v_k0, float step_size_k0, int masked_k0,
                          float* p_density, float* m_density, float* v_density, float step_size_density,
                          int masked_density, float beta1, float beta2, float eps,
                          const float* step_sizes_dev /* NULL, or device {step_size_k0, step_size_density}: read instead of
                                                         the two float arguments (a captured step replays with new values) */,
                          void* stream);

/* Combined gradient rows G [n_vox][row_stride] (built by the two calls above) -> the channels-last feature
 * gradient [n_vox][C] and the density gradient [n_vox], both fully written.  Built for row_stride 16, C 12. */
int dvgo_grid_grad_split(const float* G, int64_t n_vox, int row_stride, int C, float* grad_k0, float* grad_density,
                         void* stream);

/* ---------------------------------------------------------------------------------
 * "next" row N4: coarse-stage grid maintenance.
 *   voxel_count_views (lib/dvgo.py:265-295): per training view, acc[v] = sum of the trilinear weights voxel v receives
 *   from the view's sample points (o + d * (t_min + k * step / |d|), k < n_samples, t_min = slab entry clamped to
 *   [near, far]); count[v] += (acc[v] > 1).  Call dvgo_view_weight_accumulate for the rays of ONE view (any number of
 *   calls), then dvgo_view_count_commit, which also clears acc for the next view.  acc, count: [X*Y*Z] floats, acc zero
 *   on first use.  xyz_min / xyz_max are HOST pointers.
 *   maskout_near_cam_vox (lib/dvgo.py:215-226): density[v] = value where min_c |xyz(v) - cam_o[c]| <= near;
 *   grid_x/y/z are the per-axis voxel-centre coordinates (device, [X] / [Y] / [Z]), cam_o [n_cam,3].
 * --------------------------------------------------------------------------------- */
int dvgo_view_weight_accumulate(const float* rays_o, const float* rays_d, int64_t n_rays, const float* xyz_min,
                                const float* xyz_max, float near, float far, float step, int n_samples, int X, int Y,
                                int Z, float* acc, void* stream);
int dvgo_view_count_commit(float* acc, float* count, int64_t n_vox, void* stream);
int dvgo_maskout_near_cam(float* density, const float* grid_x, const float* grid_y, const float* grid_z, int X, int Y, int Z,
                          const float* cam_o, int n_cam, float near, float value, void* stream);

/* ---------------------------------------------------------------------------------
 * "next" row N3: fused colour head (rgbnet) forward.  Replaces lib/dvgo.py:516-541 for the
 * Sequential(Linear(d_in,width), ReLU, Linear(width,width), ReLU, Linear(width,3)) head:
 *   x   = cat([feat[:, 3:] if diffuse else feat, emb[ray_id]])       d_in = C - (3 if diffuse) + E
 *   rgb = sigmoid(MLP(x) + (feat[:, :3] if diffuse))
 * feat [M,C], emb [N_rays,E] (view-direction embedding per ray), ray_id [M] int64.  Weights in
 * nn.Linear layout ([out,in] row-major).  fp32 throughout (v_mfma_f32_32x32x2_f32).
 * Training (all or none): H1, H2 [M,width] = post-ReLU activations, row-major (operands of the weight
 * gradients), and masks [M][2 layers][2 lane halves] uint64 = their sign bits in accumulator order (the ReLU
 * masks of dvgo_shade_bwd: 32 B per sample instead of re-reading 1 KB of activations).
 * Returns DVGO_ERANGE for shapes outside the built set (width == 128, d_in <= 40): fall back.
 * --------------------------------------------------------------------------------- */
int dvgo_shade_fwd(const float* feat, int C, const float* emb, int E, const int64_t* ray_id, int64_t M, const int64_t* m_dev,
                   const float* W1, const float* b1, const float* W2, const float* b2,
                   const float* W3, const float* b3, int width, int d_in, int diffuse,
                   float* rgb, float* H1, float* H2, uint64_t* masks,
                   void* scratch /* NULL, or dvgo_shade_scratch_bytes(width) bytes of device memory */, void* stream);

/* Kernel variants of the colour head (process-global, for A/B runs; returns the previous value, negative = query):
 *   bit 0  forward, bit 1 data gradients on the bf16 matrix cores: every fp32 operand split EXACTLY into three bf16
 *          pieces, six partial products per k-step accumulated in fp32 (csrc/shade_x3.hip) -- fp32-grade results at 2.7x
 *          fewer matrix-pipe cycles than v_mfma_f32_32x32x2_f32.  Default 3; the f32-MFMA kernels run when the bit is
 *          clear or `scratch` is NULL (the bf16 variants keep the pre-split weight image there, built by a small kernel
 *          at every call). */
int64_t dvgo_shade_scratch_bytes(int width);
int dvgo_shade_variant(int flags);

/* Data-gradient part of the colour-head backward.  Inputs: g_rgb, rgb [M,3]; the saved activations
 * sign-bit masks written by dvgo_shade_fwd.  Outputs: gz [M,3] (= g_rgb * sigmoid'),
 * G1 = relu'(H1) * (W2^T G2) as [M,width] where G2 = relu'(H2) * (W3^T gz) stays in registers (dvgo_shade_wgrad
 * rebuilds it from gz and the masks rather than streaming 512 B/sample), and g_feat [M,C] fully written:
 * channels [0,3) = gz when diffuse, channels [c0, C) = (W1^T G1)[:C-c0]. */
int dvgo_shade_bwd(const float* g_rgb, const float* rgb, const uint64_t* masks, int64_t M, const int64_t* m_dev,
                   const float* W1, const float* W2, const float* W3, int width, int d_in, int C, int diffuse,
                   float* g_feat, float* G1, float* gz, void* scratch, void* stream);

/* Weight-gradient part (G1, gz from dvgo_shade_bwd; masks, H1, H2 from dvgo_shade_fwd; W3 as given to both):
 * dW2 = G2^T H1, dW1 = G1^T X (X = the layer-1 input, re-assembled from feat / emb /
 * ray_id exactly as in dvgo_shade_fwd), dW3 = gz^T H2 and the three bias gradients (column sums); fp32 MFMA
 * with both operands read row-major straight from memory.  Every one of the n_parts workgroups writes its
 * partial sums to part[p] = { dW2 [width][width], dW1 [width][64], dW3 [32][width] (rows 0..2 valid),
 * db1 [width], db2 [width], db3 [width] (db3[c] = entry c + entry 8+c, c < 3) } floats (scratch), and a second launch sums them over p
 * into `total` (same record layout, once). */
int dvgo_shade_wgrad(const float* G1, const float* gz, const uint64_t* masks, const float* W3,
                     const float* H1, const float* H2, const float* feat, int C, const float* emb, int E, const int64_t* ray_id, int64_t M,
                     const int64_t* m_dev, int width, int diffuse, int n_parts, float* part, float* total, void* stream);

/* ---------------------------------------------------------------------------------
 * "next" rows N1/N2: optimizer and regulariser kernels.
 *   adam_upd_cuda.{adam_upd,masked_adam_upd,adam_upd_with_perlr}
 *     (lib/cuda/adam_upd.cpp:36-86, adam_upd_kernel.cu) -- mode 0/1/2, in place.
 *     step_size is computed by the caller exactly as the reference host code does (:72).
 *   total_variation_cuda.total_variation_add_grad
 *     (lib/cuda/total_variation.cpp:16-24, total_variation_kernel.cu) -- in place on grad.
 * --------------------------------------------------------------------------------- */
int dvgo_adam_upd(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                  const float* perlr /* mode 2 only */, int64_t n, float step_size,
                  float beta1, float beta2, float eps, int mode, void* stream);

/* Adam / masked Adam (mode 0 / 1, as dvgo_adam_upd) for the feature grid (channels-last rows [n_vox][C]) and the
 * density grid [n_vox] straight from the combined gradient rows G [n_vox][row_stride] of dvgo_march_feat_bwd /
 * dvgo_march_density_bwd -- instead of dvgo_grid_grad_split followed by two dvgo_adam_upd.  Built for row_stride 16,
 * C 12.  step_size_* as for dvgo_adam_upd (computed by the caller in float, adam_upd_kernel.cu:72). */
int dvgo_adam_rows(const float* G, int64_t n_vox, int row_stride, int C,
                   float* p_k0, float* m_k0, float* v_k0, float step_size_k0, int mode_k0,
                   float* p_density, float* m_density, float* v_density, float step_size_density, int mode_density,
                   float beta1, float beta2, float eps, void* stream);

int dvgo_total_variation_add_grad(const float* param, float* grad, float wx, float wy, float wz,
                                  int64_t C, int64_t sz_i, int64_t sz_j, int64_t sz_k,
                                  int64_t sC, int64_t sI, int64_t sJ, int64_t sK,
                                  int dense_mode, void* stream);
/* The same, restricted to the planes [i_lo, i_hi) of the first spatial axis (only their `grad` is touched; `param` is
 * read one plane beyond on either side).  For a data-parallel rank that owns one slab of the grid.  Needs a layout whose
 * outermost memory axis is that spatial axis (channels-last, or C == 1), else DVGO_ERANGE. */
int dvgo_total_variation_add_grad_slab(const float* param, float* grad, float wx, float wy, float wz,
                                       int64_t C, int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t sC,
                                       int64_t sI, int64_t sJ, int64_t sK, int dense_mode, int64_t i_lo, int64_t i_hi,
                                       void* stream);

/* ---------------------------------------------------------------------------------
 * Harness glue (row H3): fused training loss, view-direction embedding, multi-tensor Adam.
 *
 * dvgo_loss_fwd_bwd: the loss of run.py:377-386 and its gradients in one pass.
 *   loss = w_main * sum((rgb_marched - target)^2) / (3 n_rays_global)
 *        + w_ent  * sum(-(p log p + (1-p) log(1-p))) / n_rays_global,  p = clamp(alphainv_last, 1e-6, 1-1e-6)
 *        + w_per  * sum_i weights_i * |raw_rgb_i - target[ray_id_i]|^2 / n_rays_global   (weights detached)
 *   Writes loss_out[0] (device scalar) and d loss / d {rgb_marched [N,3], alphainv_last [N], raw_rgb [M,3]}.
 *   n_rays_global is the ray count over all ranks (data parallel).
 * dvgo_viewdir_embed: emb [N, 3 + 6F] = cat([v, sin(v (x) freq), cos(v (x) freq)]) (lib/dvgo.py:524-525).
 * dvgo_adam_upd_multi: plain Adam (mode 0 of dvgo_adam_upd) over up to 16 small tensors in one launch;
 *   the pointer tables and numel are HOST arrays.
 * --------------------------------------------------------------------------------- */
int dvgo_loss_fwd_bwd(const float* rgb_marched, const float* alphainv_last, const float* target, int64_t N,
                      const float* raw_rgb, const float* weights, const int64_t* ray_id, int64_t M, const int64_t* m_dev,
                      int64_t n_rays_global, float w_main, float w_ent, float w_per,
                      float* g_marched, float* g_last, float* g_raw_rgb, float* loss_out, void* stream);
int dvgo_viewdir_embed(const float* viewdirs, const float* freq, int n_freq, int64_t N, float* emb, void* stream);
int dvgo_adam_upd_multi(float* const* params, const float* const* grads, float* const* exp_avg,
                        float* const* exp_avg_sq, const int64_t* numel, int n_tensors, float step_size,
                        float beta1, float beta2, float eps,
                        const float* step_size_dev /* NULL, or the step size on the device */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DVGO_HIP_H */
