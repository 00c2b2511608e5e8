"""Optimisation loop of one stage -- harness counterpart of /root/reference/run.py:199-437
(`scene_rep_reconstruction`), reduced to what touches the hot path (row H3 of SURVEY.md section 8a):

  * occupancy refresh every 1000 steps at (step + 500) % 1000 == 0                   run.py:329-332
  * progressive grid growth at `pg_scale` steps: scale_volume_grid, NEW optimizer,
    density -= 1                                                                      run.py:334-345
  * batch draw from a permutation stream of the training rays                        run.py:348-353
  * TrainStep (forward, loss, backward, DP reduction, TV, MaskedAdam, lr decay)       run.py:372-406

No CLI, config files, logging to disk or dataset I/O: those are out of scope.
"""
import torch
import torch.nn.functional as F

from .render import batch_indices_generator
from .train import TrainStep, create_optimizer_or_freeze_model


def _scale_volume_grid(model, num_voxels):
    """run.py:335-341: DirectMPIGO.scale_volume_grid takes (num_voxels, mpi_depth) (lib/dmpigo.py:110)."""
    if hasattr(model, 'mpi_depth'):
        model.scale_volume_grid(num_voxels, model.mpi_depth)
    else:
        model.scale_volume_grid(num_voxels)


def per_voxel_init(model, optimizer, rays_o, rays_d, imsz, near, far, stepsize, downrate=1):
    """run.py:311-320: view counts -> per-voxel learning rate, and density = -100 where at most two views look."""
    cnt = model.voxel_count_views(rays_o_tr=rays_o, rays_d_tr=rays_d, imsz=imsz, near=near, far=far, stepsize=stepsize,
                                  downrate=downrate, irregular_shape=True)
    optimizer.set_pervoxel_lr(cnt)
    with torch.no_grad():
        model.density[cnt <= 2] = -100
    return cnt


def fit_stage(model, rays_o, rays_d, viewdirs, target, cfg_train, render_kwargs, n_iters=None, num_voxels_final=None,
              seed=777, log_every=0, imsz=None):
    """rays_* / target: flat [N,3] device tensors, view after view; `imsz` = rays per view (needed when
    cfg_train['pervoxel_lr'], run.py:311-320).  `num_voxels_final` is the resolution reached after the last
    `pg_scale` step (run.py:243-245 builds the model at num_voxels_final / 2^len(pg_scale); a model handed over at
    another resolution is resized to that first).  Returns the per-step PSNR list (python floats; PSNR of the main MSE
    term as logged by run.py:378, read back once at the end)."""
    n_iters = n_iters or cfg_train['N_iters']
    n_rand = cfg_train['N_rand']
    pg_scale = list(cfg_train.get('pg_scale', []))
    num_voxels_final = num_voxels_final or model.num_voxels
    if pg_scale:
        start = int(num_voxels_final / (2 ** len(pg_scale)))
        if model.num_voxels != start:
            _scale_volume_grid(model, start)
    step = TrainStep(model, cfg_train, render_kwargs, track_mse=True)
    if cfg_train.get('pervoxel_lr', False):
        if imsz is None:
            raise ValueError("cfg_train['pervoxel_lr'] needs `imsz` (rays per training view) for voxel_count_views")
        per_voxel_init(model, step.optimizer, rays_o, rays_d, list(imsz), render_kwargs['near'], render_kwargs['far'],
                       render_kwargs['stepsize'], cfg_train.get('pervoxel_lr_downrate', 1))
    batches = batch_indices_generator(rays_o.shape[0], n_rand, seed=seed)
    mses = []
    for global_step in range(1, n_iters + 1):
        if model.mask_cache is not None and (global_step + 500) % 1000 == 0:
            with torch.no_grad():
                self_alpha = F.max_pool3d(model.activate_density(model.density), kernel_size=3, padding=1, stride=1)[0, 0]
                model.mask_cache.mask &= (self_alpha > model.fast_color_thres)
        if global_step in pg_scale:
            n_rest = len(pg_scale) - pg_scale.index(global_step) - 1
            _scale_volume_grid(model, int(num_voxels_final / (2 ** n_rest)))
            step = TrainStep(model, cfg_train, render_kwargs, track_mse=True,
                             optimizer=create_optimizer_or_freeze_model(model, cfg_train, global_step=0))
            model.density.data.sub_(1)
        sel = next(batches).to(rays_o.device)
        step(rays_o[sel], rays_d[sel], viewdirs[sel], target[sel], global_step)
        mses.append(step.last_mse)
        if log_every and global_step % log_every == 0:
            recent = -10.0 * torch.log10(torch.stack(mses[-log_every:]))
            print(f'fit_stage: iter {global_step:6d} psnr {float(recent.mean()):.2f}')
    return (-10.0 * torch.log10(torch.stack(mses))).cpu().tolist() if mses else []


# ----------------------------------------------------------------------------------------------
# Scene bounds and the coarse -> fine flow (run.py:155-196, 440-492)
# ----------------------------------------------------------------------------------------------
@torch.no_grad()
def compute_bbox_by_cam_frustrm(HW, Ks, poses, near, far, ndc=False, inverse_y=False, flip_x=False, flip_y=False,
                                device='cpu'):
    """Axis-aligned box around every training ray's near and far point (run.py:155-173)."""
    from .render import get_rays_of_a_view
    import numpy as np
    lo = torch.full((3,), float('inf'), device=device)
    hi = -lo
    for (H, W), K, c2w in zip(HW, Ks, poses):
        c2w = torch.as_tensor(np.asarray(c2w), dtype=torch.float32, device=device)
        rays_o, rays_d, viewdirs = get_rays_of_a_view(int(H), int(W), K, c2w, ndc, inverse_y, flip_x, flip_y)
        step = rays_d if ndc else viewdirs
        for t in (near, far):
            pts = (rays_o + step * t).reshape(-1, 3)
            lo = torch.minimum(lo, pts.amin(0))
            hi = torch.maximum(hi, pts.amax(0))
    return lo, hi


@torch.no_grad()
def compute_bbox_by_coarse_geo(model, thres):
    """Tight box around the voxels the coarse model considers occupied (run.py:175-196)."""
    ws = model.density.shape[2:]
    dev = model.density.device
    interp = torch.stack(torch.meshgrid(*[torch.linspace(0, 1, int(n), device=dev) for n in ws], indexing='ij'), -1)
    xyz_min, xyz_max = model.xyz_min.to(dev), model.xyz_max.to(dev)
    dense_xyz = xyz_min * (1 - interp) + xyz_max * interp
    alpha = model.activate_density(model.grid_sampler(dense_xyz, model.density))
    active = dense_xyz[alpha > thres]
    return active.amin(0), active.amax(0)


def train_two_stage(model_class, xyz_min, xyz_max, rays_o, rays_d, viewdirs, target, render_kwargs, coarse_model, fine_model,
                    coarse_train, fine_train, ckpt_dir, bbox_thres=1e-3, world_bound_scale=1.05, device='cuda', imsz=None,
                    cam_o=None):
    """The reference's train() flow (run.py:440-492) on in-memory rays: coarse stage -> checkpoint ->
    bounds from the coarse geometry -> fine stage whose occupancy grid is seeded from the coarse checkpoint
    (mask_cache_path) and whose batches only contain rays that hit it.  Returns (fine_model, psnr lists)."""
    import os
    from .checkpoint import save_checkpoint
    coarse_model = dict(coarse_model)
    near_cam = coarse_model.pop('maskout_near_cam_vox', False)                         # configs/default.py:90
    coarse = model_class(xyz_min, xyz_max, **coarse_model).to(device)
    if near_cam:                                                                        # run.py:251-252
        if cam_o is None:
            raise ValueError('maskout_near_cam_vox needs the camera centres `cam_o`')
        coarse.maskout_near_cam_vox(cam_o, render_kwargs['near'])
    ps_c = fit_stage(coarse, rays_o, rays_d, viewdirs, target, coarse_train, render_kwargs, n_iters=coarse_train['N_iters'],
                     imsz=imsz)
    ckpt = os.path.join(ckpt_dir, 'coarse_last.tar')
    save_checkpoint(ckpt, coarse, None, coarse_train['N_iters'])
    lo, hi = compute_bbox_by_coarse_geo(coarse, bbox_thres)
    shift = (hi - lo) * (world_bound_scale - 1) / 2                                   # run.py:476-479 (bound scale)
    lo, hi = (lo - shift).cpu(), (hi + shift).cpu()
    fine = model_class(lo, hi, mask_cache_path=ckpt, **fine_model).to(device)
    hit = torch.cat([fine.hit_coarse_geo(rays_o=rays_o[i:i + 65536], rays_d=rays_d[i:i + 65536], **render_kwargs)
                     for i in range(0, rays_o.shape[0], 65536)])                      # 'in_maskcache' ray sampler
    ps_f = fit_stage(fine, rays_o[hit], rays_d[hit], viewdirs[hit], target[hit], fine_train, render_kwargs,
                     n_iters=fine_train['N_iters'], num_voxels_final=fine_model['num_voxels'])
    return fine, (ps_c, ps_f)
