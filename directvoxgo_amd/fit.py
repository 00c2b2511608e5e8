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


def fit_stage(model, rays_o, rays_d, viewdirs, target, cfg_train, render_kwargs, n_iters=None, num_voxels_final=None,
              seed=777, log_every=0):
    """rays_* / target: flat [N,3] device tensors.  `num_voxels_final` is the resolution reached after the last
    `pg_scale` step (run.py:243-245 builds the model at num_voxels_final / 2^len(pg_scale); a model handed over at
    another resolution is resized to that first).  Returns the per-step PSNR list (python floats)."""
    n_iters = n_iters or cfg_train['N_iters']
    n_rand = cfg_train['N_rand']
    pg_scale = list(cfg_train.get('pg_scale', []))
    num_voxels_final = num_voxels_final or model.num_voxels
    if pg_scale:
        start = int(num_voxels_final / (2 ** len(pg_scale)))
        if model.num_voxels != start:
            model.scale_volume_grid(start)
    step = TrainStep(model, cfg_train, render_kwargs)
    batches = batch_indices_generator(rays_o.shape[0], n_rand, seed=seed)
    psnrs = []
    for global_step in range(1, n_iters + 1):
        if model.mask_cache is not None and (global_step + 500) % 1000 == 0:
            with torch.no_grad():
                self_alpha = F.max_pool3d(model.activate_density(model.density), kernel_size=3, padding=1, stride=1)[0, 0]
                model.mask_cache.mask &= (self_alpha > model.fast_color_thres)
        if global_step in pg_scale:
            n_rest = len(pg_scale) - pg_scale.index(global_step) - 1
            model.scale_volume_grid(int(num_voxels_final / (2 ** n_rest)))
            step = TrainStep(model, cfg_train, render_kwargs,
                             optimizer=create_optimizer_or_freeze_model(model, cfg_train, global_step=0))
            model.density.data.sub_(1)
        sel = next(batches).to(rays_o.device)
        loss = step(rays_o[sel], rays_d[sel], viewdirs[sel], target[sel], global_step)
        psnrs.append(float(-10.0 * torch.log10(loss)))                    # utils.mse2psnr of the full loss, run.py:378
        if log_every and global_step % log_every == 0:
            print(f'fit_stage: iter {global_step:6d} loss {float(loss):.6f} psnr {sum(psnrs[-log_every:]) / log_every:.2f}')
    return psnrs
