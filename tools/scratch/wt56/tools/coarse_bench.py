"""Train-step time of the coarse stage (configs/default.py coarse_*: 100^3 grid, k0 = RGB, no MLP, per-voxel lr off here,
8192 rays) on the synthetic lego-like scene.   python tools/coarse_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from directvoxgo_amd import _lib as L
from directvoxgo_amd.dvgo import DirectVoxGO
from directvoxgo_amd.scenes import synthetic_scene
from directvoxgo_amd.train import COARSE_TRAIN, TrainStep

dev = 'cuda'
W = 100
sc = synthetic_scene(world=W, n_rays=8192, device=dev)
m = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=W ** 3, num_voxels_base=W ** 3, alpha_init=1e-6, fast_color_thres=1e-7,
                rgbnet_dim=0).to(dev)
with torch.no_grad():
    m.density.copy_(sc['density'] * 0.2 - 2.0)       # early-training-like: soft density everywhere, nothing masked out
    m.mask_cache.mask.fill_(True)
cfg = dict(COARSE_TRAIN, pervoxel_lr=False)
rk = dict(near=sc['near'], far=sc['far'], bg=1, stepsize=0.5)
step = TrainStep(m, cfg, rk)
args = (sc['rays_o'], sc['rays_d'], sc['viewdirs'], sc['target'])
for i in range(5):
    step(*args, global_step=10 + i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(30):
    step(*args, global_step=20 + i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 30
names = ['dvgo_sample_pts_prepare', 'dvgo_march_density', 'dvgo_exclusive_scan_i32', 'dvgo_march_gather', 'dvgo_march_composite',
         'dvgo_march_composite_bwd', 'dvgo_march_feat_bwd', 'dvgo_march_density_bwd', 'dvgo_grid_grad_split', 'dvgo_adam_upd',
         'dvgo_loss_fwd_bwd']
L.profile_start(names)
for i in range(30):
    step(*args, global_step=60 + i)
torch.cuda.synchronize()
prof = {k: round(ms / 30, 4) for k, (c, ms) in L.profile_stop().items() if c}
res = m(*args[:3], **rk)
print(f'{dt * 1e3:.3f} ms / step ({8192 / dt / 1e6:.2f} M rays/s); kept samples / ray {res["weights"].numel() / 8192:.1f}')
print(prof, 'sum', round(sum(prof.values()), 3))
