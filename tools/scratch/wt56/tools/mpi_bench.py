"""Train-step time of the config-4 shape (configs/llff/llff_default.py: DirectMPIGO, 256^3 voxels with 128 MPI planes,
k0_dim 9, 64-wide head, 4096 NDC rays per step, dense TV on both grids every step) on synthetic forward-facing rays.
    python tools/mpi_bench.py [--steps 20]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from directvoxgo_amd import _lib as L
from directvoxgo_amd.dmpigo import DirectMPIGO
from directvoxgo_amd.train import FINE_TRAIN, TrainStep

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--rays', type=int, default=4096)
ap.add_argument('--voxels', type=int, default=256)
args = ap.parse_args()
dev = 'cuda'
torch.manual_seed(0)
mn, mx = np.array([-1.5, -1.67, -1.0], np.float32), np.array([1.5, 1.67, 1.0], np.float32)   # llff bbox (run.py:232-236)
m = DirectMPIGO(mn, mx, num_voxels=args.voxels ** 3, mpi_depth=128, fast_color_thres=1e-3, rgbnet_dim=9, rgbnet_depth=3,
                rgbnet_width=64, viewbase_pe=0).to(dev)
print('world_size', m.world_size.tolist(), 'k0', tuple(m.k0.shape))
with torch.no_grad():
    m.density.add_(torch.randn_like(m.density) * 2.0)          # some structure: ~half of the samples pass the alpha threshold
    m.k0.copy_(torch.randn_like(m.k0) * 0.3)
cfg = dict(FINE_TRAIN, N_iters=25000, N_rand=args.rays, pg_scale=[2000, 4000, 6000, 8000], tv_before=1e9, tv_dense_before=10000,
           weight_tv_density=1e-5, weight_tv_k0=1e-5, skip_zero_grad_fields=['density', 'k0'])      # llff_default.py:14-24
rk = dict(near=0, far=1, bg=1, stepsize=1.0, inverse_y=False, flip_x=False, flip_y=False)
g = torch.Generator(device=dev).manual_seed(1)


def batch():
    ro = torch.cat([torch.rand(args.rays, 2, device=dev, generator=g) * 2.4 - 1.2, -torch.ones(args.rays, 1, device=dev)], 1)
    rd = torch.cat([torch.rand(args.rays, 2, device=dev, generator=g) * 0.6 - 0.3, 2 * torch.ones(args.rays, 1, device=dev)], 1)
    return ro, rd, rd / rd.norm(dim=-1, keepdim=True), torch.rand(args.rays, 3, device=dev, generator=g)


step = TrainStep(m, cfg, rk)
pool = [batch() for _ in range(4)]
for i in range(3):
    step(*pool[i % 4], global_step=100 + i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(args.steps):
    step(*pool[i % 4], global_step=200 + i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
names = ['dvgo_march_density', 'dvgo_march_gather', 'dvgo_march_feat_bwd', 'dvgo_march_density_bwd', 'dvgo_shade_fwd',
         'dvgo_shade_bwd', 'dvgo_shade_wgrad', 'dvgo_total_variation_add_grad', 'dvgo_adam_upd', 'dvgo_grid_grad_split']
step.overlap_wgrad = False
L.profile_start(names)
for i in range(args.steps):
    step(*pool[i % 4], global_step=300 + i)
torch.cuda.synchronize()
prof = {k: round(ms / args.steps, 3) for k, (c, ms) in L.profile_stop().items() if c}
res = m(*pool[0][:3], **rk)
print(f'{dt * 1e3:.2f} ms / step, {args.rays / dt / 1e6:.2f} M rays/s; kept samples per ray {res["weights"].numel() / args.rays:.1f}')
print('per step ms:', prof, 'sum', round(sum(prof.values()), 3))
