"""Per-kernel times of the backward grid-gradient path on the config-2 roofline case (160^3, 8192 rays x 256 samples):
the brick scatter with the Adam update fused in, the brick scatter writing dense gradients (+ the dense Adam kernels),
and the atomic scatters it replaced.      python tools/brick_bench.py [--steps 20] [--workload roofline|lego]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from directvoxgo_amd import _lib as L
from directvoxgo_amd import fused as F
from directvoxgo_amd.dvgo import DirectVoxGO
from directvoxgo_amd.scenes import roofline_scene, synthetic_scene
from directvoxgo_amd.train import FINE_TRAIN, TrainStep

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--world', type=int, default=160)
ap.add_argument('--rays', type=int, default=8192)
ap.add_argument('--workload', default='roofline')
ap.add_argument('--slice', type=int, default=0, help='entries per work item of the brick kernel (0: library default)')
ap.add_argument('--hist', action='store_true', help='print the distribution of per-brick list lengths')
args = ap.parse_args()

NAMES = ['dvgo_sample_pts_prepare', 'dvgo_march_density', 'dvgo_march_scans', 'dvgo_exclusive_scan_i32', 'dvgo_march_gather', 'dvgo_march_composite', 'dvgo_march_composite_bwd', 'dvgo_shade_fwd', 'dvgo_shade_bwd', 'dvgo_shade_wgrad', 'dvgo_brick_scan', 'dvgo_march_density_bwd', 'dvgo_brick_accumulate', 'dvgo_march_feat_bwd',
         'dvgo_grid_grad_split', 'dvgo_adam_rows', 'dvgo_adam_upd']


def run(tag, brick, rows_adam):
    mk = roofline_scene if args.workload == 'roofline' else synthetic_scene
    sc = mk(world=args.world, n_rays=args.rays, seed=777, device='cuda')
    torch.manual_seed(777)
    m = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=args.world ** 3, num_voxels_base=args.world ** 3, alpha_init=1e-2,
                    fast_color_thres=1e-4, rgbnet_dim=12, rgbnet_width=128, rgbnet_direct=True).cuda()
    with torch.no_grad():
        m.density.copy_(sc['density']); m.k0.copy_(sc['k0']); m.mask_cache.mask.copy_(sc['mask'])
    F.BRICK_SCATTER = brick
    F.BRICK_SLICE = args.slice or None
    if args.hist and brick and rows_adam:
        import numpy as np
        cfg = m._march_cfg(sc['near'], sc['far'], 0.5)
        out = F.fused_march(m.density, m.k0, sc['rays_o'], sc['rays_d'], cfg)
        c = np.diff(out[3].grad_fn.bricks[0][0].cpu().numpy())
        nz = c[c > 0]
        print('bricks', len(c), 'non-empty', len(nz), 'entries', int(c.sum()), 'percentiles 50/90/99/max',
              [int(np.percentile(nz, q)) for q in (50, 90, 99, 100)])
    step = TrainStep(m, dict(FINE_TRAIN), dict(near=sc['near'], far=sc['far'], bg=1, stepsize=0.5), rows_adam=rows_adam,
                     overlap_wgrad=False)
    b = (sc['rays_o'], sc['rays_d'], sc['viewdirs'], sc['target'])
    for i in range(3):
        step(*b, global_step=5000 + i)
    torch.cuda.synchronize()
    L.profile_start(NAMES)
    for i in range(args.steps):
        step(*b, global_step=5003 + i)
    prof = L.profile_stop()
    F.BRICK_SCATTER = True
    row = {k: round(ms / c * 1e3, 1) for k, (c, ms) in prof.items() if c}
    print(f'{tag:34s}', ' '.join(f'{k[5:]}={v}' for k, v in row.items()), ' sum(us)=', round(sum(ms / args.steps for c, ms in prof.values() if c) * 1e3, 1))


run('brick + fused Adam', True, True)
run('brick -> dense grads + dense Adam', True, False)
if not args.slice:
    run('atomic rows + adam_rows', False, True)
    run('atomic rows -> split + dense Adam', False, False)
