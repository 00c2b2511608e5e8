"""Full-image inference throughput (BASELINE config 5 shape: 800x800 views in 8192-ray chunks, render_depth on,
no_grad) on the synthetic lego-like scene.   python tools/render_bench.py [--world 256] [--views 3]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from directvoxgo_amd.dvgo import DirectVoxGO
from directvoxgo_amd.render import render_viewpoints
from directvoxgo_amd.scenes import pose_spherical, synthetic_scene

ap = argparse.ArgumentParser()
ap.add_argument('--world', type=int, default=256)
ap.add_argument('--views', type=int, default=3)
ap.add_argument("--hw", type=int, default=800)
ap.add_argument("--chunk", type=int, default=65536)
ap.add_argument("--profile", action="store_true", help="per-kernel HIP-event times of one view")
args = ap.parse_args()

sc = synthetic_scene(world=args.world, n_rays=8, device='cuda')
torch.manual_seed(0)
m = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=args.world ** 3, num_voxels_base=args.world ** 3, alpha_init=1e-2,
                fast_color_thres=1e-4, rgbnet_dim=12, rgbnet_width=128, rgbnet_direct=True).cuda()
with torch.no_grad():
    m.density.copy_(sc['density']); m.k0.copy_(sc['k0']); m.mask_cache.mask.copy_(sc['mask'])
H = W = args.hw
K = np.array([[1111.11 * W / 800, 0, 0.5 * W], [0, 1111.11 * H / 800, 0.5 * H], [0, 0, 1]], np.float32)
poses = [pose_spherical(40.0 * i - 60, -30.0, 4.0).numpy() for i in range(args.views)]
rk = dict(near=2.0, far=6.0, bg=1, stepsize=0.5, inverse_y=False)
render_viewpoints(m, poses[:1], [(H, W)], [K], False, rk)            # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
rgbs, depths = render_viewpoints(m, poses, [(H, W)] * len(poses), [K] * len(poses), False, rk, chunk=args.chunk)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / len(poses)
print(f'grid {args.world}^3 occupancy {sc["occupancy"]:.3f}: {dt * 1e3:.1f} ms per {H}x{W} view ({H * W / dt / 1e6:.2f} M rays/s), '
      f'mean rgb {rgbs.mean():.3f}, mean depth {depths.mean():.1f}')

if args.profile:
    from directvoxgo_amd import _lib as L
    names = ['dvgo_sample_pts_prepare', 'dvgo_march_density', 'dvgo_exclusive_scan_i32', 'dvgo_march_gather',
             'dvgo_march_composite', 'dvgo_shade_fwd', 'dvgo_viewdir_embed']
    L.profile_start(names)
    render_viewpoints(m, poses[:1], [(H, W)], [K], False, rk, chunk=args.chunk)
    torch.cuda.synchronize()
    prof = L.profile_stop()
    print({k: (c, round(ms, 3)) for k, (c, ms) in prof.items()}, 'sum', round(sum(ms for _, ms in prof.values()), 3))
