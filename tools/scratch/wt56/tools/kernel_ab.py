"""GPU micro-benchmark of the fused-march kernels on the roofline case (160^3, 8192 x 256), with
A/B variants interleaved in one process (guide rule 24).  Prints avg/min us per kernel and checks
that variants agree.   python tools/kernel_ab.py [--world 160] [--rays 8192] [--rounds 10]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from directvoxgo_amd import _lib as L
from directvoxgo_amd.dvgo import DirectVoxGO
from directvoxgo_amd.scenes import roofline_scene, synthetic_scene

ap = argparse.ArgumentParser()
ap.add_argument('--world', type=int, default=160)
ap.add_argument('--rays', type=int, default=8192)
ap.add_argument('--rounds', type=int, default=10)
ap.add_argument('--workload', default='roofline')
args = ap.parse_args()

dev = 'cuda'
sc = roofline_scene(world=args.world, n_rays=args.rays, device=dev) if args.workload == 'roofline' else \
    synthetic_scene(world=args.world, n_rays=args.rays, device=dev)
m = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=args.world ** 3, num_voxels_base=args.world ** 3, alpha_init=1e-2,
                fast_color_thres=1e-4, rgbnet_dim=12, rgbnet_width=128, rgbnet_direct=True).to(dev)
with torch.no_grad():
    m.density.copy_(sc['density']); m.k0.copy_(sc['k0']); m.mask_cache.mask.copy_(sc['mask'])
rk = dict(near=sc['near'], far=sc['far'], bg=1, stepsize=0.5)
NAMES = ['dvgo_sample_pts_prepare', 'dvgo_march_density', 'dvgo_exclusive_scan_i32', 'dvgo_march_gather',
         'dvgo_march_composite', 'dvgo_march_composite_bwd', 'dvgo_march_feat_bwd', 'dvgo_march_density_bwd', 'dvgo_grid_grad_split']


def one_pass():
    m.zero_grad(set_to_none=True)
    res = m(sc['rays_o'], sc['rays_d'], sc['viewdirs'], **rk)
    loss = (res['rgb_marched'] - sc['target']).pow(2).mean() + 1e-3 * res['alphainv_last'].mean()
    loss.backward()
    return res


def run(tuning, rounds):
    from directvoxgo_amd import fused
    for k, v in tuning.items():
        if k == 'combined':
            fused.COMBINED_GRID_GRAD = v
        else:
            L.call('dvgo_set_tuning', ctypes.c_int(k), ctypes.c_int(v))
    one_pass()
    torch.cuda.synchronize()
    per = {n: [] for n in NAMES}
    for _ in range(rounds):
        L.profile_start(NAMES)
        one_pass()
        for n, (cnt, ms) in L.profile_stop().items():
            if cnt:
                per[n].append(ms / cnt * 1e3)
    return per, m.k0.grad.clone(), m.density.grad.clone()


variants = {'base(0,0)': {0: 0, 1: 0, 'combined': False}, 'dedup(1,1)': {0: 1, 1: 1, 'combined': False},
            'combined rows': {0: 1, 1: 1, 'combined': True}}
results = {}
for r in range(2):               # interleave
    for name, t in variants.items():
        per, gk, gd = run(t, args.rounds)
        results.setdefault(name, []).append((per, gk, gd))
M3 = one_pass()['weights'].numel()
print('M3 =', M3)
for name, runs in results.items():
    print('==', name)
    for n in NAMES:
        xs = [x for per, _, _ in runs for x in per[n]]
        if xs:
            print(f'  {n:28s} avg {sum(xs) / len(xs):9.1f} us   min {min(xs):9.1f} us')
ref_k, ref_d = results['base(0,0)'][0][1], results['base(0,0)'][0][2]
for name, runs in results.items():
    gk, gd = runs[-1][1], runs[-1][2]
    print(name, 'k0.grad max rel diff', float((gk - ref_k).abs().max() / ref_k.abs().max()),
          'density.grad max rel diff', float((gd - ref_d).abs().max() / ref_d.abs().max()))
