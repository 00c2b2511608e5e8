"""GPU A/B of the forward march kernels (ray_setup, march_density, march_scans, march_gather, march_composite) on the
roofline case (160^3, 8192 x 256) and the lego-like scene, variants interleaved in one process; every variant's
outputs are compared with the first one's by torch.equal.
    python tools/fwd_ab.py [--world 160] [--rays 8192] [--rounds 10] [--train]"""
import argparse
import ctypes
import itertools
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
ap.add_argument('--train', action='store_true', help='training forward (brick counts, grids require grad)')
ap.add_argument('--density', default='0,1')
ap.add_argument('--gather', default='0,4,8')
ap.add_argument('--workloads', default='roofline,lego')
ap.add_argument('--same-ray', action='store_true', help='every ray = ray 0 (all waves touch the same voxels: cache hits, same address pattern per instruction)')
ap.add_argument('--experiment', default='0', help='comma list of DVGO_TUNE_EXPERIMENT masks (non-zero: timing only, no output check)')
args = ap.parse_args()
NAMES = ['dvgo_sample_pts_prepare', 'dvgo_march_density', 'dvgo_march_scans', 'dvgo_exclusive_scan_i32', 'dvgo_march_gather',
         'dvgo_march_composite']


def tune(k, v):
    L.call('dvgo_set_tuning', ctypes.c_int(k), ctypes.c_int(v))


for workload in args.workloads.split(','):
    sc = roofline_scene(world=args.world, n_rays=args.rays, device='cuda') if workload == 'roofline' else \
        synthetic_scene(world=args.world, n_rays=args.rays, device='cuda')
    m = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=args.world ** 3, num_voxels_base=args.world ** 3, alpha_init=1e-2,
                    fast_color_thres=1e-4, rgbnet_dim=12, rgbnet_width=128, rgbnet_direct=True).cuda()
    with torch.no_grad():
        m.density.copy_(sc['density']); m.k0.copy_(sc['k0']); m.mask_cache.mask.copy_(sc['mask'])
    rk = dict(near=sc['near'], far=sc['far'], bg=1, stepsize=0.5)
    if args.same_ray:
        for k in ('rays_o', 'rays_d', 'viewdirs'):
            sc[k] = sc[k][:1].expand_as(sc[k]).contiguous()

    def fwd():
        if args.train:
            return m(sc['rays_o'], sc['rays_d'], sc['viewdirs'], **rk)
        with torch.no_grad():
            return m(sc['rays_o'], sc['rays_d'], sc['viewdirs'], **rk)

    variants = list(itertools.product([int(x) for x in args.density.split(',')], [int(x) for x in args.gather.split(',')], [int(x) for x in args.experiment.split(',')]))
    ref, times = None, {v: {n: [] for n in NAMES} for v in variants}
    for rep in range(2):
        for v in variants:
            tune(2, v[0]); tune(3, v[1]); tune(4, v[2])
            res = fwd()
            torch.cuda.synchronize()
            keys = ('weights', 'raw_alpha', 'alphainv_last', 'ray_id', 'rgb_marched')
            if v[2]:
                pass
            elif ref is None:
                ref = {k: res[k].detach().clone() for k in keys}
            else:
                for k in keys:
                    assert torch.equal(ref[k], res[k].detach()), (workload, v, k)
            for _ in range(args.rounds):
                L.profile_start(NAMES)
                fwd()
                for n, (cnt, ms) in L.profile_stop().items():
                    if cnt:
                        times[v][n].append(ms / cnt * 1e3)
    print(f'== {workload}  {"train" if args.train else "inference"} forward   M3 = {ref["weights"].numel()}   (us: avg / min)')
    for v in variants:
        tot = 0.0
        line = f'  density={v[0]} gather={v[1]} exp={v[2]}: '
        for n in NAMES:
            xs = times[v][n]
            if xs:
                line += f'{n[5:]} {sum(xs) / len(xs):6.1f}/{min(xs):6.1f}  '
                tot += sum(xs) / len(xs)
        print(line + f' | sum {tot:6.1f}')
    tune(2, 1); tune(3, 1); tune(4, 0)
