"""Where does the HOST time of a sparse (launch-bound) training step go?  cProfile over N steps of the lego-like scene.
    python tools/host_profile.py [--steps 200]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from directvoxgo_amd.dvgo import DirectVoxGO
from directvoxgo_amd.scenes import synthetic_scene
from directvoxgo_amd.train import FINE_TRAIN, TrainStep

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=200)
args = ap.parse_args()
sc = synthetic_scene(world=160, n_rays=8192, seed=777, device='cuda')
torch.manual_seed(777)
m = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=160 ** 3, num_voxels_base=160 ** 3, alpha_init=1e-2,
                fast_color_thres=1e-4, rgbnet_dim=12, rgbnet_width=128, rgbnet_direct=True).cuda()
with torch.no_grad():
    m.density.copy_(sc['density']); m.k0.copy_(sc['k0']); m.mask_cache.mask.copy_(sc['mask'])
step = TrainStep(m, dict(FINE_TRAIN), dict(near=sc['near'], far=sc['far'], bg=1, stepsize=0.5))
b = (sc['rays_o'], sc['rays_d'], sc['viewdirs'], sc['target'])
for i in range(20):
    step(*b, global_step=5000 + i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(args.steps):
    step(*b, global_step=5100 + i)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'host time per step {t_host / args.steps * 1e3:.3f} ms, wall per step {t_all / args.steps * 1e3:.3f} ms')
pr = cProfile.Profile()
pr.enable()
for i in range(args.steps):
    step(*b, global_step=6000 + i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
