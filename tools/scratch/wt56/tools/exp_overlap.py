"""Do the MFMA-bound colour-head kernels and the atomic-bound gradient scatter overlap when launched on two streams?
(register file: shade_bwd 2 x 169, shade_fwd 2 x 197, shade_wgrad 2 x 240 of 512 per SIMD; feat_bwd 88)
    python tools/exp_overlap.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from directvoxgo_amd import _lib as L
from directvoxgo_amd._lib import _flt, _i64, _int, ptr
from directvoxgo_amd.dvgo import DirectVoxGO, make_rgbnet
from directvoxgo_amd.fused import fused_march
from directvoxgo_amd.scenes import roofline_scene

dev = 'cuda'
sc = roofline_scene(world=160, n_rays=8192, device=dev)
m = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=160 ** 3, num_voxels_base=160 ** 3, alpha_init=1e-2, fast_color_thres=1e-4,
                rgbnet_dim=12, rgbnet_width=128, rgbnet_direct=True).to(dev)
with torch.no_grad():
    m.density.copy_(sc['density']); m.k0.copy_(sc['k0']); m.mask_cache.mask.copy_(sc['mask'])
cfg = m._march_cfg(sc['near'], sc['far'], 0.5)
with torch.no_grad():
    w, a, last, feat, ray_id, step_id, off3 = fused_march(m.density, m.k0, sc['rays_o'], sc['rays_d'], cfg)
M = ray_id.shape[0]
print('M3', M)
net = make_rgbnet(39, 128, 3).to(dev)
W1, W2, W3 = net[0].weight.contiguous(), net[2][0].weight.contiguous(), net[3].weight.contiguous()
g_rgb = torch.randn(M, 3, device=dev); rgb = torch.rand(M, 3, device=dev)
masks = torch.randint(-2 ** 62, 2 ** 62, (M, 4), device=dev, dtype=torch.int64)
g_feat = torch.empty(M, 12, device=dev); G1 = torch.empty(M, 128, device=dev); gz = torch.empty(M, 3, device=dev)
grad_feat = torch.randn(M, 12, device=dev); kept = torch.randn(M, device=dev)
G = torch.zeros(160 ** 3, 16, device=dev)
# the march kernels want the ray starts / dirs the forward derived
N = 8192
t_min = torch.empty(N, device=dev); t_max = torch.empty(N, device=dev); n_steps = torch.empty(N, dtype=torch.int64, device=dev)
start = torch.empty(N, 3, device=dev); dirs = torch.empty(N, 3, device=dev)
s0 = torch.cuda.current_stream()
L.call('dvgo_sample_pts_prepare', ptr(sc['rays_o']), ptr(sc['rays_d']), ptr(cfg.xyz_min_t), ptr(cfg.xyz_max_t), _flt(cfg.near),
       _flt(cfg.far), _flt(cfg.stepdist), _i64(N), ptr(t_min), ptr(t_max), ptr(n_steps), ptr(None), ptr(start), ptr(dirs),
       ctypes.c_void_p(s0.cuda_stream))


def bwd(stream):
    L.call('dvgo_shade_bwd', ptr(g_rgb), ptr(rgb), ptr(masks), _i64(M), ptr(None), ptr(W1), ptr(W2), ptr(W3), _int(128), _int(39), _int(12),
           _int(0), ptr(g_feat), ptr(G1), ptr(gz), ctypes.c_void_p(stream.cuda_stream))


def scatter(stream):
    L.call('dvgo_march_feat_bwd', ptr(grad_feat), ptr(kept), ptr(ray_id), ptr(step_id), _i64(M), ptr(start), ptr(dirs),
           _flt(cfg.stepdist), cfg.xyz_min_h, cfg.xyz_max_h, _int(12), _int(160), _int(160), _int(160), _i64(1), _i64(160 * 160 * 16),
           _i64(160 * 16), _i64(16), ptr(G), ctypes.c_void_p(stream.cuda_stream))


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def both(first_scatter):
    s1.wait_stream(s0); s2.wait_stream(s0)
    if first_scatter:
        scatter(s2); bwd(s1)
    else:
        bwd(s1); scatter(s2)
    s0.wait_stream(s1); s0.wait_stream(s2)


print('shade_bwd alone   %.3f ms' % timed(lambda: bwd(s0)))
print('feat_bwd alone    %.3f ms' % timed(lambda: scatter(s0)))
print('both, bwd first   %.3f ms' % timed(lambda: both(False)))
print('both, scatter 1st %.3f ms' % timed(lambda: both(True)))
