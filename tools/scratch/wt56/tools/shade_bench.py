"""GPU micro-benchmark of the fused colour head (csrc/shade.hip): inference (no activation stores) vs
training forward, against the torch modules.   python tools/shade_bench.py [--M 2097152]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from directvoxgo_amd.dvgo import make_rgbnet, mlp_forward
from directvoxgo_amd.shade import shade

ap = argparse.ArgumentParser()
ap.add_argument('--M', type=int, default=2097152)
ap.add_argument('--rounds', type=int, default=10)
ap.add_argument('--experiment', type=int, default=0)
ap.add_argument('--variant', type=int, default=-1, help='dvgo_shade_variant bits (default: leave as is)')
args = ap.parse_args()
from directvoxgo_amd import _lib as L
if args.variant >= 0:
    L.lib().dvgo_shade_variant(args.variant)
L.lib().dvgo_shade_experiment(args.experiment)
print('shade variant', L.lib().dvgo_shade_variant(-1))
torch.manual_seed(0)
M, N = args.M, 8192
net = make_rgbnet(39, 128, 3).cuda()          # rgbnet_direct head of configs/default.py: 12 + 27 inputs
feat = torch.randn(M, 12, device='cuda')
emb = torch.randn(N, 27, device='cuda')
ray_id = torch.arange(M, device='cuda') // (M // N)


def timeit(fn, rounds):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sum(ts) / len(ts), min(ts)


def torch_fwd():
    with torch.no_grad():
        x = torch.cat([feat, emb[ray_id]], -1)
        return torch.sigmoid(net(x))


def hip_infer():
    with torch.no_grad():
        return shade(net, feat, emb, ray_id, False)


fg = feat.clone().requires_grad_()


def hip_train_fwd():
    return shade(net, fg, emb, ray_id, False)


def hip_train_fwd_bwd():
    r = shade(net, fg, emb, ray_id, False)
    r.sum().backward()


def torch_train_fwd_bwd():
    x = torch.cat([fg, emb[ray_id]], -1)
    r = torch.sigmoid(mlp_forward(net, x))
    r.sum().backward()


flop = 2.0 * M * (36 * 128 + 128 * 128 + 128 * 3)
for name, fn in [('torch fwd (no_grad)', torch_fwd), ('hip shade infer', hip_infer), ('hip shade train fwd', hip_train_fwd),
                 ('hip shade fwd+bwd', hip_train_fwd_bwd), ('torch fwd+bwd (split-K)', torch_train_fwd_bwd)]:
    avg, mn = timeit(fn, args.rounds)
    print(f'{name:28s} avg {avg:8.3f} ms  min {mn:8.3f} ms   fwd-equivalent {flop / mn / 1e9:8.1f} TFLOP/s')
