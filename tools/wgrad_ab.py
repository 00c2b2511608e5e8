"""A/B of the colour head's three kernels per dvgo_shade_variant (bit 0 forward, bit 1 data gradients, bit 2 weight gradients on
the bf16 matrix cores), HIP-event time per kernel, variants interleaved.   python tools/wgrad_ab.py [--M 2097152] [--variants 3,7]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from directvoxgo_amd import _lib as L
from directvoxgo_amd.dvgo import make_rgbnet
from directvoxgo_amd.shade import shade

ap = argparse.ArgumentParser()
ap.add_argument('--M', type=int, default=2097152)
ap.add_argument('--rounds', type=int, default=8)
ap.add_argument('--variants', default='3,7')
ap.add_argument('--width', type=int, default=128)
ap.add_argument('--experiment', type=int, default=0, help='dvgo_shade_experiment flags (switch-off experiments; results are wrong)')
ap.add_argument('--parts', type=int, default=0, help='override shade.N_PARTS (workgroups of the weight-gradient kernels)')
args = ap.parse_args()
if args.parts:
    import directvoxgo_amd.shade as _sh
    _sh.N_PARTS = args.parts
torch.manual_seed(0)
if args.experiment:
    L.lib().dvgo_shade_experiment(args.experiment)
M, N = args.M, 8192
net = make_rgbnet(39, args.width, 3).cuda()
feat = torch.randn(M, 12, device='cuda', requires_grad=True)
emb = torch.randn(N, 27, device='cuda')
ray_id = torch.arange(M, device='cuda') // max(M // N, 1)
go = torch.randn(M, 3, device='cuda')
NAMES = ['dvgo_shade_fwd', 'dvgo_shade_bwd', 'dvgo_shade_wgrad']
res, grads = {}, {}
for rep in range(2):
    for v in [int(x) for x in args.variants.split(',')]:
        L.lib().dvgo_shade_variant(v)
        for _ in range(2):
            net.zero_grad(set_to_none=True)
            shade(net, feat, emb, ray_id, False).backward(go)
        torch.cuda.synchronize()
        grads[v] = [p.grad.clone() for p in net.parameters()]
        for _ in range(args.rounds):
            net.zero_grad(set_to_none=True)
            L.profile_start(NAMES)
            shade(net, feat, emb, ray_id, False).backward(go)
            for n, (c, ms) in L.profile_stop().items():
                res.setdefault((v, n), []).append(ms / max(c, 1) * 1e3)
L.lib().dvgo_shade_variant(67)
for (v, n), xs in sorted(res.items()):
    print(f'variant {v}  {n:18s} avg {sum(xs) / len(xs):8.1f} us   min {min(xs):8.1f} us')
vs = sorted(grads)
for v in vs[1:]:
    worst = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-20)) for a, b in zip(grads[v], grads[vs[0]]))
    print(f'variant {v} vs {vs[0]}: max relative parameter-gradient difference {worst:.2e}')
