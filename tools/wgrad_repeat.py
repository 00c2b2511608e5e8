"""Run-to-run spread of the colour head's parameter gradients on fixed inputs, per dvgo_shade_variant (the reduction of the
per-workgroup partial sums ends in float atomics, so the last bits may differ; anything larger is a race).
    python tools/wgrad_repeat.py [--variants 3,67] [--repeats 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from directvoxgo_amd import _lib as L
from directvoxgo_amd.dvgo import make_rgbnet
from directvoxgo_amd.shade import shade

ap = argparse.ArgumentParser()
ap.add_argument('--M', type=int, default=1000003)
ap.add_argument('--repeats', type=int, default=20)
ap.add_argument('--variants', default='3,67')
args = ap.parse_args()
torch.manual_seed(0)
M, N = args.M, 4096
net = make_rgbnet(39, 128, 3).cuda()
feat = torch.randn(M, 12, device='cuda', requires_grad=True)
emb = torch.randn(N, 27, device='cuda')
ray_id = torch.sort(torch.randint(N, (M,), device='cuda'))[0]
go = torch.randn(M, 3, device='cuda')
for v in [int(x) for x in args.variants.split(',')]:
    prev = L.lib().dvgo_shade_variant(v)
    ref, worst = None, 0.0
    for it in range(args.repeats):
        junk = torch.full((1 << 22,), float('nan'), device='cuda')
        del junk
        net.zero_grad(set_to_none=True)
        shade(net, feat, emb, ray_id, False).backward(go)
        g = [p.grad.clone() for p in net.parameters()]
        if ref is None:
            ref = g
        else:
            worst = max(worst, max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(g, ref)))
    L.lib().dvgo_shade_variant(prev)
    print(f'variant {v}: max |difference| / max |gradient| over {args.repeats - 1} repeats = {worst:.3e}')
