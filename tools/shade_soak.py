"""Soak test of the colour head's repeatability: N repeats of forward + data-gradient backward on the same inputs must be
bitwise identical (no atomics on that path).   python tools/shade_soak.py [--repeats 1000]
("Tried and withdrawn", for why this exists.)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from directvoxgo_amd.dvgo import make_rgbnet
from directvoxgo_amd.shade import shade

ap = argparse.ArgumentParser()
ap.add_argument('--repeats', type=int, default=1000)
args = ap.parse_args()
for (width, C, E, diffuse, M) in ((64, 9, 3, True, 400000), (128, 12, 27, False, 400000), (128, 12, 27, True, 2097152)):
    torch.manual_seed(1)
    d_in = (C - 3 if diffuse else C) + E
    net = make_rgbnet(d_in, width, 3).cuda()
    feat = torch.randn(M, C, device='cuda', requires_grad=True)
    emb = torch.randn(4096, E, device='cuda')
    ray_id = torch.sort(torch.randint(4096, (M,), device='cuda'))[0]
    go = torch.randn(M, 3, device='cuda')
    ref, bad = None, 0
    n = args.repeats if M < 1000000 else max(args.repeats // 10, 10)
    for it in range(n):
        rgb = shade(net, feat, emb, ray_id, diffuse)
        g = torch.autograd.grad(rgb, feat, go)[0]
        if ref is None:
            ref = (rgb.detach().clone(), g.clone())
        elif not (torch.equal(ref[0], rgb) and torch.equal(ref[1], g)):
            bad += 1
    print(f'width {width} C {C} diffuse {diffuse} M {M}: {bad} of {n - 1} repeats differ')
