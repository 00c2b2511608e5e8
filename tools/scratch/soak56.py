"""Round-2 commit 56b0712 (data-gradient prefetch in shade_bwd_x3: the build whose results differed from run to run), built
twice -- as then, and with -fno-slp-vectorize on shade_x3.hip (no v_pk_*_f32) -- and run through the bitwise-repeatability
check of tests/test_gpu_ops.py.      python tools/scratch/soak56.py <libdir>"""
import os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'wt56')
sys.path.insert(0, root)
import torch
from directvoxgo_amd.dvgo import make_rgbnet
from directvoxgo_amd.shade import shade
torch.manual_seed(1)
M = 400000
for width, C, E, diffuse in ((64, 9, 3, True), (128, 12, 27, False)):
    d_in = (C - 3 if diffuse else C) + E
    net = make_rgbnet(d_in, width, 3).cuda()
    feat = torch.randn(M, C, device='cuda', requires_grad=True)
    emb = torch.randn(4096, E, device='cuda')
    ray_id = torch.sort(torch.randint(4096, (M,), device='cuda'))[0]
    go = torch.randn(M, 3, device='cuda')
    ref, bad_rows, bad_reps = None, 0, 0
    for it in range(30):
        rgb = shade(net, feat, emb, ray_id, diffuse)
        g_feat = torch.autograd.grad(rgb, feat, go)[0]
        if ref is None:
            ref = g_feat.clone()
        else:
            bad = int((g_feat != ref).any(1).sum())
            bad_rows += bad
            bad_reps += bad > 0
    print(f'width {width}: {bad_reps} of 29 repeats differ from the first, {bad_rows} rows in all')
