"""Runs the colour head with every register and the LDS of every CU filled with NaNs in front of each of our launches
(needs tools/debug_pollute.hip built into the library: see its header).  No NaN may come out and results may not move."""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from directvoxgo_amd import _lib as L
from directvoxgo_amd.dvgo import make_rgbnet
from directvoxgo_amd.shade import shade
if not hasattr(L.lib(), 'dvgo_debug_pollute_registers'):
    sys.exit('libdvgo_hip.so was built without tools/debug_pollute.hip (see the header of that file)')
_orig = L.call
POLLUTE = [False]
def call(name, *args):
    if POLLUTE[0]:
        L.lib().dvgo_debug_pollute_registers(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return _orig(name, *args)
L.call = call
import directvoxgo_amd.shade as S
S.L.call = call
for (width, C, E, diffuse, M) in ((64, 9, 3, True, 400000), (128, 12, 27, False, 400000), (128, 12, 27, True, 100000)):
    torch.manual_seed(1)
    d_in = (C - 3 if diffuse else C) + E
    net = make_rgbnet(d_in, width, 3).cuda()
    feat = torch.randn(M, C, device='cuda', requires_grad=True)
    emb = torch.randn(4096, E, device='cuda')
    ray_id = torch.sort(torch.randint(4096, (M,), device='cuda'))[0]
    go = torch.randn(M, 3, device='cuda')
    outs = []
    for pol in (False, True, True, False):
        POLLUTE[0] = pol
        rgb = shade(net, feat, emb, ray_id, diffuse)
        g = torch.autograd.grad(rgb, [feat] + list(net.parameters()), go)
        POLLUTE[0] = False
        outs.append([rgb.detach().clone()] + [x.clone() for x in g])
    names = ['rgb', 'g_feat', 'gW1', 'gb1', 'gW2', 'gb2', 'gW3', 'gb3']
    for i, o in enumerate(outs):
        print(width, 'run', i, 'nan counts', {n: int(torch.isnan(t).sum()) for n, t in zip(names, o) if torch.isnan(t).any()},
              'equal to run 0 (rgb, g_feat):', torch.equal(o[0], outs[0][0]), torch.equal(o[1], outs[0][1]))
