import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from directvoxgo_amd import _lib as L
from directvoxgo_amd.dvgo import make_rgbnet
from directvoxgo_amd.shade import shade
torch.manual_seed(0)
M, N = 2097152, 8192
net = make_rgbnet(39, 128, 3).cuda()          # rgbnet_direct head of configs/default.py: 12 + 27 inputs
feat = torch.randn(M, 12, device='cuda', requires_grad=True)
emb = torch.randn(N, 27, device='cuda'); ray_id = torch.arange(M, device='cuda') // 256
go = torch.randn(M, 3, device='cuda')
NAMES = ['dvgo_shade_fwd', 'dvgo_shade_bwd', 'dvgo_shade_wgrad']
for flag in [int(a) for a in sys.argv[1:]] or (0, 1, 0, 1):
    L.lib().dvgo_shade_experiment(ctypes.c_int(flag))
    for _ in range(2):
        shade(net, feat, emb, ray_id, False).backward(go)
    torch.cuda.synchronize()
    L.profile_start(NAMES)
    for _ in range(5):
        shade(net, feat, emb, ray_id, False).backward(go)
    print(flag, {k: round(ms / c, 3) for k, (c, ms) in L.profile_stop().items()})
