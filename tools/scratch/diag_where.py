import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_brick as T
from directvoxgo_amd.train import FINE_TRAIN, TrainStep
import os
import directvoxgo_amd.fused as F
if os.environ.get('SLICE'):
    F.BRICK_SLICE = int(os.environ['SLICE'])
def run():
    sc, m = T._model(160, 8192, width=128, direct=True, scene='roofline')
    step = TrainStep(m, dict(FINE_TRAIN), dict(near=sc['near'], far=sc['far'], bg=1, stepsize=sc['stepsize']), rows_adam=True)
    step(sc['rays_o'], sc['rays_d'], sc['viewdirs'], sc['target'], global_step=5000)
    torch.cuda.synchronize()
    return m.k0.detach().clone(), m.density.detach().clone()
runs = [run() for _ in range(4)]
print('k0 shape', tuple(runs[0][0].shape), 'strides', runs[0][0].stride())
for i in range(1, 4):
    d = (runs[i][0] - runs[0][0]).abs()
    idx = (d > 1e-6).nonzero()
    print(f'run {i} vs 0: {idx.shape[0]} k0 elements differ, max {float(d.max()):.3e}; density max {float((runs[i][1] - runs[0][1]).abs().max()):.3e}')
    if idx.shape[0]:
        vox = idx[:, 2:5] if idx.shape[1] == 5 else idx[:, -3:]
        bricks = (vox // 8)
        ub, cnt = torch.unique(bricks, dim=0, return_counts=True)
        print('   bricks touched:', ub.shape[0], ' elements per brick:', cnt.tolist()[:20])
        print('   brick coords:', ub.tolist()[:10])
        ch = idx[:, 1] if idx.shape[1] == 5 else None
        if ch is not None:
            print('   channels:', torch.unique(ch).tolist())
