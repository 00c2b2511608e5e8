import sys, os, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_brick as T
from directvoxgo_amd.train import FINE_TRAIN, TrainStep
def run(overlap):
    sc, m = T._model(160, 8192, width=128, direct=True, scene='roofline')
    step = TrainStep(m, dict(FINE_TRAIN), dict(near=sc['near'], far=sc['far'], bg=1, stepsize=sc['stepsize']), rows_adam=False, overlap_wgrad=overlap)
    # one backward without the optimizer: TrainStep steps, so read the gradient through a hook on the optimizer? simpler: the
    # parameters after one dense step carry it; compare the Adam moments' first step (m = (1 - b1) g): exp_avg / 0.1 = g
    step(sc['rays_o'], sc['rays_d'], sc['viewdirs'], sc['target'], global_step=5000)
    torch.cuda.synchronize()
    st = step.optimizer.state[m.k0]
    return (st['exp_avg'] / 0.1).clone()
ref = run(False)
print('reference gradient: max |g|', float(ref.abs().max()), ' nonzero', int((ref != 0).sum()))
for i in range(4):
    g = run(True)
    d = (g - ref).abs()
    idx = (d > 1e-7 * float(ref.abs().max())).nonzero()
    print(f'overlapped run {i}: {idx.shape[0]} elements differ; max abs diff {float(d.max()):.3e}; max rel to |g| there: ', end='')
    if idx.shape[0]:
        sel = d > 1e-7 * float(ref.abs().max())
        print(f'{float((d[sel] / ref[sel].abs().clamp_min(1e-30)).max()):.3e}', ' channels', torch.unique(idx[:, 1]).tolist())
        k = idx[:5]
        for r in k.tolist():
            print('     at', r, 'ref', float(ref[tuple(r)]), 'got', float(g[tuple(r)]))
    else:
        print('-')
