import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_brick as T
from directvoxgo_amd.train import FINE_TRAIN, TrainStep
outs = []
for fused_adam in (True, False):
    sc, m = T._model(160, 8192, width=128, direct=True, scene='roofline')
    step = TrainStep(m, dict(FINE_TRAIN), dict(near=sc['near'], far=sc['far'], bg=1, stepsize=sc['stepsize']), rows_adam=fused_adam)
    for it in range(3):
        step(sc['rays_o'], sc['rays_d'], sc['viewdirs'], sc['target'], global_step=5000 + it)
    torch.cuda.synchronize()
    outs.append((m.density.detach().clone(), m.k0.detach().clone()))
for name, a, b in (('density', outs[0][0], outs[1][0]), ('k0', outs[0][1], outs[1][1])):
    d = (a - b).abs().flatten()
    print(name, 'n', d.numel(), 'max', float(d.max()), ' >2e-3:', int((d > 2e-3).sum()), ' >1e-3:', int((d > 1e-3).sum()), ' >1e-4:', int((d > 1e-4).sum()), ' >1e-5:', int((d > 1e-5).sum()))
