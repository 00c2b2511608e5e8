import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_brick as T
from directvoxgo_amd.train import FINE_TRAIN, TrainStep
import os
OVERLAP = os.environ.get('OVERLAP', '1') == '1'
def run(fused_adam, steps=3):
    sc, m = T._model(160, 8192, width=128, direct=True, scene='roofline')
    step = TrainStep(m, dict(FINE_TRAIN), dict(near=sc['near'], far=sc['far'], bg=1, stepsize=sc['stepsize']), rows_adam=fused_adam, overlap_wgrad=OVERLAP)
    for it in range(steps):
        step(sc['rays_o'], sc['rays_d'], sc['viewdirs'], sc['target'], global_step=5000 + it)
    torch.cuda.synchronize()
    return m.k0.detach().clone(), [p.detach().clone() for p in m.rgbnet.parameters()]
def cmp(tag, a, b):
    d = (a[0] - b[0]).abs().flatten()
    w = max(float((x - y).abs().max()) for x, y in zip(a[1], b[1]))
    print(tag, 'k0 max', float(d.max()), '>1e-5:', int((d > 1e-5).sum()), ' rgbnet max diff', w)
for steps in (1,):
    a1, a2, b1 = run(True, steps), run(True, steps), run(False, steps)
    cmp(f'steps {steps} fused vs fused', a1, a2)
    cmp(f'steps {steps} fused vs dense', a1, b1)
