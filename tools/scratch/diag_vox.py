import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_brick as T
from directvoxgo_amd.train import FINE_TRAIN, TrainStep
def run():
    sc, m = T._model(160, 8192, width=128, direct=True, scene='roofline')
    p0 = m.k0.detach().clone()
    step = TrainStep(m, dict(FINE_TRAIN), dict(near=sc['near'], far=sc['far'], bg=1, stepsize=sc['stepsize']), rows_adam=True)
    step(sc['rays_o'], sc['rays_d'], sc['viewdirs'], sc['target'], global_step=5000)
    torch.cuda.synchronize()
    st = step.optimizer.state[m.k0]
    return m.k0.detach().clone(), st['exp_avg'].clone(), st['exp_avg_sq'].clone(), p0
a, b = run(), run()
d = (a[0] - b[0]).abs()
idx = (d > 1e-6).nonzero()
print(idx.shape[0], 'elements differ')
torch.set_printoptions(precision=10, linewidth=200)
for r in idx[:: max(1, idx.shape[0] // 4)][:4].tolist():
    _, c, x, y, z = r
    print('voxel', (x, y, z), 'channel', c)
    for name, k in (('p0', 3), ('p ', 0), ('m ', 1), ('v ', 2)):
        print('  ', name, 'A', a[k][0, :, x, y, z].tolist())
        print('  ', name, 'B', b[k][0, :, x, y, z].tolist())
