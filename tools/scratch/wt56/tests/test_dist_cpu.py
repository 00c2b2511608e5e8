"""CPU, world_size 2, gloo: the ray-parallel data-parallel step.

The HIP ops need a GPU, so the N > 1 *host logic* is exercised with a small pure-torch stand-in model
that returns the same result dict as DirectVoxGO.forward: rays sharded by rank, every loss term
normalised by the global ray count, grid gradients summed with all-reduce, MLP gradients in one flat
bucket, optimizer step after the reduction.  Two ranks on half the batch each must end at exactly the
parameters one process reaches on the whole batch.
"""
import os
import socket

import pytest
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F

from directvoxgo_amd.train import FINE_TRAIN, TrainStep, render_loss


class ToyModel(nn.Module):
    """Same parameter names / result dict as DirectVoxGO, pure torch, S fixed samples per ray."""

    def __init__(self, S=6):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.density = nn.Parameter(torch.randn(1, 1, 6, 5, 5, generator=g))
        self.k0 = nn.Parameter((torch.randn(1, 6, 6, 5, 5, generator=g) * 0.3).contiguous(memory_format=torch.channels_last_3d))
        self.rgbnet = nn.Sequential(nn.Linear(6, 8), nn.ReLU(), nn.Linear(8, 3))
        for p in self.rgbnet.parameters():
            p.data = torch.randn(p.shape, generator=g) * 0.3
        self.S = S

    def forward(self, rays_o, rays_d, viewdirs, global_step=None, bg=1, **kw):
        N = rays_o.shape[0]
        t = torch.linspace(0.1, 0.9, self.S)
        pts = rays_o[:, None] + rays_d[:, None] * t[None, :, None]                  # [N,S,3] in [-1,1]
        grid = pts.reshape(1, 1, 1, -1, 3).flip(-1)
        dens = F.grid_sample(self.density, grid, align_corners=True).reshape(-1)
        feat = F.grid_sample(self.k0, grid, align_corners=True).reshape(6, -1).T
        alpha = (1 - torch.exp(-F.softplus(dens))).reshape(N, self.S)
        T = torch.cumprod(torch.cat([torch.ones(N, 1), 1 - alpha + 1e-10], 1), 1)
        weights = (T[:, :-1] * alpha).reshape(-1)
        rgb = torch.sigmoid(self.rgbnet(feat))
        ray_id = torch.arange(N).repeat_interleave(self.S)
        marched = torch.zeros(N, 3).index_add(0, ray_id, weights[:, None] * rgb) + T[:, -1:] * bg
        return {'alphainv_last': T[:, -1], 'weights': weights, 'rgb_marched': marched, 'raw_alpha': alpha.reshape(-1),
                'raw_rgb': rgb, 'ray_id': ray_id}


def make_batch(n):
    g = torch.Generator().manual_seed(11)
    ro = torch.rand(n, 3, generator=g) * 0.4 - 0.2
    rd = torch.rand(n, 3, generator=g) * 1.2 - 0.6
    return ro, rd, rd / rd.norm(dim=-1, keepdim=True), torch.rand(n, 3, generator=g)


class TorchMaskedAdam(torch.optim.Optimizer):
    """MaskedAdam's interface (param-group key `skip_zero_grad`, `step`, `step_shard`) on torch CPU ops
    (oracle/torch_cpu.adam_step = lib/masked_adam.py:39-71 over adam_upd_kernel.cu:8-58): the stand-in that lets the
    sharded data-parallel update of TrainStep run without a GPU."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.99), eps=1e-8):
        super().__init__(params, {'lr': lr, 'betas': betas, 'eps': eps})

    def _st(self, p):
        st = self.state[p]
        if not st:
            st.update(step=0, exp_avg=torch.zeros_like(p, memory_format=torch.preserve_format),
                      exp_avg_sq=torch.zeros_like(p, memory_format=torch.preserve_format))
        return st

    def _group_of(self, p):
        return next(g for g in self.param_groups if any(q is p for q in g['params']))

    @torch.no_grad()
    def step(self):
        from oracle.torch_cpu import adam_step
        for g in self.param_groups:
            for p in g['params']:
                if p.grad is None:
                    continue
                st = self._st(p)
                st['step'] += 1
                adam_step(p.data, p.grad, st['exp_avg'], st['exp_avg_sq'], st['step'], g['lr'], mode=1 if g.get('skip_zero_grad') else 0)

    @torch.no_grad()
    def step_shard(self, p, flat_p, flat_g, lo, hi):
        from directvoxgo_amd.train import flat_view
        from oracle.torch_cpu import adam_step
        g, st = self._group_of(p), self._st(p)
        st['step'] += 1
        adam_step(flat_p[lo:hi], flat_g[lo:hi], flat_view(st['exp_avg'])[lo:hi], flat_view(st['exp_avg_sq'])[lo:hi], st['step'],
                  g['lr'], mode=1 if g.get('skip_zero_grad') else 0)


def run_steps(model, batch, rank, world, n_steps=3, mode='dense', out=None):
    cfg = dict(FINE_TRAIN, weight_entropy_last=0.01, weight_rgbper=0.05)
    if mode in ('sharded', 'sharded_off'):
        # Adam with the masked rule on the grids, the sharded update on (reduce-scatter -> slab Adam -> all-gather)
        # or off (all-reduce -> full Adam on every rank)
        opt = TorchMaskedAdam([{'params': [model.density], 'lr': 0.1, 'skip_zero_grad': True},
                               {'params': [model.k0], 'lr': 0.1, 'skip_zero_grad': True},
                               {'params': list(model.rgbnet.parameters()), 'lr': 1e-2}])
        step = TrainStep(model, cfg, dict(bg=1), optimizer=opt, touched_reduce=False, shard_grids=(mode == 'sharded'))
        step.sharded_steps = 0
        orig = step._sharded_update
        def counted(shards):
            step.sharded_steps += 1
            return orig(shards)
        step._sharded_update = counted
    else:
        opt = torch.optim.SGD(model.parameters(), lr=0.5)
        step = TrainStep(model, cfg, dict(bg=1), optimizer=opt, touched_reduce=(mode != 'dense'))
    if mode == 'touched':
        step.TOUCHED_MAX = 2.0            # always take the compact (touched-voxel) reduction
    elif mode == 'adaptive':
        step.TOUCHED_MAX = 0.0            # probe, find the union too large, fall back to the dense all-reduce
    n = batch[0].shape[0] // world
    shard = tuple(t[rank * n:(rank + 1) * n] for t in batch)
    losses = []
    for s in range(n_steps):
        losses.append(step(*shard, global_step=s))
    if mode == 'sharded' and world > 1:
        assert step.sharded_steps == n_steps                    # the slab path really ran
    if mode == 'sharded_off':
        assert step.sharded_steps == 0
    if out is not None:
        out['step'] = step
    return torch.stack(losses)


def _moments(model, step):
    return {f'{name}.{key}': step.optimizer.state[p][key].detach().numpy().copy()
            for name, p in (('density', model.density), ('k0', model.k0)) for key in ('exp_avg', 'exp_avg_sq')}


def _worker(rank, world, port, q, mode='dense'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = ToyModel()
    out = {}
    losses = run_steps(model, make_batch(32), rank, world, mode=mode, out=out)
    dist.all_reduce(losses)           # per-rank shares of the global loss add up to it
    moments = None
    if mode == 'sharded':             # what a checkpoint of a data-parallel run needs: every rank's slab of the moments
        assert out['step'].gather_optimizer_state()
        moments = _moments(model, out['step'])
    if rank == 0:
        # numpy: pickled by value (torch tensors would travel as shared-memory handles of a process about to exit)
        q.put(({k: v.detach().numpy().copy() for k, v in model.state_dict().items()}, losses.numpy().copy(), moments))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(120)
@pytest.mark.parametrize('mode', ['dense', 'touched', 'adaptive', 'sharded', 'sharded_off'])
def test_two_ranks_equal_one_process(mode):
    """`touched`: the grid gradients travel as the compacted union of the voxels either rank touched.
    `sharded`: reduce-scatter of the grid gradients, Adam on the owned X-slab only, all-gather of the parameters."""
    ref_model = ToyModel()
    ref_out = {}
    ref_losses = run_steps(ref_model, make_batch(32), 0, 1, mode=mode if mode.startswith('sharded') else 'dense', out=ref_out)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    sd, losses, moments = q.get(timeout=100)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.allclose(torch.from_numpy(losses), ref_losses, rtol=1e-5, atol=1e-7)
    for k, v in ref_model.state_dict().items():
        assert torch.allclose(torch.from_numpy(sd[k]), v, rtol=1e-5, atol=1e-6), k
    if mode == 'sharded':             # gathered moments == the single process's: the checkpoint of the run is complete
        for k, v in _moments(ref_model, ref_out['step']).items():
            assert np.allclose(moments[k], v, rtol=1e-5, atol=1e-8), k
            assert np.abs(moments[k]).sum() > 0


def test_flat_view_of_channels_last_grid_is_a_view():
    from directvoxgo_amd.train import flat_view
    g = torch.randn(1, 12, 5, 6, 7).contiguous(memory_format=torch.channels_last_3d)
    v = flat_view(g)
    assert v is not None and v.is_contiguous() and v.data_ptr() == g.data_ptr() and v.numel() == g.numel()
    v.mul_(2)                                           # writes through to the grid
    assert torch.equal(g.permute(0, 2, 3, 4, 1).reshape(-1), v)
    assert flat_view(torch.randn(1, 1, 4, 4, 4)).numel() == 64
    assert flat_view(torch.randn(4, 6)[:, ::2]) is None


def test_render_loss_equals_reference_formula_on_one_rank():
    """run.py:377-386 written with means == the sum / global-count form used for DP."""
    m = ToyModel()
    ro, rd, vd, tgt = make_batch(16)
    res = m(ro, rd, vd)
    cfg = dict(FINE_TRAIN)
    a = render_loss(res, tgt, 16, cfg)
    b = cfg['weight_main'] * F.mse_loss(res['rgb_marched'], tgt)
    pout = res['alphainv_last'].clamp(1e-6, 1 - 1e-6)
    b = b + cfg['weight_entropy_last'] * (-(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout)).mean())
    rgbper = (res['raw_rgb'] - tgt[res['ray_id']]).pow(2).sum(-1)
    b = b + cfg['weight_rgbper'] * ((rgbper * res['weights'].detach()).sum() / 16)
    assert torch.allclose(a, b, rtol=1e-6)
