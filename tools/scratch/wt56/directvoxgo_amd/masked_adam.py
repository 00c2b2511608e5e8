"""MaskedAdam on the HIP kernels ("next" row N1 of SURVEY.md section 8f).

Same optimizer contract as /root/reference/lib/masked_adam.py:17-71: Adam(betas=(0.9, 0.99),
eps=1e-8) with (a) optional per-voxel learning rate for the parameter whose shape matches
``per_lr`` and (b) ``skip_zero_grad`` groups that leave voxels with a zero gradient untouched.
Dispatch order per-lr -> masked -> plain as at :60-71.  The bias-corrected step size is
computed on the host in float32 exactly as lib/cuda/adam_upd_kernel.cu:72 does.
"""
import numpy as np
import torch

from . import _lib as L
from ._lib import _flt, _i64, _int, ptr, stream_of


def _dense_same_layout(*ts):
    s0 = ts[0].stride()
    return all(t.stride() == s0 and t.shape == ts[0].shape for t in ts)


def adam_step_size(lr, beta1, beta2, step):
    f = np.float32
    return float(f(lr) * np.sqrt(f(1) - np.power(f(beta2), f(step))) / (f(1) - np.power(f(beta1), f(step))))


def adam_upd(param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps, mode=0, perlr=None):
    """adam_upd_cuda.{adam_upd, masked_adam_upd, adam_upd_with_perlr} (lib/cuda/adam_upd.cpp:36-86):
    mode 0 / 1 / 2.  Elementwise and in place, so any memory layout works as long as all
    tensors share it."""
    ts = [param, grad, exp_avg, exp_avg_sq] + ([perlr] if mode == 2 else [])
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError('param must be a CUDA tensor')
    if not _dense_same_layout(*ts):
        raise RuntimeError('param, grad and optimizer state must share one memory layout')
    n = param.numel()
    with L.device_of(param):
        L.call('dvgo_adam_upd', ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), ptr(perlr if mode == 2 else None),
               _i64(n), _flt(adam_step_size(lr, beta1, beta2, step)), _flt(beta1), _flt(beta2), _flt(eps), _int(mode),
               stream_of(param))


def adam_upd_multi(items, step, beta1, beta2, lr, eps, step_size_dev=None):
    """Plain Adam over up to 16 small (param, state) pairs in one launch (csrc/loss.hip).  `step_size_dev`: a 1-element
    device float tensor holding the step size (captured steps), read instead of the host value."""
    import ctypes
    n = len(items)
    PT = ctypes.c_void_p * n
    ps = PT(*[p.data_ptr() for p, _ in items])
    gs = PT(*[p.grad.data_ptr() for p, _ in items])
    ms = PT(*[st['exp_avg'].data_ptr() for _, st in items])
    vs = PT(*[st['exp_avg_sq'].data_ptr() for _, st in items])
    ne = (ctypes.c_int64 * n)(*[p.numel() for p, _ in items])
    p0 = items[0][0]
    with L.device_of(p0):
        L.call('dvgo_adam_upd_multi', ps, gs, ms, vs, ne, _int(n), _flt(adam_step_size(lr, beta1, beta2, step)), _flt(beta1),
               _flt(beta2), _flt(eps), ptr(step_size_dev), stream_of(p0))


class MaskedAdam(torch.optim.Optimizer):
    """Drop-in for lib/masked_adam.py:17-71 (same constructor, `set_pervoxel_lr`, param-group key
    `skip_zero_grad`, state keys `step` / `exp_avg` / `exp_avg_sq`)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.99), eps=1e-8):
        for name, val, ok in (('learning rate', lr, lr >= 0.0), ('epsilon value', eps, eps >= 0.0),
                              ('beta parameter at index 0', betas[0], 0.0 <= betas[0] < 1.0),
                              ('beta parameter at index 1', betas[1], 0.0 <= betas[1] < 1.0)):
            if not ok:
                raise ValueError(f'Invalid {name}: {val}')
        self.per_lr = None
        # captured training steps (train.TrainStep.capture): the bias-corrected step sizes live in this device tensor
        # ([0] feature grid, [1] density grid, [2 + g] small tensors of param group g) and the kernels read them from
        # there, so that a replayed HIP graph sees the values of ITS step; `hyper_begin` fills it
        self.hyper_dev, self._hyper_pin = None, None
        super().__init__(params, {'lr': lr, 'betas': betas, 'eps': eps})

    def set_pervoxel_lr(self, count):
        """Per-voxel learning-rate multiplier = view count / max view count (run.py:311-320)."""
        first = self.param_groups[0]['params'][0]
        assert first.shape == count.shape
        self.per_lr = count.float() / count.max()

    def _state_of(self, p):
        st = self.state[p]
        if not st:
            st.update(step=0,
                      exp_avg=torch.zeros_like(p, memory_format=torch.preserve_format),
                      exp_avg_sq=torch.zeros_like(p, memory_format=torch.preserve_format))
        else:
            # state restored from a checkpoint arrives in the canonical contiguous layout (checkpoint.py, and any
            # reference-written file): the update kernels are element-wise over raw memory, so re-lay it once
            for k in ('exp_avg', 'exp_avg_sq'):
                if st[k].stride() != p.stride() or st[k].device != p.device:
                    st[k] = self._like(p, st[k].to(p.device))
        return st

    @staticmethod
    def _like(p, t):
        """`t` in the memory layout of `p` (the update kernels are element-wise over raw memory)."""
        if t.stride() == p.stride():
            return t
        return torch.empty_like(p, memory_format=torch.preserve_format).copy_(t)

    def _group_of(self, p):
        for group in self.param_groups:
            if any(q is p for q in group['params']):
                return group
        return None

    def can_step_grid_rows(self, density, k0):
        """True when `step_grid_rows` reproduces what `step` would do for these two parameters."""
        gd, gk = self._group_of(density), self._group_of(k0)
        if gd is None or gk is None or gd['betas'] != gk['betas'] or gd['eps'] != gk['eps']:
            return False
        if self.per_lr is not None and self.per_lr.shape in (density.shape, k0.shape):
            return False                                  # per-voxel learning rates go through the dense kernels
        for p in (density, k0):                     # the rows kernel walks raw memory: a resumed (contiguous) state
            if self.state.get(p):                   # is re-laid to the parameter's strides first
                self._state_of(p)
        return (k0.dim() == 5 and k0.shape[1] == 12 and k0.is_contiguous(memory_format=torch.channels_last_3d)
                and density.is_contiguous() and density.shape[2:] == k0.shape[2:])

    def can_fuse_grid_step(self, density, k0):
        """True when the brick scatter may apply this optimizer's update of the two grids itself
        (`grid_step_args`): as `can_step_grid_rows` (any built channel count), and both groups must be masked
        (`skip_zero_grad`) -- the fused update only visits the bricks a sample touched, which is the set the masked
        rule (adam_upd_kernel.cu:35) updates; plain Adam also moves voxels whose gradient is zero."""
        gd, gk = self._group_of(density), self._group_of(k0)
        if gd is None or gk is None or gd['betas'] != gk['betas'] or gd['eps'] != gk['eps']:
            return False
        if not (gd.get('skip_zero_grad', False) and gk.get('skip_zero_grad', False)):
            return False
        if self.per_lr is not None and self.per_lr.shape in (density.shape, k0.shape):
            return False
        if not (k0.dim() == 5 and k0.shape[1] > 1 and k0.is_contiguous(memory_format=torch.channels_last_3d)
                and density.is_contiguous() and density.shape[2:] == k0.shape[2:]):
            return False
        for p in (density, k0):                     # a resumed state arrives contiguous: re-lay it to the parameter's
            if self.state.get(p):                   # strides first (the kernels walk raw memory)
                self._state_of(p)
        return True

    def hyper_begin(self, density, k0, advance=False):
        """Step sizes of the NEXT optimizer step (state step + 1, current lr) -> device, ahead of the kernels that will
        read them.  `advance=True` (graph replay: the captured Python does not run) also counts the step on the host."""
        if self.hyper_dev is None:
            self.hyper_dev = torch.zeros(2 + len(self.param_groups), dtype=torch.float32, device=k0.device)
            # the host may run many replays ahead of the GPU: every upload gets its own pinned slot, recycled only after
            # the copy that read it has executed
            self._hyper_pin = [(torch.zeros(2 + len(self.param_groups), dtype=torch.float32).pin_memory(), torch.cuda.Event())
                               for _ in range(64)]
            self._hyper_next = 0
        vals, done = self._hyper_pin[self._hyper_next]
        self._hyper_next = (self._hyper_next + 1) % len(self._hyper_pin)
        done.synchronize()
        for slot, p in ((0, k0), (1, density)):
            g, st = self._group_of(p), self._state_of(p)
            vals[slot] = adam_step_size(g['lr'], g['betas'][0], g['betas'][1], st['step'] + 1)
            if advance:
                st['step'] += 1
        for gi, g in enumerate(self.param_groups):
            steps = {self._state_of(p)['step'] for p in g['params'] if p is not density and p is not k0 and p.requires_grad}
            if len(steps) > 1:
                raise RuntimeError('captured steps need one step count per param group')
            if steps:
                vals[2 + gi] = adam_step_size(g['lr'], g['betas'][0], g['betas'][1], steps.pop() + 1)
                if advance:
                    for p in g['params']:
                        if p is not density and p is not k0 and p.requires_grad:
                            self._state_of(p)['step'] += 1
        self.hyper_dev.copy_(vals, non_blocking=True)
        done.record()

    def grid_step_args(self, density, k0):
        """Counts one step for both grids and returns the Adam argument tail of dvgo_brick_accumulate
        (csrc/brick.hip): the update `step()` would make, applied by the scatter kernel from its LDS tile."""
        gd, gk = self._group_of(density), self._group_of(k0)
        sd, sk = self._state_of(density), self._state_of(k0)
        sd['step'] += 1
        sk['step'] += 1
        b1, b2 = gk['betas']
        return (ptr(k0), ptr(sk['exp_avg']), ptr(sk['exp_avg_sq']), _flt(adam_step_size(gk['lr'], b1, b2, sk['step'])),
                _int(1 if gk.get('skip_zero_grad', False) else 0),
                ptr(density), ptr(sd['exp_avg']), ptr(sd['exp_avg_sq']), _flt(adam_step_size(gd['lr'], b1, b2, sd['step'])),
                _int(1 if gd.get('skip_zero_grad', False) else 0), _flt(b1), _flt(b2), _flt(gk['eps']),
                ptr(self.hyper_dev[0:2]) if self.hyper_dev is not None else ptr(None))

    @torch.no_grad()
    def step_grid_rows(self, density, k0, G):
        """The update of `step()` for the density and feature grids, read from the combined gradient rows G
        ([n_vox, 16], fused.grid_rows_capture) in one pass (csrc/optim.hip adam_rows_kernel)."""
        gd, gk = self._group_of(density), self._group_of(k0)
        sd, sk = self._state_of(density), self._state_of(k0)
        sd['step'] += 1
        sk['step'] += 1
        b1, b2 = gk['betas']
        n_vox = density.numel()
        assert G.shape == (n_vox, 16) and G.is_contiguous()
        with L.device_of(k0):
            L.call('dvgo_adam_rows', ptr(G), _i64(n_vox), _int(16), _int(12), ptr(k0), ptr(sk['exp_avg']), ptr(sk['exp_avg_sq']),
                   _flt(adam_step_size(gk['lr'], b1, b2, sk['step'])), _int(1 if gk.get('skip_zero_grad', False) else 0),
                   ptr(density), ptr(sd['exp_avg']), ptr(sd['exp_avg_sq']),
                   _flt(adam_step_size(gd['lr'], b1, b2, sd['step'])), _int(1 if gd.get('skip_zero_grad', False) else 0),
                   _flt(b1), _flt(b2), _flt(gk['eps']), stream_of(k0))

    @torch.no_grad()
    def step_shard(self, p, flat_p, flat_g, lo, hi):
        """The update of `step()` for the elements [lo, hi) of parameter `p` in memory order (`flat_p` / `flat_g`: flat
        views of the parameter's and the gradient's memory).  Data parallel: each rank updates the slab of a grid it
        owns from the reduce-scattered gradient (train.py); the element-wise rule and the dispatch order
        (per-voxel lr -> masked -> plain, lib/masked_adam.py:60-71) are those of `step()`.  The moments stay full-size
        tensors of which a rank only ever touches its own slab."""
        group = self._group_of(p)
        st = self._state_of(p)
        st['step'] += 1
        b1, b2 = group['betas']
        use_perlr = self.per_lr is not None and p.shape == self.per_lr.shape
        if use_perlr:
            self.per_lr = self._like(p, self.per_lr)
        mode = 2 if use_perlr else (1 if group.get('skip_zero_grad', False) else 0)
        from .train import flat_view
        m, v = flat_view(st['exp_avg']), flat_view(st['exp_avg_sq'])
        pl = flat_view(self.per_lr)[lo:hi] if use_perlr else None
        adam_upd(flat_p[lo:hi], flat_g[lo:hi], m[lo:hi], v[lo:hi], st['step'], b1, b2, group['lr'], group['eps'], mode=mode,
                 perlr=pl)

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            b1, b2 = group['betas']
            masked = bool(group.get('skip_zero_grad', False))
            small = []                                                   # plain-Adam tensors batched into one launch
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self._state_of(p)
                st['step'] += 1
                use_perlr = self.per_lr is not None and p.shape == self.per_lr.shape
                if use_perlr:
                    self.per_lr = self._like(p, self.per_lr)
                mode = 2 if use_perlr else (1 if masked else 0)           # dispatch order of :60-71
                if mode == 0 and p.numel() <= 262144 and p.is_contiguous() and p.grad.is_contiguous() and p.is_cuda:
                    small.append((p, st))
                    continue
                adam_upd(p, self._like(p, p.grad), st['exp_avg'], st['exp_avg_sq'], st['step'], b1, b2, group['lr'],
                         group['eps'], mode=mode, perlr=self.per_lr if use_perlr else None)
            by_step = {}
            for p, st in small:
                by_step.setdefault(st['step'], []).append((p, st))
            gi = next(i for i, g in enumerate(self.param_groups) if g is group)
            for stp, items in by_step.items():
                for i in range(0, len(items), 16):
                    adam_upd_multi(items[i:i + 16], stp, b1, b2, group['lr'], group['eps'],
                                   self.hyper_dev[2 + gi:3 + gi] if self.hyper_dev is not None else None)
