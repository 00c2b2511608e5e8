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
    with torch.cuda.device_of(param):
        L.call('dvgo_adam_upd', ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), ptr(perlr if mode == 2 else None),
               _i64(n), _flt(adam_step_size(lr, beta1, beta2, step)), _flt(beta1), _flt(beta2), _flt(eps), _int(mode),
               stream_of(param))


class MaskedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.99), eps=1e-8):
        if not 0.0 <= lr:
            raise ValueError('Invalid learning rate: {}'.format(lr))
        if not 0.0 <= eps:
            raise ValueError('Invalid epsilon value: {}'.format(eps))
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError('Invalid beta parameters: {}'.format(betas))
        self.per_lr = None
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    def set_pervoxel_lr(self, count):
        assert self.param_groups[0]['params'][0].shape == count.shape
        self.per_lr = count.float() / count.max()

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            lr, (beta1, beta2), eps = group['lr'], group['betas'], group['eps']
            skip_zero_grad = group.get('skip_zero_grad', False)
            for param in group['params']:
                if param.grad is None:
                    continue
                state = self.state[param]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(param, memory_format=torch.preserve_format)
                    state['exp_avg_sq'] = torch.zeros_like(param, memory_format=torch.preserve_format)
                state['step'] += 1
                grad = param.grad
                if grad.stride() != param.stride():
                    grad = torch.empty_like(param, memory_format=torch.preserve_format).copy_(grad)
                if self.per_lr is not None and param.shape == self.per_lr.shape:
                    per_lr = self.per_lr
                    if per_lr.stride() != param.stride():
                        per_lr = torch.empty_like(param, memory_format=torch.preserve_format).copy_(per_lr)
                        self.per_lr = per_lr
                    adam_upd(param, grad, state['exp_avg'], state['exp_avg_sq'], state['step'], beta1, beta2, lr,
                             eps, mode=2, perlr=per_lr)
                elif skip_zero_grad:
                    adam_upd(param, grad, state['exp_avg'], state['exp_avg_sq'], state['step'], beta1, beta2, lr,
                             eps, mode=1)
                else:
                    adam_upd(param, grad, state['exp_avg'], state['exp_avg_sq'], state['step'], beta1, beta2, lr,
                             eps, mode=0)
