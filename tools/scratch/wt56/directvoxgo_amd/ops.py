"""Autograd ops and small modules of the hot path, host side.

Mirrors the Python operator surface of the reference (same names, argument meaning and
saved-tensor conventions) on top of the HIP kernels:

  Raw2Alpha, Alphas2Weights   lib/dvgo.py:618-660
  MaskCache                   lib/dvgo.py:583-613
  grid_sample                 lib/dvgo.py:312-328 (grid_sampler -> F.grid_sample + its backward)
  segment_coo                 torch_scatter.segment_coo(src, index, out, reduce='sum')
  total_variation_add_grad    lib/cuda/total_variation.cpp:16-24
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import render_utils as render_utils_hip
from ._lib import _flt, _i64, _int, check_f32, check_input, ptr, stream_of


class Raw2Alpha(torch.autograd.Function):
    """alpha = 1 - (1 + exp(density + shift)) ** (-interval)   (lib/dvgo.py:618-642)"""

    @staticmethod
    def forward(ctx, density, shift, interval):
        exp, alpha = render_utils_hip.raw2alpha(density, shift, interval)
        if density.requires_grad:
            ctx.save_for_backward(exp)
            ctx.interval = interval
        return alpha

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_back):
        exp = ctx.saved_tensors[0]
        return render_utils_hip.raw2alpha_backward(exp, grad_back.contiguous(), ctx.interval), None, None


class Alphas2Weights(torch.autograd.Function):
    """weights_i = T_i * alpha_i with early stop, plus the residual transmittance
    (lib/dvgo.py:644-660)."""

    @staticmethod
    def forward(ctx, alpha, ray_id, N):
        weights, T, alphainv_last, i_start, i_end = render_utils_hip.alpha2weight(alpha, ray_id, N)
        if alpha.requires_grad:
            ctx.save_for_backward(alpha, weights, T, alphainv_last, i_start, i_end)
            ctx.n_rays = N
        return weights, alphainv_last

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_weights, grad_last):
        alpha, weights, T, alphainv_last, i_start, i_end = ctx.saved_tensors
        grad = render_utils_hip.alpha2weight_backward(
            alpha, weights, T, alphainv_last, i_start, i_end, ctx.n_rays,
            grad_weights.contiguous(), grad_last.contiguous())
        return grad, None, None


def _grid_geom(grid):
    """grid [1,C,X,Y,Z] (any dense strides) -> (C,X,Y,Z, sC,sX,sY,sZ) in elements."""
    if grid.dim() != 5 or grid.shape[0] != 1:
        raise RuntimeError('grid must be [1,C,X,Y,Z]')
    _, C, X, Y, Z = grid.shape
    _, sC, sX, sY, sZ = grid.stride()
    return C, X, Y, Z, sC, sX, sY, sZ


class _GridSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grid, xyz, xyz_min, xyz_max):
        if not grid.is_cuda:
            raise RuntimeError('grid must be a CUDA tensor')
        check_f32(grid, 'grid')
        check_input(xyz, 'xyz'); check_f32(xyz, 'xyz')
        C, X, Y, Z, sC, sX, sY, sZ = _grid_geom(grid)
        M = xyz.shape[0]
        out = torch.empty((M, C), dtype=torch.float32, device=xyz.device)
        with L.device_of(xyz):
            L.call('dvgo_grid_sample_fwd', ptr(grid), _int(C), _int(X), _int(Y), _int(Z), _i64(sC), _i64(sX),
                   _i64(sY), _i64(sZ), ptr(xyz), ptr(xyz_min), ptr(xyz_max), _i64(M), ptr(out), stream_of(xyz))
        ctx.save_for_backward(xyz, xyz_min, xyz_max)
        ctx.geom = (C, X, Y, Z, sC, sX, sY, sZ)
        ctx.grid_meta = grid
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        xyz, xyz_min, xyz_max = ctx.saved_tensors
        C, X, Y, Z, sC, sX, sY, sZ = ctx.geom
        grad_grid = None
        if ctx.needs_input_grad[0]:
            grad_out = grad_out.contiguous()
            # zero-filled, same strides as the parameter (F.grid_sample's backward does the same)
            grad_grid = torch.zeros_like(ctx.grid_meta, memory_format=torch.preserve_format)
            assert grad_grid.stride() == ctx.grid_meta.stride()
            with L.device_of(xyz):
                L.call('dvgo_grid_sample_bwd', ptr(grad_out), _int(C), _int(X), _int(Y), _int(Z), _i64(sC),
                       _i64(sX), _i64(sY), _i64(sZ), ptr(xyz), ptr(xyz_min), ptr(xyz_max), _i64(xyz.shape[0]),
                       ptr(grad_grid), stream_of(xyz))
        return grad_grid, None, None, None


def grid_sample(grid, xyz, xyz_min, xyz_max):
    """Trilinear interpolation with the exact contract of DirectVoxGO.grid_sampler
    (lib/dvgo.py:312-328): xyz [...,3] world coordinates -> [...,C], squeezed when C == 1.
    Differentiable w.r.t. ``grid`` (xyz never requires grad on this path)."""
    shape = xyz.shape[:-1]
    flat = xyz.reshape(-1, 3).contiguous()
    out = _GridSample.apply(grid, flat, xyz_min.contiguous(), xyz_max.contiguous())
    out = out.reshape(*shape, grid.shape[1])
    if out.shape[-1] == 1:
        out = out.squeeze(-1)
    return out


class MaskCache(nn.Module):
    """Occupancy grid for free-space skipping (lib/dvgo.py:583-613).  ``path`` loads a coarse
    checkpoint ({'model_state_dict': {'density'}, 'model_kwargs': {...}}) exactly like the
    reference; otherwise ``mask`` + bbox are given."""

    def __init__(self, path=None, mask_cache_thres=None, mask=None, xyz_min=None, xyz_max=None):
        super().__init__()
        if path is not None:
            from .checkpoint import safe_load
            st = safe_load(path)                    # weights-only: nothing in the file is executed
            self.mask_cache_thres = mask_cache_thres
            density = F.max_pool3d(st['model_state_dict']['density'].float().contiguous(), kernel_size=3,
                                   padding=1, stride=1)
            kw = st['model_kwargs']
            alpha = 1 - torch.exp(-F.softplus(density + float(kw['act_shift'])) * float(kw['voxel_size_ratio']))
            mask = (alpha >= self.mask_cache_thres).squeeze(0).squeeze(0)
            xyz_min = torch.as_tensor(kw['xyz_min'], dtype=torch.float32)
            xyz_max = torch.as_tensor(kw['xyz_max'], dtype=torch.float32)
        else:
            mask = mask.bool()
            xyz_min = torch.as_tensor(xyz_min, dtype=torch.float32).detach().clone()
            xyz_max = torch.as_tensor(xyz_max, dtype=torch.float32).detach().clone()
        self.register_buffer('mask', mask.contiguous())
        xyz_len = xyz_max - xyz_min
        scale = (torch.tensor(list(mask.shape), dtype=torch.float32, device=xyz_len.device) - 1) / xyz_len
        self.register_buffer('xyz2ijk_scale', scale)
        self.register_buffer('xyz2ijk_shift', -xyz_min.to(scale.device) * scale)

    @torch.no_grad()
    def forward(self, xyz):
        shape = xyz.shape[:-1]
        xyz = xyz.reshape(-1, 3).contiguous()
        mask = render_utils_hip.maskcache_lookup(self.mask, xyz, self.xyz2ijk_scale, self.xyz2ijk_shift)
        return mask.reshape(shape)


class _SegmentSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, out):
        check_input(src, 'src'); check_input(index, 'index'); check_f32(src, 'src')
        if index.dtype != torch.int64:
            raise RuntimeError('index must be int64')
        squeeze = src.dim() == 1
        C = 1 if squeeze else src.shape[1]
        res = out.clone().contiguous()
        with L.device_of(src):
            L.call('dvgo_segment_sum', ptr(src), ptr(index), _i64(src.shape[0]), _int(C), _i64(res.shape[0]),
                   ptr(res), stream_of(src))
        ctx.save_for_backward(index)
        return res

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        return g.index_select(0, index), None, g   # d/d src = gather, d/d out = identity


def segment_coo(src, index, out, reduce='sum'):
    """torch_scatter.segment_coo(src, index, out, reduce='sum') as used at
    lib/dvgo.py:554-559,571-575: index sorted, out = zeros[N(,C)]; returns the summed tensor."""
    if reduce != 'sum':
        raise NotImplementedError("only reduce='sum' is used by the reference")
    return _SegmentSum.apply(src.contiguous(), index, out)


def total_variation_add_grad(param, grad, wx, wy, wz, dense_mode, x_range=None):
    """total_variation_cuda.total_variation_add_grad (lib/cuda/total_variation.cpp:16-24);
    in place on ``grad``; param/grad [1,C,X,Y,Z] sharing one (dense) stride pattern.
    ``x_range=(lo, hi)``: only the planes lo <= x < hi (the slab a data-parallel rank owns)."""
    if not (param.is_cuda and grad.is_cuda):
        raise RuntimeError('param must be a CUDA tensor')
    if param.stride() != grad.stride():
        raise RuntimeError('param and grad must share strides')
    C, X, Y, Z, sC, sX, sY, sZ = _grid_geom(param)
    lo, hi = (0, X) if x_range is None else x_range
    with L.device_of(param):
        L.call('dvgo_total_variation_add_grad_slab', ptr(param), ptr(grad), _flt(float(wx)), _flt(float(wy)),
               _flt(float(wz)), _i64(C), _i64(X), _i64(Y), _i64(Z), _i64(sC), _i64(sX), _i64(sY), _i64(sZ),
               _int(1 if dense_mode else 0), _i64(lo), _i64(hi), stream_of(param))
