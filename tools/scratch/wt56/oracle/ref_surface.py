"""torch-tensor facade over the CPU oracle with the reference's native-op surface.

TEST INFRASTRUCTURE ONLY (see dvgo_oracle.c).  It exists so that the reference's own Python
orchestration (lib/dvgo.py, run under tests/golden/make_golden.py in the build container)
and the parity tests can call ``render_utils_cuda.<op>(...)`` with CPU tensors and get
what lib/cuda/render_utils.cpp:144-155 would have returned.

  render_utils  -- the 10 callables of render_utils_cuda (render_utils.cpp:44-141)
  total_variation, adam_upd -- the sibling extension surfaces
  segment_coo   -- torch_scatter.segment_coo(src, index, out, reduce='sum')
"""
import types

import numpy as np
import torch

from . import oracle as O


def _np(t):
    return t.detach().cpu().contiguous().numpy()


def _float(x):
    # pybind converts python floats / 0-dim tensors to C float (render_utils.cpp signatures)
    return float(x)


def infer_t_minmax(rays_o, rays_d, xyz_min, xyz_max, near, far):
    a, b = O.infer_t_minmax(_np(rays_o), _np(rays_d), _np(xyz_min), _np(xyz_max), _float(near), _float(far))
    return [torch.from_numpy(a), torch.from_numpy(b)]


def infer_n_samples(t_min, t_max, stepdist):
    return torch.from_numpy(O.infer_n_samples(_np(t_min), _np(t_max), _float(stepdist)))


def infer_ray_start_dir(rays_o, rays_d, t_min):
    a, b = O.infer_ray_start_dir(_np(rays_o), _np(rays_d), _np(t_min))
    return [torch.from_numpy(a), torch.from_numpy(b)]


def sample_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist):
    out = O.sample_pts_on_rays(_np(rays_o), _np(rays_d), _np(xyz_min), _np(xyz_max),
                               _float(near), _float(far), _float(stepdist))
    return [torch.from_numpy(np.ascontiguousarray(x)) for x in out]


def sample_ndc_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, N_samples):
    a, b = O.sample_ndc_pts_on_rays(_np(rays_o), _np(rays_d), _np(xyz_min), _np(xyz_max), int(N_samples))
    return [torch.from_numpy(a), torch.from_numpy(b)]


def maskcache_lookup(world, xyz, xyz2ijk_scale, xyz2ijk_shift):
    return torch.from_numpy(O.maskcache_lookup(_np(world), _np(xyz), _np(xyz2ijk_scale), _np(xyz2ijk_shift)))


def raw2alpha(density, shift, interval):
    e, a = O.raw2alpha(_np(density), _float(shift), _float(interval))
    return [torch.from_numpy(e), torch.from_numpy(a)]


def raw2alpha_backward(exp, grad_back, interval):
    return torch.from_numpy(O.raw2alpha_backward(_np(exp), _np(grad_back), _float(interval)))


def alpha2weight(alpha, ray_id, n_rays):
    out = O.alpha2weight(_np(alpha), _np(ray_id), int(n_rays))
    return [torch.from_numpy(x) for x in out]


def alpha2weight_backward(alpha, weight, T, alphainv_last, i_start, i_end, n_rays, grad_weights, grad_last):
    g = O.alpha2weight_backward(_np(alpha), _np(weight), _np(T), _np(alphainv_last), _np(i_start),
                                _np(i_end), int(n_rays), _np(grad_weights), _np(grad_last))
    return torch.from_numpy(g)


render_utils = types.SimpleNamespace(
    infer_t_minmax=infer_t_minmax, infer_n_samples=infer_n_samples,
    infer_ray_start_dir=infer_ray_start_dir, sample_pts_on_rays=sample_pts_on_rays,
    sample_ndc_pts_on_rays=sample_ndc_pts_on_rays, maskcache_lookup=maskcache_lookup,
    raw2alpha=raw2alpha, raw2alpha_backward=raw2alpha_backward,
    alpha2weight=alpha2weight, alpha2weight_backward=alpha2weight_backward)


def total_variation_add_grad(param, grad, wx, wy, wz, dense_mode):
    p = _np(param); g = grad.detach().numpy()
    assert g.flags['C_CONTIGUOUS']
    O.total_variation_add_grad(p, g, _float(wx), _float(wy), _float(wz), bool(dense_mode))


total_variation = types.SimpleNamespace(total_variation_add_grad=total_variation_add_grad)


def _adam(mode):
    def fn(param, grad, exp_avg, exp_avg_sq, *rest):
        if mode == 2:
            perlr, step, beta1, beta2, lr, eps = rest
            perlr = _np(perlr)
        else:
            step, beta1, beta2, lr, eps = rest
            perlr = None
        O.adam_upd(param.detach().numpy(), _np(grad), exp_avg.numpy(), exp_avg_sq.numpy(),
                   int(step), beta1, beta2, lr, eps, mode=mode, perlr=perlr)
    return fn


adam_upd = types.SimpleNamespace(adam_upd=_adam(0), masked_adam_upd=_adam(1), adam_upd_with_perlr=_adam(2))


def segment_coo(src, index, out, reduce='sum'):
    """torch_scatter.segment_coo stand-in (differentiable: index_add)."""
    assert reduce == 'sum'
    return out.index_add(0, index, src)
