"""Drop-in for the reference's ``render_utils_cuda`` extension module.

Same 10 callables, same positional signatures, same return conventions and error text as
/root/reference/lib/cuda/render_utils.cpp:44-155, so ``lib/dvgo.py`` / ``lib/dmpigo.py`` can
bind this module where they bind ``load(name='render_utils_cuda', ...)`` (see INTEGRATION.md
and directvoxgo_amd/compat.py).  Every op runs a hand-written gfx950 kernel through the C ABI
of include/dvgo_hip.h on torch's current stream and on the inputs' device.
"""
import torch

from . import _lib as L
from ._lib import _flt, _i64, _int, check_f32, check_input, ptr, stream_of


def _f(x):
    # pybind converts python floats and 0-dim tensors to C float
    return _flt(float(x))


def infer_t_minmax(rays_o, rays_d, xyz_min, xyz_max, near, far):
    """render_utils.cpp:44-52 -> [t_min, t_max]"""
    for x, n in ((rays_o, 'rays_o'), (rays_d, 'rays_d'), (xyz_min, 'xyz_min'), (xyz_max, 'xyz_max')):
        check_input(x, n); check_f32(x, n)
    n = rays_o.shape[0]
    t_min = torch.empty(n, dtype=torch.float32, device=rays_o.device)
    t_max = torch.empty_like(t_min)
    with L.device_of(rays_o):
        L.call('dvgo_infer_t_minmax', ptr(rays_o), ptr(rays_d), ptr(xyz_min), ptr(xyz_max), _f(near), _f(far),
               _i64(n), ptr(t_min), ptr(t_max), stream_of(rays_o))
    return [t_min, t_max]


def infer_n_samples(t_min, t_max, stepdist):
    """render_utils.cpp:54-58 -> n_samples int64"""
    for x, n in ((t_min, 't_min'), (t_max, 't_max')):
        check_input(x, n); check_f32(x, n)
    n = t_min.shape[0]
    out = torch.empty(n, dtype=torch.int64, device=t_min.device)
    with L.device_of(t_min):
        L.call('dvgo_infer_n_samples', ptr(t_min), ptr(t_max), _f(stepdist), _i64(n), ptr(out), stream_of(t_min))
    return out


def infer_ray_start_dir(rays_o, rays_d, t_min):
    """render_utils.cpp:60-65 -> [rays_start, rays_dir]"""
    for x, n in ((rays_o, 'rays_o'), (rays_d, 'rays_d'), (t_min, 't_min')):
        check_input(x, n); check_f32(x, n)
    n = rays_o.shape[0]
    start = torch.empty_like(rays_o)
    dirs = torch.empty_like(rays_o)
    with L.device_of(rays_o):
        L.call('dvgo_infer_ray_start_dir', ptr(rays_o), ptr(rays_d), ptr(t_min), _i64(n), ptr(start), ptr(dirs),
               stream_of(rays_o))
    return [start, dirs]


def _prepare(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist):
    """K1+K2+K3+cumsum on device; returns the per-ray tensors (no host sync)."""
    n = rays_o.shape[0]
    dev = rays_o.device
    t_min = torch.empty(n, dtype=torch.float32, device=dev)
    t_max = torch.empty_like(t_min)
    n_steps = torch.empty(n, dtype=torch.int64, device=dev)
    cum = torch.empty(n, dtype=torch.int64, device=dev)
    start = torch.empty((n, 3), dtype=torch.float32, device=dev)
    dirs = torch.empty((n, 3), dtype=torch.float32, device=dev)
    with L.device_of(rays_o):
        L.call('dvgo_sample_pts_prepare', ptr(rays_o), ptr(rays_d), ptr(xyz_min), ptr(xyz_max), _f(near), _f(far),
               _f(stepdist), _i64(n), ptr(t_min), ptr(t_max), ptr(n_steps), ptr(cum), ptr(start), ptr(dirs),
               stream_of(rays_o))
    return t_min, t_max, n_steps, cum, start, dirs


def sample_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist):
    """render_utils.cpp:67-78 -> [rays_pts, mask_outbbox, ray_id, step_id, N_steps, t_min, t_max]

    One device->host read of the total sample count (the reference's own ``.item()``,
    render_utils_kernel.cu:206).  Zero rays are accepted (run.py:91 produces empty chunks).
    """
    for x, n in ((rays_o, 'rays_o'), (rays_d, 'rays_d'), (xyz_min, 'xyz_min'), (xyz_max, 'xyz_max')):
        check_input(x, n); check_f32(x, n)
    assert rays_o.dim() == 2 and rays_o.shape[1] == 3
    n = rays_o.shape[0]
    dev = rays_o.device
    t_min, t_max, n_steps, cum, start, dirs = _prepare(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist)
    total = int(cum[-1].item()) if n > 0 else 0
    pts = torch.empty((total, 3), dtype=torch.float32, device=dev)
    mask = torch.empty(total, dtype=torch.bool, device=dev)
    ray_id = torch.empty(total, dtype=torch.int64, device=dev)
    step_id = torch.empty(total, dtype=torch.int64, device=dev)
    with L.device_of(rays_o):
        L.call('dvgo_sample_pts_fill', ptr(start), ptr(dirs), ptr(xyz_min), ptr(xyz_max), ptr(cum), _i64(n),
               _f(stepdist), _i64(total), ptr(pts), ptr(mask), ptr(ray_id), ptr(step_id), stream_of(rays_o))
    return [pts, mask, ray_id, step_id, n_steps, t_min, t_max]


def sample_ndc_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, N_samples):
    """render_utils.cpp:80-91 -> [rays_pts [N,S,3], mask_outbbox [N,S]]"""
    for x, n in ((rays_o, 'rays_o'), (rays_d, 'rays_d'), (xyz_min, 'xyz_min'), (xyz_max, 'xyz_max')):
        check_input(x, n); check_f32(x, n)
    assert rays_o.dim() == 2 and rays_o.shape[1] == 3
    n = rays_o.shape[0]
    S = int(N_samples)
    pts = torch.empty((n, S, 3), dtype=torch.float32, device=rays_o.device)
    mask = torch.empty((n, S), dtype=torch.bool, device=rays_o.device)
    with L.device_of(rays_o):
        L.call('dvgo_sample_ndc_pts_on_rays', ptr(rays_o), ptr(rays_d), ptr(xyz_min), ptr(xyz_max), _int(S), _i64(n),
               ptr(pts), ptr(mask), stream_of(rays_o))
    return [pts, mask]


def maskcache_lookup(world, xyz, xyz2ijk_scale, xyz2ijk_shift):
    """render_utils.cpp:93-102 -> bool [n_pts]"""
    for x, n in ((world, 'world'), (xyz, 'xyz'), (xyz2ijk_scale, 'xyz2ijk_scale'), (xyz2ijk_shift, 'xyz2ijk_shift')):
        check_input(x, n)
    if world.dtype != torch.bool:
        raise RuntimeError('world must be a bool tensor')
    check_f32(xyz, 'xyz')
    assert world.dim() == 3 and xyz.dim() == 2 and xyz.shape[1] == 3
    n = xyz.shape[0]
    out = torch.empty(n, dtype=torch.bool, device=xyz.device)
    with L.device_of(xyz):
        L.call('dvgo_maskcache_lookup', ptr(world), ptr(xyz), ptr(xyz2ijk_scale), ptr(xyz2ijk_shift),
               _int(world.shape[0]), _int(world.shape[1]), _int(world.shape[2]), _i64(n), ptr(out), stream_of(xyz))
    return out


def raw2alpha(density, shift, interval):
    """render_utils.cpp:104-108 -> [exp, alpha]"""
    check_input(density, 'density'); check_f32(density, 'density')
    assert density.dim() == 1
    e = torch.empty_like(density)
    a = torch.empty_like(density)
    with L.device_of(density):
        L.call('dvgo_raw2alpha', ptr(density), _f(shift), _f(interval), _i64(density.shape[0]), ptr(e), ptr(a),
               stream_of(density))
    return [e, a]


def raw2alpha_backward(exp, grad_back, interval):
    """render_utils.cpp:110-114 -> grad"""
    check_input(exp, 'exp'); check_input(grad_back, 'grad_back')
    check_f32(exp, 'exp'); check_f32(grad_back, 'grad_back')
    g = torch.empty_like(exp)
    with L.device_of(exp):
        L.call('dvgo_raw2alpha_backward', ptr(exp), ptr(grad_back), _f(interval), _i64(exp.shape[0]), ptr(g),
               stream_of(exp))
    return g


def alpha2weight(alpha, ray_id, n_rays):
    """render_utils.cpp:116-123 -> [weights, T, alphainv_last, i_start, i_end]"""
    check_input(alpha, 'alpha'); check_input(ray_id, 'ray_id'); check_f32(alpha, 'alpha')
    if ray_id.dtype != torch.int64:
        raise RuntimeError('ray_id must be int64')
    assert alpha.dim() == 1 and ray_id.dim() == 1 and alpha.shape == ray_id.shape
    n_rays = int(n_rays)
    m = alpha.shape[0]
    dev = alpha.device
    w = torch.empty_like(alpha)
    T = torch.empty_like(alpha)
    last = torch.empty(n_rays, dtype=torch.float32, device=dev)
    i_start = torch.empty(n_rays, dtype=torch.int64, device=dev)
    i_end = torch.empty(n_rays, dtype=torch.int64, device=dev)
    with L.device_of(alpha):
        L.call('dvgo_alpha2weight', ptr(alpha), ptr(ray_id), _i64(m), _i64(n_rays), ptr(w), ptr(T), ptr(last),
               ptr(i_start), ptr(i_end), stream_of(alpha))
    return [w, T, last, i_start, i_end]


def alpha2weight_backward(alpha, weight, T, alphainv_last, i_start, i_end, n_rays, grad_weights, grad_last):
    """render_utils.cpp:125-141 -> grad"""
    for x, n in ((alpha, 'alpha'), (weight, 'weight'), (T, 'T'), (alphainv_last, 'alphainv_last'),
                 (i_start, 'i_start'), (i_end, 'i_end'), (grad_weights, 'grad_weights'), (grad_last, 'grad_last')):
        check_input(x, n)
    g = torch.empty_like(alpha)
    with L.device_of(alpha):
        L.call('dvgo_alpha2weight_backward', ptr(alpha), ptr(weight), ptr(T), ptr(alphainv_last), ptr(i_start),
               ptr(i_end), _i64(int(n_rays)), _i64(alpha.shape[0]), ptr(grad_weights), ptr(grad_last), ptr(g),
               stream_of(alpha))
    return g
