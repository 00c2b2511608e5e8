"""numpy front-end of the CPU oracle (oracle/dvgo_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of dvgo_oracle.c.  Importable from tests/,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of bench.py; never from
``directvoxgo_amd``.

Every function takes/returns C-contiguous numpy arrays (float32 / int64 / uint8-as-bool)
and mirrors one reference op; the reference file:line each follows is cited in the C file.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libdvgo_oracle.so')

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i64 = ctypes.c_int64
_int = ctypes.c_int
_flt = ctypes.c_float


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    src = os.path.join(_HERE, 'dvgo_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', '_build/libdvgo_oracle.so'])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _l(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(_i64p)


def _b(a):
    a = np.ascontiguousarray(a).astype(np.uint8, copy=False)
    a = np.ascontiguousarray(a)
    return a, a.ctypes.data_as(_u8p)


def _call(name, *args):
    fn = getattr(lib(), name)
    fn.restype = None
    fn(*args)


# --------------------------------------------------------------------------- K1-K3
def infer_t_minmax(rays_o, rays_d, xyz_min, xyz_max, near, far):
    rays_o, po = _f(rays_o); rays_d, pd = _f(rays_d)
    xyz_min, pmin = _f(xyz_min); xyz_max, pmax = _f(xyz_max)
    n = rays_o.shape[0]
    t_min = np.empty(n, np.float32); t_max = np.empty(n, np.float32)
    _call('ora_infer_t_minmax', po, pd, pmin, pmax, _flt(near), _flt(far), _i64(n),
          t_min.ctypes.data_as(_f32p), t_max.ctypes.data_as(_f32p))
    return t_min, t_max


def infer_n_samples(t_min, t_max, stepdist):
    t_min, p0 = _f(t_min); t_max, p1 = _f(t_max)
    n = t_min.shape[0]
    out = np.empty(n, np.int64)
    _call('ora_infer_n_samples', p0, p1, _flt(stepdist), _i64(n), out.ctypes.data_as(_i64p))
    return out


def infer_ray_start_dir(rays_o, rays_d, t_min):
    rays_o, po = _f(rays_o); rays_d, pd = _f(rays_d); t_min, pt = _f(t_min)
    n = rays_o.shape[0]
    start = np.empty((n, 3), np.float32); dirs = np.empty((n, 3), np.float32)
    _call('ora_infer_ray_start_dir', po, pd, pt, _i64(n),
          start.ctypes.data_as(_f32p), dirs.ctypes.data_as(_f32p))
    return start, dirs


# --------------------------------------------------------------------------- A1
def sample_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist):
    """-> rays_pts[M,3], mask_outbbox[M] bool, ray_id[M] i64, step_id[M] i64, N_steps[N] i64,
    t_min[N], t_max[N]   (render_utils_kernel.cu:190-236)"""
    rays_o, po = _f(rays_o); rays_d, pd = _f(rays_d)
    xyz_min, pmin = _f(xyz_min); xyz_max, pmax = _f(xyz_max)
    n = rays_o.shape[0]
    t_min = np.empty(n, np.float32); t_max = np.empty(n, np.float32)
    n_steps = np.empty(n, np.int64); cums = np.empty(n, np.int64)
    start = np.empty((n, 3), np.float32); dirs = np.empty((n, 3), np.float32)
    fn = lib().ora_sample_pts_prepare
    fn.restype = ctypes.c_int64
    total = fn(po, pd, pmin, pmax, _flt(near), _flt(far), _flt(stepdist), _i64(n),
               t_min.ctypes.data_as(_f32p), t_max.ctypes.data_as(_f32p),
               n_steps.ctypes.data_as(_i64p), cums.ctypes.data_as(_i64p),
               start.ctypes.data_as(_f32p), dirs.ctypes.data_as(_f32p))
    total = int(total) if n > 0 else 0
    pts = np.empty((total, 3), np.float32); mask = np.empty(total, np.uint8)
    ray_id = np.empty(total, np.int64); step_id = np.empty(total, np.int64)
    _call('ora_sample_pts_fill', start.ctypes.data_as(_f32p), dirs.ctypes.data_as(_f32p),
          pmin, pmax, n_steps.ctypes.data_as(_i64p), _i64(n), _flt(stepdist),
          pts.ctypes.data_as(_f32p), mask.ctypes.data_as(_u8p),
          ray_id.ctypes.data_as(_i64p), step_id.ctypes.data_as(_i64p))
    return pts, mask.astype(bool), ray_id, step_id, n_steps, t_min, t_max


# --------------------------------------------------------------------------- A2
def sample_ndc_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, n_samples):
    rays_o, po = _f(rays_o); rays_d, pd = _f(rays_d)
    xyz_min, pmin = _f(xyz_min); xyz_max, pmax = _f(xyz_max)
    n = rays_o.shape[0]
    pts = np.empty((n, n_samples, 3), np.float32); mask = np.empty((n, n_samples), np.uint8)
    _call('ora_sample_ndc_pts_on_rays', po, pd, pmin, pmax, _int(n_samples), _i64(n),
          pts.ctypes.data_as(_f32p), mask.ctypes.data_as(_u8p))
    return pts, mask.astype(bool)


# --------------------------------------------------------------------------- A3
def maskcache_lookup(world, xyz, scale, shift):
    world, pw = _b(world); xyz, px = _f(xyz); scale, ps = _f(scale); shift, ph = _f(shift)
    n = xyz.shape[0]
    out = np.zeros(n, np.uint8)
    _call('ora_maskcache_lookup', pw, px, ps, ph, _int(world.shape[0]), _int(world.shape[1]),
          _int(world.shape[2]), _i64(n), out.ctypes.data_as(_u8p))
    return out.astype(bool)


# --------------------------------------------------------------------------- A5
def raw2alpha(density, shift, interval):
    density, pd = _f(density)
    n = density.shape[0]
    e = np.empty(n, np.float32); a = np.empty(n, np.float32)
    _call('ora_raw2alpha', pd, _flt(shift), _flt(interval), _i64(n),
          e.ctypes.data_as(_f32p), a.ctypes.data_as(_f32p))
    return e, a


def raw2alpha_backward(exp_d, grad_back, interval):
    exp_d, pe = _f(exp_d); grad_back, pg = _f(grad_back)
    n = exp_d.shape[0]
    g = np.empty(n, np.float32)
    _call('ora_raw2alpha_backward', pe, pg, _flt(interval), _i64(n), g.ctypes.data_as(_f32p))
    return g


# --------------------------------------------------------------------------- A6/A7
def alpha2weight(alpha, ray_id, n_rays):
    alpha, pa = _f(alpha); ray_id, pr = _l(ray_id)
    m = alpha.shape[0]
    w = np.empty(m, np.float32); T = np.empty(m, np.float32)
    last = np.empty(n_rays, np.float32)
    i_start = np.empty(n_rays, np.int64); i_end = np.empty(n_rays, np.int64)
    _call('ora_alpha2weight', pa, pr, _i64(m), _i64(n_rays), w.ctypes.data_as(_f32p),
          T.ctypes.data_as(_f32p), last.ctypes.data_as(_f32p),
          i_start.ctypes.data_as(_i64p), i_end.ctypes.data_as(_i64p))
    return w, T, last, i_start, i_end


def alpha2weight_backward(alpha, weight, T, alphainv_last, i_start, i_end, n_rays,
                          grad_weights, grad_last, fma=True):
    alpha, pa = _f(alpha); weight, pw = _f(weight); T, pT = _f(T)
    alphainv_last, pl = _f(alphainv_last); i_start, ps = _l(i_start); i_end, pe = _l(i_end)
    grad_weights, pgw = _f(grad_weights); grad_last, pgl = _f(grad_last)
    m = alpha.shape[0]
    g = np.empty(m, np.float32)
    _call('ora_alpha2weight_backward_fma' if fma else 'ora_alpha2weight_backward',
          pa, pw, pT, pl, ps, pe, _i64(n_rays), _i64(m), pgw, pgl, g.ctypes.data_as(_f32p))
    return g


# --------------------------------------------------------------------------- A4/A8
def _grid_args(grid):
    """grid: numpy [C,X,Y,Z] view (any strides, float32).  Returns base array + element strides."""
    assert grid.dtype == np.float32 and grid.ndim == 4
    es = grid.itemsize
    C, X, Y, Z = grid.shape
    sC, sX, sY, sZ = (s // es for s in grid.strides)
    return C, X, Y, Z, sC, sX, sY, sZ


def grid_sample_fwd(grid, xyz, xyz_min, xyz_max, use_fma=True):
    """grid [C,X,Y,Z] (strided ok) , xyz [M,3] -> [M,C]   (lib/dvgo.py:312-328)"""
    C, X, Y, Z, sC, sX, sY, sZ = _grid_args(grid)
    xyz, px = _f(xyz); xyz_min, pmin = _f(xyz_min); xyz_max, pmax = _f(xyz_max)
    M = xyz.shape[0]
    out = np.empty((M, C), np.float32)
    _call('ora_grid_sample_fwd', ctypes.cast(grid.ctypes.data, _f32p), _int(C), _int(X), _int(Y),
          _int(Z), _i64(sC), _i64(sX), _i64(sY), _i64(sZ), px, pmin, pmax, _i64(M),
          _int(1 if use_fma else 0), out.ctypes.data_as(_f32p))
    return out


def grid_sample_bwd(grad_out, grid_shape, xyz, xyz_min, xyz_max, grad_grid=None):
    """grad_out [M,C] -> grad_grid [C,X,Y,Z] (contiguous; accumulated into when given)."""
    C, X, Y, Z = grid_shape
    grad_out, pg = _f(grad_out); xyz, px = _f(xyz)
    xyz_min, pmin = _f(xyz_min); xyz_max, pmax = _f(xyz_max)
    M = xyz.shape[0]
    if grad_grid is None:
        grad_grid = np.zeros((C, X, Y, Z), np.float32)
    assert grad_grid.flags['C_CONTIGUOUS'] and grad_grid.dtype == np.float32
    _call('ora_grid_sample_bwd', pg, _int(C), _int(X), _int(Y), _int(Z),
          _i64(X * Y * Z), _i64(Y * Z), _i64(Z), _i64(1), px, pmin, pmax, _i64(M),
          grad_grid.ctypes.data_as(_f32p))
    return grad_grid


# --------------------------------------------------------------------------- A9
def segment_sum(src, index, n_out):
    src, ps = _f(src); index, pi = _l(index)
    squeeze = src.ndim == 1
    C = 1 if squeeze else src.shape[1]
    out = np.zeros((n_out, C), np.float32)
    _call('ora_segment_sum', ps, pi, _i64(src.shape[0]), _int(C), _i64(n_out),
          out.ctypes.data_as(_f32p))
    return out[:, 0] if squeeze else out


# --------------------------------------------------------------------------- K14-K17
def adam_upd(param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps, mode=0, perlr=None):
    """In place on the given float32 contiguous arrays.  mode 0 plain, 1 masked, 2 per-lr."""
    for a in (param, grad, exp_avg, exp_avg_sq):
        assert a.dtype == np.float32 and a.flags['C_CONTIGUOUS']
    pl = perlr.ctypes.data_as(_f32p) if perlr is not None else ctypes.cast(0, _f32p)
    _call('ora_adam_upd', param.ctypes.data_as(_f32p), grad.ctypes.data_as(_f32p),
          exp_avg.ctypes.data_as(_f32p), exp_avg_sq.ctypes.data_as(_f32p), pl,
          _i64(param.size), _int(step), _flt(beta1), _flt(beta2), _flt(lr), _flt(eps), _int(mode))


def total_variation_add_grad(param, grad, wx, wy, wz, dense_mode):
    """param/grad: contiguous [1,C,X,Y,Z] float32; grad modified in place."""
    assert param.flags['C_CONTIGUOUS'] and grad.flags['C_CONTIGUOUS']
    _, _, X, Y, Z = param.shape
    _call('ora_total_variation_add_grad', param.ctypes.data_as(_f32p), grad.ctypes.data_as(_f32p),
          _flt(wx), _flt(wy), _flt(wz), _i64(X), _i64(Y), _i64(Z), _i64(param.size),
          _int(1 if dense_mode else 0))
