"""Generate the golden fixtures in tests/golden/*.npz by RUNNING THE REFERENCE's own Python.

Runs only in the build container (needs /root/reference); the fixtures it writes are
committed, the reference is not.  A fixture is data: inputs + expected outputs.

How the reference is run without a GPU / without touching its tree
------------------------------------------------------------------
* The needed reference modules (lib/dvgo.py, lib/dmpigo.py, lib/ray_utils.py,
  lib/masked_adam.py, lib/multiscene_dvgo.py, lib/load_blender.py ...) are copied to a scratch
  directory under /tmp and imported from there (never from /root/reference: importing them in
  place would let torch's cpp_extension.load() hipify into the read-only tree).
* ``torch.utils.cpp_extension.load`` is replaced *before* import by a function that returns,
  by extension name, the oracle-backed stand-ins of oracle/ref_surface.py
  (render_utils_cuda / total_variation_cuda / adam_upd_cuda).  ``torch_scatter`` (absent from
  the image) is replaced by an index_add based segment_coo.  Third-party modules the data
  loaders import but this script never calls (imageio, cv2, torchvision) are stubbed.

So every fixture below is  "reference orchestration / reference PyTorch code  o  oracle native
ops".  Pure-PyTorch reference fragments (grid_sampler, sample_ray_py, voxel_count_views,
MaskCache(path=...), get_rays_of_a_view, pose_spherical) do not touch the oracle at all and are
what PINS the oracle; the forward()/MaskedAdam fixtures pin the orchestration the product's
host code must reproduce.

Usage:  python tests/golden/make_golden.py           (writes next to this file)
"""
import os
import shutil
import sys
import tempfile
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from oracle import ref_surface as RS  # noqa: E402


def import_reference():
    scratch = tempfile.mkdtemp(prefix='dvgo_ref_', dir='/tmp')
    shutil.copytree(os.path.join(REF, 'lib'), os.path.join(scratch, 'lib'),
                    ignore=shutil.ignore_patterns('cuda', '__pycache__'))
    open(os.path.join(scratch, 'lib', '__init__.py'), 'a').close()

    def fake_load(name, sources=None, **kw):
        return {'render_utils_cuda': RS.render_utils,
                'total_variation_cuda': RS.total_variation,
                'adam_upd_cuda': RS.adam_upd}[name]

    import torch.utils.cpp_extension as cpp_ext
    cpp_ext.load = fake_load
    ts = types.ModuleType('torch_scatter')
    ts.segment_coo = RS.segment_coo
    ts.scatter_add = None
    sys.modules['torch_scatter'] = ts
    for missing in ('imageio', 'cv2', 'torchvision', 'torchvision.transforms',
                    'torchvision.models', 'mmcv', 'lpips'):
        try:
            __import__(missing)
        except Exception:
            sys.modules[missing] = mock.MagicMock()
    sys.path.insert(0, scratch)
    import lib.dvgo as dvgo
    import lib.dmpigo as dmpigo
    import lib.ray_utils as ray_utils
    import lib.masked_adam as masked_adam
    import lib.load_blender as load_blender
    import lib.multiscene_dvgo as multiscene_dvgo
    return types.SimpleNamespace(dvgo=dvgo, dmpigo=dmpigo, ray_utils=ray_utils,
                                 masked_adam=masked_adam, load_blender=load_blender,
                                 multiscene_dvgo=multiscene_dvgo, scratch=scratch)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


def lego_like_rays(R, rng, n_views, H, W, focal, radius=4.0):
    """Cameras as in lib/load_blender.py:37-42,83 (pose_spherical, phi=-30), pixel-centre rays
    by lib/ray_utils.py get_rays_of_a_view."""
    ro, rd, vd = [], [], []
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float32)
    for _ in range(n_views):
        theta = float(rng.uniform(-180, 180))
        c2w = R.load_blender.pose_spherical(theta, -30.0, radius)
        o, d, v = R.ray_utils.get_rays_of_a_view(H, W, K, c2w, ndc=False, inverse_y=False,
                                                 flip_x=False, flip_y=False, mode='center')
        ro.append(o.reshape(-1, 3)); rd.append(d.reshape(-1, 3)); vd.append(v.reshape(-1, 3))
    return (torch.cat(ro).contiguous().float(), torch.cat(rd).contiguous().float(),
            torch.cat(vd).contiguous().float())


def blob_density(world_size, xyz_min, xyz_max, rng, amp=12.0, bias=-4.0, noise=0.5):
    """SURVEY.md 8d synthetic density: smooth blob + low-passed noise."""
    X, Y, Z = world_size
    gx = np.linspace(xyz_min[0], xyz_max[0], X, dtype=np.float32)
    gy = np.linspace(xyz_min[1], xyz_max[1], Y, dtype=np.float32)
    gz = np.linspace(xyz_min[2], xyz_max[2], Z, dtype=np.float32)
    xx, yy, zz = np.meshgrid(gx, gy, gz, indexing='ij')
    r = np.sqrt(xx ** 2 + yy ** 2 + zz ** 2) / 0.9
    d = amp * np.exp(-r ** 4) + bias
    n = rng.standard_normal((X, Y, Z)).astype(np.float32)
    n = torch.nn.functional.avg_pool3d(torch.from_numpy(n)[None, None], 3, 1, 1).numpy()[0, 0]
    return (d + noise * 3 * n).astype(np.float32)


# --------------------------------------------------------------------------------------
def gen_constants(R):
    """Grid sizing, act_shift, voxel_size_ratio, MaskCache affine map (lib/dvgo.py:55-62,
    155-160,600-602) for the BASELINE configs."""
    rows = {}
    for tag, nv, nvb, ainit, scale in [('coarse100', 1024000, 1024000, 1e-6, 1.0),
                                       ('fine160', 160 ** 3, 160 ** 3, 1e-2, 1.05),
                                       ('fine256', 256 ** 3, 256 ** 3, 1e-2, 1.05),
                                       ('pg63', 160 ** 3 // 16, 160 ** 3, 1e-2, 1.05),
                                       ('aniso', 90000, 160 ** 3, 1e-2, 1.0)]:
        if tag == 'aniso':
            mn, mx = np.array([-1.0, -0.7, -0.4], np.float32), np.array([1.2, 0.9, 0.5], np.float32)
        else:
            mn, mx = np.array([-1.5] * 3, np.float32) * scale, np.array([1.5] * 3, np.float32) * scale
        # rgbnet_dim=0 -> k0 is [1,3,X,Y,Z]; 256^3 allocates ~270 MB on the host, fine here
        m = R.dvgo.DirectVoxGO(mn, mx, num_voxels=nv, num_voxels_base=nvb, alpha_init=ainit,
                               fast_color_thres=1e-4, rgbnet_dim=0)
        rows[tag + '_xyz_min'] = mn; rows[tag + '_xyz_max'] = mx
        rows[tag + '_num_voxels'] = nv; rows[tag + '_num_voxels_base'] = nvb
        rows[tag + '_alpha_init'] = ainit
        rows[tag + '_world_size'] = m.world_size.numpy()
        rows[tag + '_voxel_size'] = m.voxel_size.numpy()
        rows[tag + '_voxel_size_ratio'] = m.voxel_size_ratio.numpy()
        rows[tag + '_act_shift'] = np.float64(m.act_shift)
        rows[tag + '_xyz2ijk_scale'] = m.mask_cache.xyz2ijk_scale.numpy()
        rows[tag + '_xyz2ijk_shift'] = m.mask_cache.xyz2ijk_shift.numpy()
        del m
    save('constants', **rows)


def gen_grid_sampler(R):
    """A4/A8: reference DirectVoxGO.grid_sampler (lib/dvgo.py:312-328) forward + autograd backward."""
    rng = np.random.default_rng(101)
    mn, mx = np.array([-1.0, -0.8, -0.6], np.float32), np.array([0.9, 1.1, 0.7], np.float32)
    m = R.dvgo.DirectVoxGO(mn, mx, num_voxels=14 * 15 * 11, num_voxels_base=4096, alpha_init=1e-2,
                           fast_color_thres=1e-4, rgbnet_dim=12, rgbnet_width=16)
    ws = tuple(int(v) for v in m.world_size)
    out = {'xyz_min': mn, 'xyz_max': mx, 'world_size': np.array(ws)}
    M = 1500
    xyz = (rng.random((M, 3)) * (mx - mn) + mn).astype(np.float32)
    # a few points exactly on faces / corners, and a few outside (zero padding)
    xyz[:3] = [mn, mx, (mn + mx) / 2]
    xyz[3] = [mn[0], mx[1], 0.1]
    xyz[4:12] += (mx - mn) * rng.choice([-1.02, 1.02], size=(8, 3)).astype(np.float32) * 0.5
    out['xyz'] = xyz
    for C in (1, 3, 12):
        grid = torch.from_numpy(rng.standard_normal((1, C, *ws)).astype(np.float32)).requires_grad_()
        go = rng.standard_normal((M, C)).astype(np.float32)
        val = m.grid_sampler(torch.from_numpy(xyz), grid)
        val.reshape(M, C).backward(torch.from_numpy(go))
        out[f'grid_c{C}'] = grid.detach().numpy()
        out[f'out_c{C}'] = val.detach().numpy()
        out[f'gout_c{C}'] = go
        out[f'ggrid_c{C}'] = grid.grad.numpy()
    save('grid_sampler', **out)


def gen_sampler_py(R):
    """K1/K3/K6 cross-check: the surviving PyTorch sampler lib/multiscene_dvgo.py:493-515
    (fixed-length, step/|d| parametrisation) and MaskCache lookups via the oracle natives."""
    rng = np.random.default_rng(202)
    mn, mx = np.array([-1.0, -1.0, -1.0], np.float32), np.array([1.0, 1.0, 1.0], np.float32)
    ro, rd, vd = lego_like_rays(R, rng, n_views=3, H=6, W=6, focal=6 * 1111.11 / 800 * 2.5)
    # special rays: axis aligned (zero components), missing the box, origin inside the box
    extra_o = np.array([[-3, 0.2, 0.1], [0.3, -3, 0.2], [0.1, 0.2, 3.0], [3, 3, 3], [0.1, -0.2, 0.3],
                        [-3, 0.2, 0.1]], np.float32)
    extra_d = np.array([[1, 0, 0], [0, 2, 0], [0, 0, -0.5], [1, 0.1, 0.1], [0.3, 0.5, -0.2],
                        [1.7, 0.05, -0.02]], np.float32)
    ro = torch.cat([ro, torch.from_numpy(extra_o)]); rd = torch.cat([rd, torch.from_numpy(extra_d)])
    fake = types.SimpleNamespace(density=torch.zeros(1, 1, 20, 20, 20), xyz_min=torch.from_numpy(mn),
                                 xyz_max=torch.from_numpy(mx),
                                 voxel_size=torch.tensor(2.0 / 20, dtype=torch.float32))
    near, far, stepsize = 0.2, 6.0, 0.5
    pts, mask = R.multiscene_dvgo.DirectVoxGO.sample_ray_py(fake, ro, rd, near, far, stepsize, is_train=False)
    save('sampler_py', rays_o=ro, rays_d=rd, xyz_min=mn, xyz_max=mx, near=near, far=far,
         stepsize=stepsize, voxel_size=np.float32(2.0 / 20), rays_pts=pts, mask_outbbox=mask)


def gen_maskcache_path(R):
    """A5 cross-check: MaskCache(path=...) (lib/dvgo.py:586-593) evaluates the softplus form of
    the activation on a max-pooled density and thresholds it."""
    rng = np.random.default_rng(303)
    dens = (rng.standard_normal((1, 1, 12, 13, 14)) * 4).astype(np.float32)
    mn, mx = np.array([-1.0, -1.1, -1.2], np.float32), np.array([1.0, 1.1, 1.2], np.float32)
    out = {'density': dens, 'xyz_min': mn, 'xyz_max': mx}
    tmp = tempfile.mkdtemp(prefix='dvgo_ck_', dir='/tmp')
    for i, (shift, ratio, thres) in enumerate([(-4.5951, 1.0, 1e-3), (-13.8155, 0.5, 1e-7), (0.0, 2.0, 0.5)]):
        p = os.path.join(tmp, f'ck{i}.tar')
        torch.save({'model_state_dict': {'density': torch.from_numpy(dens)},
                    'model_kwargs': {'act_shift': shift, 'voxel_size_ratio': ratio,
                                     'xyz_min': mn.tolist(), 'xyz_max': mx.tolist()}}, p)
        mc = R.dvgo.MaskCache(path=p, mask_cache_thres=thres)
        out[f'case{i}_params'] = np.array([shift, ratio, thres], np.float64)
        out[f'case{i}_mask'] = mc.mask.numpy()
        out[f'case{i}_scale'] = mc.xyz2ijk_scale.numpy()
        out[f'case{i}_shift'] = mc.xyz2ijk_shift.numpy()
    shutil.rmtree(tmp)
    save('maskcache_path', **out)


def _scene(R, rng, fine=True, nvox=16 ** 3, width=32, direct=False):
    mn, mx = np.array([-1.05, -1.05, -1.05], np.float32), np.array([1.05, 1.05, 1.05], np.float32)
    kw = dict(num_voxels=nvox, num_voxels_base=nvox, alpha_init=1e-2 if fine else 1e-6,
              fast_color_thres=1e-4 if fine else 1e-7)
    if fine:
        kw.update(rgbnet_dim=12, rgbnet_depth=3, rgbnet_width=width, viewbase_pe=4, rgbnet_direct=direct)
    else:
        kw.update(rgbnet_dim=0)
    torch.manual_seed(777)
    m = R.dvgo.DirectVoxGO(mn, mx, **kw)
    ws = tuple(int(v) for v in m.world_size)
    with torch.no_grad():
        dens = blob_density(ws, mn, mx, rng, amp=12.0 if fine else 20.0, bias=-4.0 if fine else -6.0)
        m.density.copy_(torch.from_numpy(dens)[None, None])
        m.k0.copy_(torch.from_numpy((rng.standard_normal(m.k0.shape) * 0.3).astype(np.float32)))
        alpha = torch.nn.functional.max_pool3d(m.activate_density(m.density), 3, 1, 1)[0, 0]
        m.mask_cache.mask.copy_(alpha > m.fast_color_thres)
    return m, mn, mx


def _loss(render_result, target, n_rays, w_main=1.0, w_ent=0.001, w_per=0.01):
    """run.py:377-386 restated (run.py itself needs mmcv + datasets and is not importable)."""
    import torch.nn.functional as F
    loss = w_main * F.mse_loss(render_result['rgb_marched'], target)
    pout = render_result['alphainv_last'].clamp(1e-6, 1 - 1e-6)
    loss = loss + w_ent * (-(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout)).mean())
    rgbper = (render_result['raw_rgb'] - target[render_result['ray_id']]).pow(2).sum(-1)
    loss = loss + w_per * ((rgbper * render_result['weights'].detach()).sum() / n_rays)
    return loss


def gen_forward(R, fine, name, width=32, direct=False):
    """H1/H2: reference DirectVoxGO.forward + hit_coarse_geo + autograd backward
    (lib/dvgo.py:412-577,618-660) on a small scene; natives = oracle.
    `forward_fine_direct` is the head of configs/default.py as run.py builds it for configs/nerf/lego.py:
    rgbnet_direct=True (no diffuse term, all 12 features into the MLP) and width 128."""
    rng = np.random.default_rng((404 if fine else 405) + (1000 if direct else 0))
    m, mn, mx = _scene(R, rng, fine, width=width, direct=direct)
    ro, rd, vd = lego_like_rays(R, rng, n_views=4, H=5, W=5, focal=5 * 1111.11 / 800 * 3.0, radius=3.0)
    # two rays that miss the box entirely and one starting inside it
    ro = torch.cat([ro, torch.tensor([[3.0, 3.0, 3.0], [0.0, 0.0, 3.0], [0.1, 0.1, 0.2]])])
    rd = torch.cat([rd, torch.tensor([[1.0, 0.2, 0.1], [0.0, 1.0, 0.0], [0.4, -0.3, 0.2]])])
    vd = rd / rd.norm(dim=-1, keepdim=True)
    N = ro.shape[0]
    target = torch.from_numpy(rng.random((N, 3)).astype(np.float32))
    rk = dict(near=0.5, far=6.0, bg=1, stepsize=0.5, inverse_y=False, flip_x=False, flip_y=False,
              render_depth=True)
    res = m(ro, rd, vd, global_step=0, **rk)
    loss = _loss(res, target, N, w_ent=0.001 if fine else 0.01, w_per=0.01 if fine else 0.1)
    loss.backward()
    hit = m.hit_coarse_geo(rays_o=ro, rays_d=rd, **rk)
    ray_pts, ray_id, step_id = m.sample_ray(rays_o=ro, rays_d=rd, **rk)
    out = dict(xyz_min=mn, xyz_max=mx, world_size=m.world_size.numpy(), density=m.density, k0=m.k0,
               mask=m.mask_cache.mask, act_shift=np.float64(m.act_shift),
               voxel_size=m.voxel_size.numpy(), voxel_size_ratio=m.voxel_size_ratio.numpy(),
               fast_color_thres=np.float64(m.fast_color_thres),
               rays_o=ro, rays_d=rd, viewdirs=vd, target=target,
               near=rk['near'], far=rk['far'], bg=rk['bg'], stepsize=rk['stepsize'],
               loss=loss.detach(), hit=hit, sample_ray_pts=ray_pts, sample_ray_id=ray_id,
               sample_step_id=step_id, grad_density=m.density.grad, grad_k0=m.k0.grad)
    for k, v in res.items():
        out['out_' + k] = v
    if m.rgbnet is not None:
        for k, v in m.rgbnet.state_dict().items():
            out['rgbnet_' + k] = v
        for k, v in m.rgbnet.named_parameters():
            out['grad_rgbnet_' + k] = v.grad
    save(name, **out)


def gen_mpi_forward(R, width=16, name='forward_mpi'):
    """Config-4 path: reference DirectMPIGO.forward (lib/dmpigo.py:173-283) incl. K7.
    `forward_mpi_w64` has the 64-wide head of configs/llff/llff_default.py (the fused colour-head kernels)."""
    rng = np.random.default_rng(506 + (width if width != 16 else 0))
    mn, mx = np.array([-1.2, -1.0, -1.0], np.float32), np.array([1.2, 1.0, 1.0], np.float32)
    torch.manual_seed(777)
    m = R.dmpigo.DirectMPIGO(mn, mx, num_voxels=12 * 10 * 16, mpi_depth=16, fast_color_thres=1e-3,
                             rgbnet_dim=9, rgbnet_depth=3, rgbnet_width=width, viewbase_pe=0)
    with torch.no_grad():
        m.density.add_(torch.from_numpy((rng.standard_normal(m.density.shape) * 2).astype(np.float32)))
        m.k0.copy_(torch.from_numpy((rng.standard_normal(m.k0.shape) * 0.3).astype(np.float32)))
    N = 40
    ro = torch.from_numpy(np.concatenate([rng.uniform(-1.1, 1.1, (N, 1)), rng.uniform(-0.9, 0.9, (N, 1)),
                                          -np.ones((N, 1))], 1).astype(np.float32))
    rd = torch.from_numpy(np.concatenate([rng.uniform(-0.4, 0.4, (N, 2)), 2 * np.ones((N, 1))], 1).astype(np.float32))
    vd = rd / rd.norm(dim=-1, keepdim=True)
    target = torch.from_numpy(rng.random((N, 3)).astype(np.float32))
    rk = dict(near=0, far=1, bg=0, stepsize=0.5, render_depth=True)
    res = m(ro, rd, vd, global_step=0, **rk)
    loss = _loss(res, target, N)
    loss.backward()
    out = dict(xyz_min=mn, xyz_max=mx, world_size=m.world_size.numpy(), mpi_depth=16, density=m.density,
               k0=m.k0, mask=m.mask_cache.mask, voxel_size_ratio=np.float64(m.voxel_size_ratio),
               fast_color_thres=np.float64(m.fast_color_thres), rays_o=ro, rays_d=rd, viewdirs=vd,
               target=target, bg=0, stepsize=0.5, loss=loss.detach(),
               grad_density=m.density.grad, grad_k0=m.k0.grad)
    for k, v in res.items():
        out['out_' + k] = v
    for k, v in m.rgbnet.state_dict().items():
        out['rgbnet_' + k] = v
    for k, v in m.rgbnet.named_parameters():
        out['grad_rgbnet_' + k] = v.grad
    save(name, **out)


def gen_voxel_count_views(R):
    """Pure-PyTorch reference path (lib/dvgo.py:265-295): slab test + fixed-length sampling +
    grid_sample backward, no native op involved."""
    rng = np.random.default_rng(607)
    m, mn, mx = _scene(R, rng, fine=False, nvox=12 ** 3)
    ro, rd, vd = lego_like_rays(R, rng, n_views=3, H=8, W=8, focal=8 * 1111.11 / 800 * 2.0, radius=3.0)
    H = W = 8
    ro_tr = ro.reshape(3, H, W, 3); rd_tr = rd.reshape(3, H, W, 3)
    cnt = m.voxel_count_views(rays_o_tr=ro_tr, rays_d_tr=rd_tr, imsz=[1] * 3, near=0.5, far=6.0,
                              stepsize=0.5, downrate=1)
    save('voxel_count_views', xyz_min=mn, xyz_max=mx, world_size=m.world_size.numpy(),
         voxel_size=m.voxel_size.numpy(), rays_o=ro_tr, rays_d=rd_tr, near=0.5, far=6.0, stepsize=0.5,
         count=cnt)


def gen_checkpoint(R):
    """N5: a checkpoint written exactly as run.py:430-437 writes it -- torch.save of {'global_step', 'model_kwargs':
    model.get_kwargs() (numpy bbox, numpy act_shift, tensor voxel_size_ratio, the fork's extra flags),
    'model_state_dict', 'optimizer_state_dict'} -- from the imported reference DirectVoxGO and MaskedAdam (param groups
    as lib/utils.py:20-48 builds them) after one optimisation step, so that the Adam moments exist, in the
    reference's contiguous [1,C,X,Y,Z] layout.  `ref_checkpoint_next.npz` holds the batch and the reference's
    parameters after ONE MORE step from that state: what a resumed run must reproduce."""
    rng = np.random.default_rng(911)
    m, mn, mx = _scene(R, rng, fine=True, nvox=10 ** 3, width=32)
    ro, rd, vd = lego_like_rays(R, rng, n_views=4, H=6, W=6, focal=6 * 1111.11 / 800 * 3.0, radius=3.0)
    target = torch.from_numpy(rng.random((ro.shape[0], 3)).astype(np.float32))
    rk = dict(near=0.5, far=6.0, bg=1, stepsize=0.5, inverse_y=False, flip_x=False, flip_y=False)
    groups = [{'params': m.density, 'lr': 0.1, 'skip_zero_grad': True},
              {'params': m.k0, 'lr': 0.1, 'skip_zero_grad': True},
              {'params': m.rgbnet.parameters(), 'lr': 1e-3, 'skip_zero_grad': False}]
    opt = R.masked_adam.MaskedAdam(groups)
    decay = 0.1 ** (1 / 20000)

    def one_step(step):
        res = m(ro, rd, vd, global_step=step, **rk)
        opt.zero_grad(set_to_none=True)
        _loss(res, target, ro.shape[0]).backward()
        opt.step()
        for g in opt.param_groups:
            g['lr'] = g['lr'] * decay

    one_step(1)
    path = os.path.join(HERE, 'ref_checkpoint.tar')
    torch.save({'global_step': 1, 'model_kwargs': m.get_kwargs(), 'model_state_dict': m.state_dict(),
                'optimizer_state_dict': opt.state_dict()}, path)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')
    one_step(2)
    out = {'rays_o': ro, 'rays_d': rd, 'viewdirs': vd, 'target': target, 'density': m.density, 'k0': m.k0}
    for k, p in m.rgbnet.named_parameters():
        out['rgbnet_' + k] = p
    st = opt.state[m.k0]
    out['k0_exp_avg'] = st['exp_avg']; out['k0_step'] = st['step']
    save('ref_checkpoint_next', **{k: v for k, v in out.items()}, **{kk: np.asarray(vv) for kk, vv in rk.items() if kk in ('near', 'far', 'stepsize')})


def gen_masked_adam(R):
    """N1 host logic: reference MaskedAdam.step (lib/masked_adam.py:39-71) for the three dispatch
    branches, 3 steps each; natives = oracle."""
    rng = np.random.default_rng(708)
    out = {}
    for tag, skip, perlr in [('plain', False, False), ('masked', True, False), ('perlr', False, True)]:
        p = torch.nn.Parameter(torch.from_numpy(rng.standard_normal((1, 2, 5, 6, 7)).astype(np.float32)))
        out[f'{tag}_p0'] = p.detach().clone()
        opt = R.masked_adam.MaskedAdam([{'params': [p], 'lr': 0.1, 'skip_zero_grad': skip}])
        if perlr:
            cnt = torch.from_numpy(rng.integers(0, 5, p.shape).astype(np.float32))
            opt.set_pervoxel_lr(cnt)
            out[f'{tag}_count'] = cnt
        for s in range(3):
            g = rng.standard_normal(p.shape).astype(np.float32)
            g[rng.random(p.shape) < 0.5] = 0
            p.grad = torch.from_numpy(g)
            opt.step()
            out[f'{tag}_g{s}'] = g
            out[f'{tag}_p{s + 1}'] = p.detach().clone()
        st = opt.state[p]
        out[f'{tag}_exp_avg'] = st['exp_avg']; out[f'{tag}_exp_avg_sq'] = st['exp_avg_sq']
    save('masked_adam', **out)


def gen_trajectory(R):
    """H3: three optimisation steps of the reference pieces -- DirectVoxGO.forward (lib/dvgo.py), the loss of
    run.py:377-386 (restated in _loss), MaskedAdam with skip_zero_grad on the grids (lib/masked_adam.py,
    lib/utils.py:20-48) and the per-step lr decay of run.py:401-406 -- natives = oracle.  Parameters after
    every step are the fixture."""
    rng = np.random.default_rng(910)
    m, mn, mx = _scene(R, rng, fine=True, nvox=12 ** 3, width=16)
    ro, rd, vd = lego_like_rays(R, rng, n_views=3, H=6, W=6, focal=6 * 1111.11 / 800 * 3.0, radius=3.0)
    N = ro.shape[0]
    target = torch.from_numpy(rng.random((N, 3)).astype(np.float32))
    rk = dict(near=0.5, far=6.0, bg=1, stepsize=0.5)
    out = dict(xyz_min=mn, xyz_max=mx, world_size=m.world_size.numpy(), density0=m.density.detach().clone(),
               k00=m.k0.detach().clone(), mask=m.mask_cache.mask, rays_o=ro, rays_d=rd, viewdirs=vd, target=target,
               near=rk['near'], far=rk['far'], stepsize=rk['stepsize'], fast_color_thres=np.float64(m.fast_color_thres))
    for k, v in m.rgbnet.state_dict().items():
        out['rgbnet0_' + k] = v.detach().clone()
    groups = [{'params': [m.density], 'lr': 0.1, 'skip_zero_grad': True}, {'params': [m.k0], 'lr': 0.1, 'skip_zero_grad': True},
              {'params': list(m.rgbnet.parameters()), 'lr': 1e-3, 'skip_zero_grad': False}]
    opt = R.masked_adam.MaskedAdam(groups)
    decay = 0.1 ** (1 / (20 * 1000))
    for step in range(1, 4):
        res = m(ro, rd, vd, global_step=step, **rk)
        opt.zero_grad(set_to_none=True)
        loss = _loss(res, target, N)
        loss.backward()
        opt.step()
        for g in opt.param_groups:
            g['lr'] = g['lr'] * decay
        out[f'loss{step}'] = loss.detach().clone()
        out[f'density{step}'] = m.density.detach().clone()
        out[f'k0{step}'] = m.k0.detach().clone()
        for k, v in m.rgbnet.state_dict().items():
            out[f'rgbnet{step}_' + k] = v.detach().clone()
    save('trajectory', **out)


def gen_scale_volume_grid(R):
    """N4: DirectVoxGO.scale_volume_grid (lib/dvgo.py:228-263) -- trilinear resize of both grids, the new occupancy mask
    from max_pool3d(activate_density) > fast_color_thres, and the `mask_cache_path` branch (AND with a coarse-stage
    MaskCache evaluated at the new grid's voxel centres).  Pure PyTorch in the reference except the mask lookup (oracle)."""
    rng = np.random.default_rng(515)
    mn, mx = np.array([-1.0, -1.1, -0.9], np.float32), np.array([1.0, 1.2, 1.1], np.float32)
    out = {'xyz_min': mn, 'xyz_max': mx}
    # a coarse-stage checkpoint for the mask_cache_path branch
    cd = (rng.standard_normal((1, 1, 9, 10, 9)) * 5).astype(np.float32)
    ck = os.path.join(HERE, 'scale_volume_coarse.tar')
    torch.save({'model_state_dict': {'density': torch.from_numpy(cd)},
                'model_kwargs': {'act_shift': -13.8155, 'voxel_size_ratio': 1.0, 'xyz_min': mn.tolist(), 'xyz_max': mx.tolist()}}, ck)
    for tag, path in (('plain', None), ('coarse', ck)):
        torch.manual_seed(777)
        m = R.dvgo.DirectVoxGO(mn, mx, num_voxels=12 ** 3, num_voxels_base=24 ** 3, alpha_init=1e-2, fast_color_thres=1e-4,
                               rgbnet_dim=4, rgbnet_depth=3, rgbnet_width=16, viewbase_pe=4, mask_cache_path=path,
                               mask_cache_thres=1e-3)
        with torch.no_grad():
            ws = tuple(int(v) for v in m.world_size)
            m.density.copy_(torch.from_numpy(blob_density(ws, mn, mx, rng, amp=22.0, bias=-15.0))[None, None])
            m.k0.copy_(torch.from_numpy((rng.standard_normal(m.k0.shape) * 0.3).astype(np.float32)))
        out[f'{tag}_density_in'] = m.density.detach().numpy().copy()
        out[f'{tag}_k0_in'] = m.k0.detach().numpy().copy()
        m.scale_volume_grid(20 ** 3)
        out[f'{tag}_world_size'] = m.world_size.numpy().copy()
        out[f'{tag}_voxel_size_ratio'] = np.float64(m.voxel_size_ratio)
        out[f'{tag}_density_out'] = m.density.detach().numpy().copy()
        out[f'{tag}_k0_out'] = m.k0.detach().numpy().copy()
        out[f'{tag}_mask_out'] = m.mask_cache.mask.numpy().copy()
    save('scale_volume_grid', **out)


def gen_rays(R):
    """Ray generator pin (lib/ray_utils.py:9-47,80-85 + lib/load_blender.py:37-42)."""
    rng = np.random.default_rng(809)
    ro, rd, vd = lego_like_rays(R, rng, n_views=2, H=4, W=6, focal=6 * 1111.11 / 800)
    rng = np.random.default_rng(809)
    thetas = [float(rng.uniform(-180, 180)) for _ in range(2)]
    save('rays', thetas=np.array(thetas), H=4, W=6, focal=6 * 1111.11 / 800, radius=4.0, phi=-30.0,
         rays_o=ro, rays_d=rd, viewdirs=vd)


def main():
    R = import_reference()
    try:
        if len(sys.argv) > 1 and sys.argv[1] == 'forward_fine_direct':      # add this one fixture only
            gen_forward(R, fine=True, name='forward_fine_direct', width=128, direct=True)
            return
        if len(sys.argv) > 1 and sys.argv[1] == 'checkpoint':
            gen_checkpoint(R)
            return
        if len(sys.argv) > 1 and sys.argv[1] == 'scale_volume_grid':
            gen_scale_volume_grid(R)
            return
        if len(sys.argv) > 1 and sys.argv[1] == 'forward_mpi_w64':
            gen_mpi_forward(R, width=64, name='forward_mpi_w64')
            return
        gen_constants(R)
        gen_grid_sampler(R)
        gen_sampler_py(R)
        gen_maskcache_path(R)
        gen_forward(R, fine=True, name='forward_fine')
        gen_forward(R, fine=False, name='forward_coarse')
        gen_forward(R, fine=True, name='forward_fine_direct', width=128, direct=True)
        gen_mpi_forward(R)
        gen_mpi_forward(R, width=64, name='forward_mpi_w64')
        gen_voxel_count_views(R)
        gen_masked_adam(R)
        gen_checkpoint(R)
        gen_trajectory(R)
        gen_rays(R)
        gen_scale_volume_grid(R)
    finally:
        shutil.rmtree(R.scratch, ignore_errors=True)


if __name__ == '__main__':
    main()
