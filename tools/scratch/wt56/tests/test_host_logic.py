"""CPU: host-side mirror of the reference interface -- grid sizing, constants, state_dict layout,
optimizer host maths, synthetic inputs -- against values produced by the reference's own code
(tests/golden/constants.npz, rays.npz)."""
import numpy as np
import pytest
import torch

from conftest import load_golden


@pytest.mark.parametrize('tag', ['coarse100', 'fine160', 'pg63', 'aniso'])
def test_grid_sizing_and_constants_match_reference(tag):
    from directvoxgo_amd.dvgo import DirectVoxGO
    c = load_golden('constants')
    nv = int(c[f'{tag}_num_voxels'])
    if nv > 2_000_000:
        # avoid allocating 160^3 x 12 floats on the CPU: size-only check through the same code path
        m = DirectVoxGO.__new__(DirectVoxGO)
        torch.nn.Module.__init__(m)
        m.verbose = False
        m._xyz_min_cpu = torch.from_numpy(c[f'{tag}_xyz_min']); m._xyz_max_cpu = torch.from_numpy(c[f'{tag}_xyz_max'])
        m.num_voxels_base = int(c[f'{tag}_num_voxels_base'])
        m.voxel_size_base = ((m._xyz_max_cpu - m._xyz_min_cpu).prod() / m.num_voxels_base).pow(1 / 3)
        m._set_grid_resolution(nv)
        act_shift = np.log(1 / (1 - float(c[f'{tag}_alpha_init'])) - 1)
    else:
        m = DirectVoxGO(c[f'{tag}_xyz_min'], c[f'{tag}_xyz_max'], num_voxels=nv,
                        num_voxels_base=int(c[f'{tag}_num_voxels_base']), alpha_init=float(c[f'{tag}_alpha_init']),
                        fast_color_thres=1e-4, rgbnet_dim=0)
        act_shift = m.act_shift
        np.testing.assert_array_equal(m.mask_cache.xyz2ijk_scale.numpy(), c[f'{tag}_xyz2ijk_scale'])
        np.testing.assert_array_equal(m.mask_cache.xyz2ijk_shift.numpy(), c[f'{tag}_xyz2ijk_shift'])
    assert m.world_size.tolist() == c[f'{tag}_world_size'].tolist()
    assert float(m.voxel_size) == float(c[f'{tag}_voxel_size'])
    assert float(m.voxel_size_ratio) == float(c[f'{tag}_voxel_size_ratio'])
    assert act_shift == float(c[f'{tag}_act_shift'])


def test_state_dict_keys_and_layout_round_trip():
    """same keys / logical shapes as the reference model (lib/dvgo.py:44-45,68,94,96,123-131,599-602)
    while the feature grid is physically channels-last."""
    from directvoxgo_amd.dvgo import DirectVoxGO
    m = DirectVoxGO([-1, -1, -1], [1, 1, 1], num_voxels=10 ** 3, num_voxels_base=10 ** 3, alpha_init=1e-2,
                    fast_color_thres=1e-4, rgbnet_dim=12, rgbnet_depth=3, rgbnet_width=16, viewbase_pe=4)
    sd = m.state_dict()
    expect = {'xyz_min', 'xyz_max', 'density', 'k0', 'viewfreq', 'rgbnet.0.weight', 'rgbnet.0.bias',
              'rgbnet.2.0.weight', 'rgbnet.2.0.bias', 'rgbnet.3.weight', 'rgbnet.3.bias', 'mask_cache.mask',
              'mask_cache.xyz2ijk_scale', 'mask_cache.xyz2ijk_shift'}
    assert set(sd.keys()) == expect
    assert tuple(sd['density'].shape) == (1, 1, 10, 10, 10) and tuple(sd['k0'].shape) == (1, 12, 10, 10, 10)
    assert m.k0.stride()[1] == 1 and m.rgbnet[0].in_features == 27 + 9
    m2 = DirectVoxGO(**{**m.get_kwargs(), 'rgbnet_width': 16})
    sd['k0'] = torch.randn(1, 12, 10, 10, 10)        # a reference-layout (contiguous) checkpoint tensor
    m2.load_state_dict(sd)
    assert m2.k0.stride()[1] == 1 and torch.equal(m2.k0.detach(), sd['k0'])
    with pytest.raises(NotImplementedError):
        DirectVoxGO([-1] * 3, [1] * 3, num_voxels=8, num_voxels_base=8, alpha_init=1e-2, implicit_voxel_feat=True)


def test_adam_step_size_is_float32_host_math():
    """lib/cuda/adam_upd_kernel.cu:72 evaluates the bias correction with float overloads"""
    from directvoxgo_amd.masked_adam import adam_step_size
    for step in (1, 2, 10, 1000):
        f = np.float32
        ref = f(0.1) * np.sqrt(f(1) - np.power(f(0.99), f(step))) / (f(1) - np.power(f(0.9), f(step)))
        assert adam_step_size(0.1, 0.9, 0.99, step) == float(ref)


def test_synthetic_rays_follow_reference_conventions():
    """scenes.pose_spherical / camera_rays against the reference run (lib/load_blender.py:37-42,
    lib/ray_utils.py get_rays_of_a_view) stored in tests/golden/rays.npz."""
    from directvoxgo_amd.scenes import camera_rays, pose_spherical
    g = load_golden('rays')
    H, W, focal = int(g['H']), int(g['W']), float(g['focal'])
    ro, rd, vd = [], [], []
    for th in g['thetas']:
        o, d, v = camera_rays(H, W, focal, pose_spherical(float(th), float(g['phi']), float(g['radius'])))
        ro.append(o); rd.append(d); vd.append(v)
    np.testing.assert_allclose(torch.cat(ro).numpy(), g['rays_o'], atol=1e-6)
    np.testing.assert_allclose(torch.cat(rd).numpy(), g['rays_d'], atol=1e-6)
    np.testing.assert_allclose(torch.cat(vd).numpy(), g['viewdirs'], atol=1e-6)


def test_roofline_scene_yields_exactly_256_kept_samples_per_ray(oracle):
    """SURVEY 8d roofline case, checked with the oracle on a subsample: N_steps == 256 for every ray,
    all samples in the box, none culled by the alpha / weight thresholds, no early stop."""
    from directvoxgo_amd.scenes import roofline_scene
    sc = roofline_scene(world=160, n_rays=64, seed=777)
    mn, mx = sc['xyz_min'].numpy(), sc['xyz_max'].numpy()
    voxel_size = np.float32(((mx - mn).prod() / 160 ** 3) ** (1 / 3))
    stepdist = np.float32(0.5) * voxel_size
    pts, mo, rid, sid, n_steps, t_min, t_max = oracle.sample_pts_on_rays(
        sc['rays_o'].numpy(), sc['rays_d'].numpy(), mn, mx, sc['near'], sc['far'], stepdist)
    assert (n_steps == 256).all() and not mo.any()
    dens = oracle.grid_sample_fwd(sc['density'][0].numpy(), pts, mn, mx)[:, 0]
    _, alpha = oracle.raw2alpha(dens, np.log(1 / (1 - 1e-2) - 1), 0.5)
    assert (alpha > 1e-4).all()
    w, T, last, i_s, i_e = oracle.alpha2weight(alpha, rid, 64)
    assert (w > 1e-4).all() and (last > 1e-3).all() and ((i_e - i_s) == 256).all()


def test_rec_stride_bound():
    from directvoxgo_amd.fused import MarchConfig, _rec_stride
    cfg = MarchConfig(torch.tensor([-1., -1, -1]), torch.tensor([1., 1, 1]), stepdist=0.009375, act_shift=0.0,
                      interval=0.5, fast_color_thres=1e-4, near=2.0, far=6.0)
    assert _rec_stride(cfg, 8192) == 429            # ceil(4 / 0.009375) + 2
    cfg.far = 1e9
    assert _rec_stride(cfg, 8192) == 0              # falls back to the exact (cumsum) layout


@pytest.mark.skipif(not __import__('os').path.isdir('/root/reference/lib'), reason='reference tree only exists in the build container')
def test_compat_shim_binds_reference_modules_by_name(tmp_path):
    """directvoxgo_amd.compat.install(): an unmodified copy of the reference's lib/dvgo.py and
    lib/masked_adam.py picks up the HIP op modules where it calls cpp_extension.load(name=...)."""
    import shutil
    import subprocess
    import sys
    shutil.copytree('/root/reference/lib', tmp_path / 'lib', ignore=shutil.ignore_patterns('cuda', '__pycache__'))
    (tmp_path / 'lib' / '__init__.py').touch()
    code = (
        "import sys; sys.dont_write_bytecode = True\n"
        f"sys.path.insert(0, {str(tmp_path)!r}); sys.path.insert(0, {str(__import__('conftest').REPO)!r})\n"
        "import directvoxgo_amd.compat as compat; compat.install()\n"
        "import lib.dvgo as d, lib.masked_adam as ma\n"
        "import directvoxgo_amd.render_utils as ru, directvoxgo_amd.ops as ops\n"
        "assert d.render_utils_cuda is ru\n"
        "assert d.segment_coo is ops.segment_coo\n"
        "assert d.total_variation_cuda.total_variation_add_grad is ops.total_variation_add_grad\n"
        "assert callable(ma.adam_upd_cuda.masked_adam_upd)\n"
        "m = d.DirectVoxGO([-1,-1,-1],[1,1,1], num_voxels=512, num_voxels_base=512, alpha_init=1e-2, fast_color_thres=1e-4)\n"
        "print('ok', tuple(m.density.shape))\n")
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'ok (1, 1, 8, 8, 8)' in out.stdout


def test_ray_generator_matches_reference_run():
    """render.get_rays_of_a_view vs the reference's lib/ray_utils.py output (tests/golden/rays.npz)."""
    from directvoxgo_amd.render import get_rays_of_a_view
    from directvoxgo_amd.scenes import pose_spherical
    g = load_golden('rays')
    H, W, focal = int(g['H']), int(g['W']), float(g['focal'])
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], np.float32)
    outs = [get_rays_of_a_view(H, W, K, pose_spherical(float(th), float(g['phi']), float(g['radius'])), ndc=False,
                               inverse_y=False, flip_x=False, flip_y=False) for th in g['thetas']]
    for k, name in enumerate(['rays_o', 'rays_d', 'viewdirs']):
        got = torch.cat([o[k].reshape(-1, 3) for o in outs]).numpy()
        np.testing.assert_allclose(got, g[name], atol=1e-6)


def test_checkpoint_round_trip_in_reference_format(tmp_path):
    """run.py:420-437 dict layout; grids stored contiguous; MaskCache(path=...) can read it."""
    from directvoxgo_amd.checkpoint import load_model, save_checkpoint
    from directvoxgo_amd.dvgo import DirectVoxGO
    from directvoxgo_amd.ops import MaskCache
    torch.manual_seed(0)
    m = DirectVoxGO([-1, -1, -1], [1, 1, 1], num_voxels=9 ** 3, num_voxels_base=9 ** 3, alpha_init=1e-2,
                    fast_color_thres=1e-4, rgbnet_dim=12, rgbnet_width=16)
    with torch.no_grad():
        m.density.normal_(); m.k0.normal_()
    p = str(tmp_path / 'fine_last.tar')
    save_checkpoint(p, m, None, 123)
    from directvoxgo_amd.checkpoint import safe_load
    ck = safe_load(p)
    assert set(ck) == {'global_step', 'model_kwargs', 'model_state_dict', 'optimizer_state_dict'}
    assert ck['model_state_dict']['k0'].is_contiguous() and ck['global_step'] == 123
    m2 = load_model(DirectVoxGO, p)
    assert m2.k0.stride()[1] == 1
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    mc = MaskCache(path=p, mask_cache_thres=1e-3)
    assert mc.mask.shape == (9, 9, 9)


def test_reference_written_checkpoint_loads_without_executing_anything():
    """tests/golden/ref_checkpoint.tar was written by the imported reference (tests/golden/make_golden.py
    gen_checkpoint: run.py:430-437 dict, DirectVoxGO.get_kwargs() with numpy values, contiguous grids and Adam moments).
    It must load through the weights-only loader (only numpy's array / scalar reconstructors allow-listed), rebuild the
    model (lib/utils.py:63-79) and feed MaskCache(path=...) (lib/dvgo.py:586-593)."""
    import os
    from conftest import GOLDEN
    from directvoxgo_amd.checkpoint import load_model, model_kwargs_of, safe_load
    from directvoxgo_amd.dvgo import DirectVoxGO
    from directvoxgo_amd.ops import MaskCache
    path = os.path.join(GOLDEN, 'ref_checkpoint.tar')
    ck = safe_load(path)
    assert set(ck) == {'global_step', 'model_kwargs', 'model_state_dict', 'optimizer_state_dict'}
    kw = ck['model_kwargs']
    assert isinstance(kw['xyz_min'], np.ndarray) and isinstance(kw['act_shift'], np.floating)      # as the reference writes them
    sd = ck['model_state_dict']
    assert sd['k0'].is_contiguous() and sd['k0'].shape[1] == 12 and sd['density'].dim() == 5
    ost = ck['optimizer_state_dict']
    assert ost['state'][1]['exp_avg'].is_contiguous() and ost['state'][1]['exp_avg'].shape == sd['k0'].shape
    assert [g['skip_zero_grad'] for g in ost['param_groups']] == [True, True, False]
    m = load_model(DirectVoxGO, path)
    assert m.k0.stride()[1] == 1 and torch.equal(m.k0.detach().contiguous(), sd['k0'])            # channels-last in memory
    assert set(m.state_dict()) == set(sd), set(m.state_dict()) ^ set(sd)
    np.testing.assert_allclose(m.act_shift, float(kw['act_shift']), rtol=1e-12)
    np.testing.assert_allclose(float(m.voxel_size_ratio), float(kw['voxel_size_ratio']), rtol=1e-6)
    assert 'act_shift' not in model_kwargs_of(ck)
    mc = MaskCache(path=path, mask_cache_thres=1e-3)
    assert tuple(mc.mask.shape) == tuple(sd['density'].shape[2:]) and 0 < float(mc.mask.float().mean()) < 1


def test_safe_load_refuses_a_pickle_that_would_run_code(tmp_path):
    import os
    import pickle
    from directvoxgo_amd.checkpoint import safe_load
    from directvoxgo_amd.ops import MaskCache

    class Evil:
        def __reduce__(self):
            return (os.system, ('echo pwned > ' + str(tmp_path / 'pwned'),))

    p = str(tmp_path / 'evil.tar')
    torch.save({'global_step': 1, 'model_kwargs': Evil(), 'model_state_dict': {}, 'optimizer_state_dict': None}, p)
    with pytest.raises(pickle.UnpicklingError):
        safe_load(p)
    with pytest.raises(pickle.UnpicklingError):
        MaskCache(path=p, mask_cache_thres=1e-3)
    assert not (tmp_path / 'pwned').exists()


def test_optimizer_state_is_saved_in_the_reference_layout_and_relaid_on_resume(tmp_path):
    """the Adam moments of the channels-last feature grid are written contiguous [1,C,X,Y,Z] (what the
    reference's adam_upd reads as raw memory) and come back in the parameter's strides."""
    from directvoxgo_amd.checkpoint import load_checkpoint, safe_load, save_checkpoint
    from directvoxgo_amd.dvgo import DirectVoxGO
    from directvoxgo_amd.masked_adam import MaskedAdam
    torch.manual_seed(0)
    m = DirectVoxGO([-1, -1, -1], [1, 1, 1], num_voxels=7 ** 3, num_voxels_base=7 ** 3, alpha_init=1e-2, rgbnet_dim=12, rgbnet_width=16)
    opt = MaskedAdam([{'params': [m.density], 'lr': 0.1, 'skip_zero_grad': True}, {'params': [m.k0], 'lr': 0.1, 'skip_zero_grad': True}])
    st = opt._state_of(m.k0)
    st['step'] = 4
    st['exp_avg'].copy_(torch.randn(m.k0.shape)); st['exp_avg_sq'].copy_(torch.rand(m.k0.shape))
    assert st['exp_avg'].stride() == m.k0.stride() and not st['exp_avg'].is_contiguous()
    opt._state_of(m.density)
    p = str(tmp_path / 'x.tar')
    save_checkpoint(p, m, opt, 4)
    on_disk = safe_load(p)['optimizer_state_dict']['state'][1]
    assert on_disk['exp_avg'].is_contiguous() and torch.equal(on_disk['exp_avg'], st['exp_avg'].contiguous())
    m2 = DirectVoxGO([-1, -1, -1], [1, 1, 1], num_voxels=7 ** 3, num_voxels_base=7 ** 3, alpha_init=1e-2, rgbnet_dim=12, rgbnet_width=16)
    opt2 = MaskedAdam([{'params': [m2.density], 'lr': 0.1, 'skip_zero_grad': True}, {'params': [m2.k0], 'lr': 0.1, 'skip_zero_grad': True}])
    load_checkpoint(m2, opt2, p)
    st2 = opt2._state_of(m2.k0)
    assert st2['step'] == 4 and st2['exp_avg'].stride() == m2.k0.stride() and torch.equal(st2['exp_avg'], st['exp_avg'])
