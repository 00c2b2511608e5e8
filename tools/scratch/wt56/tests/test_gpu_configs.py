"""GPU: the BASELINE configs that the fixtures only cover at toy size, at their REAL shapes.

  config 5  Tanks&Temples Truck, 256^3 fine grid, 800x800 full-image render (run.py:57-143,
            configs/tankstemple/Truck.py:9 `inverse_y`, `render_depth=True`, no_grad), 65536-ray chunks:
            1 GB feature grid, 16.7 M-voxel occupancy keys, the multi-launch scans.
  config 4  LLFF fern, DirectMPIGO (lib/dmpigo.py:97-107,173-283; configs/llff/llff_default.py:16-34): 343 x 382 x 128
            grid from num_voxels = 256^3 / mpi_depth = 128, 9 features, 64-wide head, 4096 rays x 255 NDC samples,
            dense total variation on both grids in the step.

No oracle run is affordable at these sizes, so each test holds the fused path against the op-by-op HIP path
(`fused=False`: every kernel of it is pinned bit-exact / to tolerance against the oracle in test_gpu_ops.py) on the
same inputs, plus size-independent properties (sum w + T = 1, depth range, the inverse_y / pose-flip identity).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _truck_like_model(world=256):
    from directvoxgo_amd.dvgo import DirectVoxGO
    from directvoxgo_amd.scenes import synthetic_scene
    sc = synthetic_scene(world=world, n_rays=8, seed=11, device='cpu')
    torch.manual_seed(0)
    m = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=world ** 3, num_voxels_base=world ** 3, alpha_init=1e-2,
                    fast_color_thres=1e-4, rgbnet_dim=12, rgbnet_width=128, rgbnet_direct=True)
    assert m.world_size.tolist() == [world] * 3
    with torch.no_grad():
        m.density.copy_(sc['density']); m.k0.copy_(sc['k0']); m.mask_cache.mask.copy_(sc['mask'])
    return sc, m.cuda()


def test_config5_full_image_render_256_cubed_inverse_y():
    from directvoxgo_amd.render import get_rays_of_a_view, render_viewpoints
    from directvoxgo_amd.scenes import pose_spherical
    sc, m = _truck_like_model(256)
    assert m.k0.numel() * 4 > 800e6 and m.k0.stride()[1] == 1               # 805 MB channels-last feature grid
    H = W = 800
    K = np.array([[1111.11, 0, 0.5 * W], [0, 1111.11, 0.5 * H], [0, 0, 1]], np.float32)
    pose_gl = pose_spherical(35.0, -30.0, 4.0)                               # blender / OpenGL camera
    pose_cv = pose_gl.clone()
    pose_cv[:3, 1] *= -1; pose_cv[:3, 2] *= -1                               # the same camera in the OpenCV convention
    # inward_nearfar_heuristic(ratio=0) (lib/load_data.py:221-225): near = 0, far = the largest camera distance
    rk = dict(near=0.0, far=8.0, bg=1, stepsize=0.5, inverse_y=True, flip_x=False, flip_y=False)
    rgbs, depths = render_viewpoints(m, [pose_cv.numpy()], [(H, W)], [K], False, rk, chunk=65536)
    assert rgbs.shape == (1, H, W, 3) and depths.shape == (1, H, W, 1)
    assert np.isfinite(rgbs).all() and 0.0 <= rgbs.min() and rgbs.max() <= 1.0 + 1e-5
    n_steps_max = 8.0 / (0.5 * float(m.voxel_size)) + 2
    assert 0.0 <= depths.min() and depths.max() < n_steps_max
    assert (rgbs < 0.99).mean() > 0.05 and (depths > 0).mean() > 0.05       # the object is in view
    # inverse_y with the OpenCV pose generates the same rays as the OpenGL pose without it (lib/ray_utils.py:28-35):
    rk_gl = dict(rk, inverse_y=False)
    rgbs_gl, depths_gl = render_viewpoints(m, [pose_gl.numpy()], [(H, W)], [K], False, rk_gl, chunk=8192)   # run.py's own chunk
    np.testing.assert_allclose(rgbs, rgbs_gl, atol=1e-6)                   # chunking and convention do not change a pixel
    np.testing.assert_allclose(depths, depths_gl, atol=1e-4)

    # a 4096-ray slice of the image through the op-by-op HIP path (reference op order on the oracle-pinned kernels)
    ro, rd, vd = (t.flatten(0, -2).contiguous() for t in
                  get_rays_of_a_view(H, W, K, pose_cv.cuda(), False, inverse_y=True, flip_x=False, flip_y=False))
    sl = slice(400 * 800 + 100, 400 * 800 + 100 + 4096)
    kw = dict(near=0.0, far=8.0, bg=1, stepsize=0.5, render_depth=True)
    with torch.no_grad():
        a = m(ro[sl], rd[sl], vd[sl], **kw)
        m.fused = False
        b = m(ro[sl], rd[sl], vd[sl], **kw)
        m.fused = True
    assert a['weights'].numel() > 10000
    assert torch.equal(a['ray_id'], b['ray_id'])                             # index outputs exact
    assert torch.equal(a['weights'], b['weights']) and torch.equal(a['alphainv_last'], b['alphainv_last'])
    assert torch.allclose(a['raw_rgb'], b['raw_rgb'], atol=2e-5)             # MFMA colour head vs torch modules
    assert torch.allclose(a['rgb_marched'], b['rgb_marched'], atol=2e-5)
    assert torch.allclose(a['depth'], b['depth'], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(rgbs[0].reshape(-1, 3)[sl], a['rgb_marched'].cpu().numpy(), atol=1e-6)
    # compositing identity on a whole 65536-ray chunk: sum of the kept weights + residual transmittance = 1, up to the
    # weights the `> fast_color_thres` filter dropped (each <= 1e-4)
    with torch.no_grad():
        c = m(ro[:65536], rd[:65536], vd[:65536], **kw)
    wsum = torch.zeros(65536, device='cuda').index_add_(0, c['ray_id'], c['weights'])
    tot = wsum + c['alphainv_last']
    assert float(tot.max()) <= 1 + 1e-4 and float(tot.min()) >= 0.95
    rid = c['ray_id']
    assert bool((rid[1:] >= rid[:-1]).all())                                 # ray-major order


def _fern_like(fused, seed=0):
    from directvoxgo_amd.dmpigo import DirectMPIGO
    torch.manual_seed(seed)
    mn, mx = np.array([-1.5, -1.67, -1.0], np.float32), np.array([1.5, 1.67, 1.0], np.float32)   # llff bbox (run.py:232-236)
    m = DirectMPIGO(mn, mx, num_voxels=256 ** 3, mpi_depth=128, fast_color_thres=1e-3, rgbnet_dim=9, rgbnet_depth=3,
                    rgbnet_width=64, viewbase_pe=0, fused=fused)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        m.density.add_(torch.randn(m.density.shape, generator=g) * 2.0)      # structure: about half of the samples pass alpha > 1e-3
        m.k0.copy_(torch.randn(m.k0.shape, generator=g) * 0.3)
    return m.cuda()


def _fern_rays(n=4096, seed=1):
    g = torch.Generator().manual_seed(seed)
    ro = torch.cat([torch.rand(n, 2, generator=g) * 2.4 - 1.2, -torch.ones(n, 1)], 1)
    rd = torch.cat([torch.rand(n, 2, generator=g) * 0.6 - 0.3, 2 * torch.ones(n, 1)], 1)
    return ro.cuda(), rd.cuda(), (rd / rd.norm(dim=-1, keepdim=True)).cuda(), torch.rand(n, 3, generator=g).cuda()


def test_config4_mpi_real_shape_fused_equals_op_by_op_and_trains_with_dense_tv():
    from directvoxgo_amd.train import FINE_TRAIN, TrainStep
    ro, rd, vd, tgt = _fern_rays()
    rk = dict(near=0, far=1, bg=0, stepsize=0.5, inverse_y=False, flip_x=False, flip_y=False)
    outs = {}
    for fused in (True, False):
        m = _fern_like(fused)
        assert m.world_size.tolist() == [343, 382, 128] and m.k0.shape[1] == 9 and m.n_samples(0.5) == 255
        res = m(ro, rd, vd, global_step=0, render_depth=True, **rk)
        loss = (res['rgb_marched'] - tgt).pow(2).mean()
        loss.backward()
        outs[fused] = (res, m.density.grad.clone(), m.k0.grad.clone())
        del m
    a, b = outs[True][0], outs[False][0]
    assert a['weights'].numel() > 4096 * 20
    assert torch.equal(a['ray_id'], b['ray_id']) and torch.equal(a['weights'], b['weights'])
    assert torch.equal(a['alphainv_last'], b['alphainv_last'])
    assert torch.allclose(a['rgb_marched'], b['rgb_marched'], atol=2e-5)
    assert torch.allclose(a['depth'], b['depth'], rtol=1e-5, atol=1e-3)
    for ga, gb in ((outs[True][1], outs[False][1]), (outs[True][2], outs[False][2])):
        assert float((ga - gb).abs().max()) <= 2e-4 * float(gb.abs().max()) + 1e-9
        assert torch.equal(ga != 0, gb != 0)
    del outs
    torch.cuda.empty_cache()

    # one optimisation step of configs/llff/llff_default.py:14-24: TV on both grids, dense while step < 10000
    cfg = dict(FINE_TRAIN, N_rand=4096, tv_before=1e9, tv_dense_before=10000, weight_tv_density=1e-5, weight_tv_k0=1e-5,
               weight_entropy_last=0.001, weight_rgbper=0.01, skip_zero_grad_fields=['density', 'k0'])
    ends = []
    for fused in (True, False):
        m = _fern_like(fused)
        d0 = m.density.detach().clone()
        step = TrainStep(m, cfg, rk)
        loss = step(ro, rd, vd, tgt, global_step=200)
        assert torch.isfinite(loss)
        moved = (m.density.detach() != d0).float().mean()
        assert float(moved) > 0.99                                           # dense TV: every voxel has a gradient
        ends.append((float(loss), m.density.detach().clone(), m.k0.detach().clone()))
        del m, step
        torch.cuda.empty_cache()
    assert abs(ends[0][0] - ends[1][0]) <= 2e-5 * abs(ends[1][0])
    for x, y in ((ends[0][1], ends[1][1]), (ends[0][2], ends[1][2])):
        d = (x - y).abs()
        # first Adam step: |update| = lr for every element whose gradient has a definite sign; rounding noise can only
        # flip near-cancelled gradients
        assert float((d > 1e-4).float().mean()) <= 2e-3 and float(d.max()) <= 0.21
