"""GPU, end to end (row H3): a student model trained with the full harness (TrainStep: fused march, colour
head, loss of run.py:377-386, MaskedAdam with skip_zero_grad, lr decay; fit_stage: progressive grid growth,
occupancy refresh) recovers images rendered from a teacher scene.  Also: BASELINE-size properties of the
fused path (160^3, 8192 rays x 256 samples)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_student_fits_teacher_scene():
    from directvoxgo_amd.dvgo import DirectVoxGO
    from directvoxgo_amd.fit import fit_stage
    from directvoxgo_amd.scenes import synthetic_scene
    from directvoxgo_amd.train import FINE_TRAIN
    sc = synthetic_scene(world=40, n_rays=40000, seed=5, device='cuda')
    kw = dict(num_voxels=40 ** 3, num_voxels_base=40 ** 3, alpha_init=1e-2, fast_color_thres=1e-4, rgbnet_dim=12,
              rgbnet_width=128)
    torch.manual_seed(3)
    teacher = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], **kw).cuda()
    with torch.no_grad():
        teacher.density.copy_(sc['density']); teacher.k0.copy_(sc['k0'])
        for p in teacher.rgbnet.parameters():
            p.mul_(3.0)
    rk = dict(near=sc['near'], far=sc['far'], bg=1, stepsize=0.5)
    with torch.no_grad():
        target = torch.cat([teacher(sc['rays_o'][i:i + 8192], sc['rays_d'][i:i + 8192], sc['viewdirs'][i:i + 8192], **rk)
                            ['rgb_marched'] for i in range(0, 40000, 8192)])
    assert target.std() > 0.05                       # the teacher scene is not trivially white
    torch.manual_seed(4)
    student = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], **kw).cuda()
    cfg = dict(FINE_TRAIN, N_rand=4096, pg_scale=[100, 200], N_iters=600)
    psnrs = fit_stage(student, sc['rays_o'], sc['rays_d'], sc['viewdirs'], target, cfg, rk, n_iters=600,
                      num_voxels_final=40 ** 3)
    assert tuple(student.density.shape[2:]) == (40, 40, 40)          # grew back to the final resolution
    first, last = np.mean(psnrs[:20]), np.mean(psnrs[-50:])
    assert np.isfinite(psnrs).all()
    assert last > first + 6.0, (first, last)          # > 6 dB better than the untrained model


def test_fused_path_full_size_properties():
    """160^3, 8192 rays x 256 samples (BASELINE config 2 roofline case): sample bookkeeping is exact and the
    compositing identities hold; gradient mass is conserved by the scatter."""
    from directvoxgo_amd.dvgo import DirectVoxGO
    from directvoxgo_amd.scenes import roofline_scene
    sc = roofline_scene(world=160, n_rays=8192, device='cuda')
    m = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=160 ** 3, num_voxels_base=160 ** 3, alpha_init=1e-2,
                    fast_color_thres=1e-4, rgbnet_dim=0).cuda()           # colour grid (k0_dim 3): rgb = sigmoid(k0)
    with torch.no_grad():
        m.density.copy_(sc['density']); m.k0.copy_(sc['k0'][:, :3])
    res = m(sc['rays_o'], sc['rays_d'], sc['viewdirs'], near=sc['near'], far=sc['far'], bg=1, stepsize=0.5,
            render_depth=True)
    M = res['weights'].numel()
    assert M == 8192 * 256                                                    # nothing culled, nothing terminated
    rid = res['ray_id']
    assert bool((rid[1:] >= rid[:-1]).all()) and int(rid[-1]) == 8191         # ray-major order
    assert torch.equal(torch.bincount(rid, minlength=8192), torch.full((8192,), 256, device='cuda'))
    wsum = torch.zeros(8192, device='cuda').index_add_(0, rid, res['weights'].detach())
    assert torch.allclose(wsum + res['alphainv_last'].detach(), torch.ones_like(wsum), atol=2e-5)   # sum w + T_last = 1
    assert float(res['rgb_marched'].min()) >= 0 and float(res['rgb_marched'].max()) <= 1 + 1e-5
    assert float(res['depth'].min()) > 0 and float(res['depth'].max()) < 256
    # backward: d/d k0 of sum(rgb_marched) -- the scatter conserves the per-sample gradient mass
    g = torch.autograd.grad(res['rgb_marched'].sum(), [m.k0, m.density])
    rgb = res['raw_rgb'].detach()
    expect = (res['weights'].detach()[:, None] * rgb * (1 - rgb)).sum(0)       # trilinear weights sum to 1 per sample
    assert torch.allclose(g[0].sum((0, 2, 3, 4)), expect, rtol=2e-3)
    assert torch.isfinite(g[1]).all() and float(g[1].abs().sum()) > 0


def test_full_size_gradient_paths_agree():
    """BASELINE configs[1] sizes (160^3, 8192 x 256 kept samples, 12 features, 128-wide rgbnet_direct head on the MFMA
    kernels): the combined 64-byte-row scatter and the one-scatter-per-grid path give the same grid gradients (same
    voxels touched: the masked Adam branches on grad != 0), and a second stream for the weight gradients changes
    nothing."""
    from directvoxgo_amd import fused as fused_mod
    from directvoxgo_amd.dvgo import DirectVoxGO
    from directvoxgo_amd.scenes import roofline_scene
    from directvoxgo_amd.shade import defer_wgrad
    from directvoxgo_amd.train import FINE_TRAIN, fused_render_loss
    sc = roofline_scene(world=160, n_rays=8192, device='cuda')
    torch.manual_seed(0)
    m = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=160 ** 3, num_voxels_base=160 ** 3, alpha_init=1e-2,
                    fast_color_thres=1e-4, rgbnet_dim=12, rgbnet_width=128, rgbnet_direct=True).cuda()
    with torch.no_grad():
        m.density.copy_(sc['density']); m.k0.copy_(sc['k0'])
    rk = dict(near=sc['near'], far=sc['far'], bg=1, stepsize=0.5)
    outs = []
    for combined, side in ((True, True), (False, False)):
        fused_mod.COMBINED_GRID_GRAD = combined
        m.zero_grad(set_to_none=True)
        res = m(sc['rays_o'], sc['rays_d'], sc['viewdirs'], **rk)
        assert res['weights'].numel() == 8192 * 256
        loss = fused_render_loss(res, sc['target'], 8192, dict(FINE_TRAIN))
        with defer_wgrad(side_stream=side) as d:
            loss.backward()
        d.flush()
        torch.cuda.synchronize()
        outs.append([float(loss)] + [p.grad.clone() for p in (m.density, m.k0, *m.rgbnet.parameters())])
    fused_mod.COMBINED_GRID_GRAD = True
    a, b = outs
    assert abs(a[0] - b[0]) <= 1e-6 * abs(b[0])              # (block partial sums meet in float atomics)
    for x, y in zip(a[1:3], b[1:3]):                       # grids: atomic summation order differs, nothing else
        assert torch.equal(x != 0, y != 0)
        assert float((x - y).abs().max()) <= 2e-5 * float(y.abs().max())
    for x, y in zip(a[3:], b[3:]):                         # colour head: same kernels; partial sums meet in float atomics
        assert float((x - y).abs().max()) <= 1e-5 * float(y.abs().max())


def _view_rays(n_views, hw, device='cuda'):
    """n_views low-resolution full views of the lego-like camera ring, view after view: what run.py's
    get_training_rays_flatten hands to the coarse stage (rays grouped per image, `imsz` rays each)."""
    from directvoxgo_amd.scenes import camera_rays, pose_spherical
    ro, rd, vd, cams = [], [], [], []
    for v in range(n_views):
        c2w = pose_spherical(360.0 * v / n_views - 180.0, -30.0, 4.0)
        o, d, u = camera_rays(hw, hw, 1111.11 * hw / 800, c2w)
        ro.append(o); rd.append(d); vd.append(u); cams.append(c2w[:3, 3])
    return (torch.cat(ro).to(device), torch.cat(rd).to(device), torch.cat(vd).to(device), [hw * hw] * n_views,
            torch.stack(cams))


def test_two_stage_flow_coarse_to_fine(tmp_path):
    """run.py:440-492 on in-memory rays: coarse stage (colour grid; view-count per-voxel learning rate and
    density = -100 where at most two views look, run.py:311-320; voxels next to the cameras masked out, run.py:251-252)
    -> checkpoint -> bbox from the coarse geometry -> fine stage seeded by mask_cache_path, trained on the rays that
    hit the coarse geometry."""
    from directvoxgo_amd.dvgo import DirectVoxGO
    from directvoxgo_amd.fit import compute_bbox_by_cam_frustrm, train_two_stage
    from directvoxgo_amd.scenes import pose_spherical, synthetic_scene
    from directvoxgo_amd.train import COARSE_TRAIN, FINE_TRAIN
    sc = synthetic_scene(world=32, n_rays=8, seed=6, device='cuda')
    rays_o, rays_d, viewdirs, imsz, cam_o = _view_rays(12, 50)
    torch.manual_seed(3)
    teacher = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=32 ** 3, num_voxels_base=32 ** 3, alpha_init=1e-2,
                          fast_color_thres=1e-4, rgbnet_dim=12, rgbnet_width=128).cuda()
    with torch.no_grad():
        teacher.density.copy_(sc['density']); teacher.k0.copy_(sc['k0'])
    rk = dict(near=sc['near'], far=sc['far'], bg=1, stepsize=0.5)
    n = rays_o.shape[0]
    with torch.no_grad():
        target = torch.cat([teacher(rays_o[i:i + 8192], rays_d[i:i + 8192], viewdirs[i:i + 8192], **rk)['rgb_marched']
                            for i in range(0, n, 8192)])
    # scene bounds from the camera frusta contain the teacher's box
    K = np.array([[1111.11, 0, 400], [0, 1111.11, 400], [0, 0, 1]], np.float32)
    lo, hi = compute_bbox_by_cam_frustrm([(800, 800)] * 4, [K] * 4, [pose_spherical(t, -30.0, 4.0).numpy() for t in (0, 90, 180, 270)],
                                         near=2.0, far=6.0)
    assert bool((lo < -1.5).all()) and bool((hi > 1.5).all())
    nv = 24 ** 3
    coarse_model = dict(num_voxels=nv, num_voxels_base=nv, alpha_init=1e-6, fast_color_thres=1e-7, rgbnet_dim=0,
                        maskout_near_cam_vox=True)
    fine_model = dict(num_voxels=32 ** 3, num_voxels_base=32 ** 3, alpha_init=1e-2, fast_color_thres=1e-4, rgbnet_dim=12,
                      rgbnet_width=128)
    ct = dict(COARSE_TRAIN, N_iters=300, N_rand=4096)
    assert ct['pervoxel_lr']                                              # configs/default.py:43: Adam mode 2 (K17) end to end
    ft = dict(FINE_TRAIN, N_iters=300, N_rand=4096, pg_scale=[100])
    with pytest.raises(ValueError):                                       # the flag is acted on, never silently dropped
        train_two_stage(DirectVoxGO, sc['xyz_min'].cpu(), sc['xyz_max'].cpu(), rays_o, rays_d, viewdirs, target, rk,
                        dict(coarse_model, maskout_near_cam_vox=False), fine_model, ct, ft, str(tmp_path))
    fine, (ps_c, ps_f) = train_two_stage(DirectVoxGO, sc['xyz_min'].cpu(), sc['xyz_max'].cpu(), rays_o, rays_d,
                                         viewdirs, target, rk, coarse_model, fine_model, ct, ft, str(tmp_path),
                                         imsz=imsz, cam_o=cam_o)
    assert np.isfinite(ps_c).all() and np.isfinite(ps_f).all()
    assert np.mean(ps_c[-30:]) > np.mean(ps_c[:10]) + 3.0               # the coarse stage learns
    assert np.mean(ps_f[-30:]) > np.mean(ps_f[:10]) + 3.0               # and so does the fine stage on top of it
    assert 0.0 < float(fine.mask_cache.mask.float().mean()) < 0.9       # occupancy seeded from the coarse checkpoint
    assert bool((fine.xyz_max.cpu() - fine.xyz_min.cpu() < sc['xyz_max'].cpu() - sc['xyz_min'].cpu() + 1e-3).all())
    # the coarse checkpoint carries the per-voxel init: voxels seen by <= 2 views were set to -100 and, with a
    # view-count learning rate of 0 or next to it, stayed there
    from directvoxgo_amd.checkpoint import load_model
    coarse = load_model(DirectVoxGO, str(tmp_path / 'coarse_last.tar'))
    assert 0.02 < float((coarse.density <= -99).float().mean()) < 0.98


def test_pervoxel_lr_reaches_adam_mode_2_through_the_training_step():
    """per_voxel_init (run.py:311-320) -> MaskedAdam dispatches the per-voxel-lr kernel (K17) for the density grid:
    voxels no view counted do not move, counted ones do."""
    from directvoxgo_amd.dvgo import DirectVoxGO
    from directvoxgo_amd.fit import per_voxel_init
    from directvoxgo_amd.scenes import synthetic_scene
    from directvoxgo_amd.train import COARSE_TRAIN, TrainStep
    sc = synthetic_scene(world=24, n_rays=8, seed=6, device='cuda', k0_dim=3)
    rays_o, rays_d, viewdirs, imsz, _ = _view_rays(6, 40)
    m = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=24 ** 3, num_voxels_base=24 ** 3, alpha_init=1e-6,
                    fast_color_thres=1e-7, rgbnet_dim=0).cuda()
    rk = dict(near=sc['near'], far=sc['far'], bg=1, stepsize=0.5)
    step = TrainStep(m, dict(COARSE_TRAIN, N_rand=4096), rk)
    cnt = per_voxel_init(m, step.optimizer, rays_o, rays_d, imsz, rk['near'], rk['far'], rk['stepsize'])
    assert step.optimizer.per_lr is not None and float(cnt.max()) == 6 and float(cnt.min()) == 0
    assert torch.all(m.density[cnt <= 2] == -100)
    d0 = m.density.detach().clone()
    sel = torch.randperm(rays_o.shape[0], device='cuda')[:4096]
    step(rays_o[sel], rays_d[sel], viewdirs[sel], torch.rand(4096, 3, device='cuda'), global_step=1)
    moved = m.density.detach() != d0
    assert not bool(moved[cnt == 0].any())                       # per-voxel lr 0: exactly unchanged (adam_upd_kernel.cu:52-57)
    assert bool(moved[cnt == cnt.max()].any())


@pytest.mark.parametrize('fused', [True, False])
def test_three_step_trajectory_matches_reference_pieces(fused):
    """H3 pin: parameters after each of three optimisation steps equal the trajectory produced by the
    reference's own DirectVoxGO.forward + MaskedAdam (+ the run.py loss) over the oracle natives."""
    from conftest import load_golden
    from directvoxgo_amd.dvgo import DirectVoxGO
    from directvoxgo_amd.train import FINE_TRAIN, TrainStep
    g = load_golden('trajectory')
    nv = int(np.prod(g['world_size']))
    m = DirectVoxGO(g['xyz_min'], g['xyz_max'], num_voxels=nv, num_voxels_base=nv, alpha_init=1e-2,
                    fast_color_thres=float(g['fast_color_thres']), rgbnet_dim=12, rgbnet_depth=3, rgbnet_width=16,
                    viewbase_pe=4, fused=fused)
    with torch.no_grad():
        m.density.copy_(torch.from_numpy(g['density0'])); m.k0.copy_(torch.from_numpy(g['k00']))
        m.mask_cache.mask.copy_(torch.from_numpy(g['mask']))
        m.rgbnet.load_state_dict({k[len('rgbnet0_'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('rgbnet0_')})
    m = m.cuda()
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rk = dict(near=float(g['near']), far=float(g['far']), bg=1, stepsize=float(g['stepsize']))
    step = TrainStep(m, dict(FINE_TRAIN), rk)
    ro, rd, vd, tgt = cu(g['rays_o']), cu(g['rays_d']), cu(g['viewdirs']), cu(g['target'])
    for s in (1, 2, 3):
        loss = step(ro, rd, vd, tgt, global_step=s)
        np.testing.assert_allclose(float(loss), float(g[f'loss{s}']), rtol=2e-5)
        # Adam normalises the update to ~lr per element, so parameters are compared with an absolute
        # tolerance well below one update (lr = 0.1 for the grids, 1e-3 for the MLP)
        np.testing.assert_allclose(m.density.detach().cpu().numpy(), g[f'density{s}'], atol=2e-3)
        np.testing.assert_allclose(m.k0.detach().cpu().numpy(), g[f'k0{s}'], atol=2e-3)
        for k, p in m.rgbnet.state_dict().items():
            np.testing.assert_allclose(p.cpu().numpy(), g[f'rgbnet{s}_' + k], atol=5e-5)
        # and the set of updated voxels (masked Adam: grad != 0) is identical
        assert np.array_equal(m.density.detach().cpu().numpy() != g['density0'], g[f'density{s}'] != g['density0'])


def _assert_same_up_to_adam_noise(x, y, name=''):
    """Two runs that differ only in the order of float atomics.  Adam divides by sqrt(v): where a gradient is the
    near-cancellation of many contributions its rounding noise is amplified to a visible update, so a handful of
    voxels may move by a fraction of lr per step while everything else agrees to rounding."""
    d = (x - y).abs()
    scale = max(float(y.abs().max()), 1e-6)
    assert float((d > 1e-4 * scale).float().mean()) <= 1e-3, name
    assert float(d.max()) <= 0.35, name                       # 3 steps x lr 0.1 (+ margin)


def _dp_worker(rank, world, port, q, mode):
    import os
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)      # RCCL needs one GPU per rank; gloo moves the same bytes
    torch.cuda.set_device(0)
    params, losses = _dp_run(rank, world, mode)
    if rank == 0:
        q.put(({k: v.cpu().numpy().copy() for k, v in params.items()}, losses))
    dist.barrier()
    dist.destroy_process_group()


def _dp_run(rank, world, mode, n_steps=3):
    from directvoxgo_amd.dvgo import DirectVoxGO
    from directvoxgo_amd.scenes import synthetic_scene
    from directvoxgo_amd.train import FINE_TRAIN, TrainStep
    sc = synthetic_scene(world=32, n_rays=2048, seed=9, device='cuda')
    torch.manual_seed(4)
    m = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=32 ** 3, num_voxels_base=32 ** 3, alpha_init=1e-2,
                    fast_color_thres=1e-4, rgbnet_dim=12, rgbnet_width=128, rgbnet_direct=True).cuda()
    with torch.no_grad():
        m.density.copy_(sc['density']); m.k0.copy_(sc['k0']); m.mask_cache.mask.copy_(sc['mask'])
    cfg = dict(FINE_TRAIN)
    if mode == 'sharded_tv':               # total variation on both grids, sparse mode: it branches on the REDUCED gradient
        cfg.update(tv_before=1e9, tv_dense_before=0, weight_tv_density=1e-4, weight_tv_k0=1e-4)
    step = TrainStep(m, cfg, dict(near=sc['near'], far=sc['far'], bg=1, stepsize=0.5),
                     touched_reduce=(mode == 'touched'), shard_grids=(mode != 'allreduce'))
    if mode == 'touched':
        step.TOUCHED_MAX = 2.0
    if world > 1 and mode in ('dense', 'sharded_tv'):
        assert step._grid_shards is not None
        ran = []
        orig = step._sharded_update
        step._sharded_update = lambda shards: (ran.append(1), orig(shards))[1]
        step._ran_sharded = ran
    n = 2048 // world
    shard = tuple(sc[k][rank * n:(rank + 1) * n] for k in ('rays_o', 'rays_d', 'viewdirs', 'target'))
    losses = [float(step(*shard, global_step=1 + s)) for s in range(n_steps)]
    torch.cuda.synchronize()
    if hasattr(step, '_ran_sharded'):
        assert len(step._ran_sharded) == n_steps        # reduce-scatter -> slab TV + Adam -> all-gather really ran
    return {k: v.detach().clone() for k, v in m.state_dict().items() if v.is_floating_point()}, losses


@pytest.mark.timeout(300)
@pytest.mark.parametrize('mode', ['dense', 'allreduce', 'touched', 'sharded_tv'])
def test_two_ranks_on_one_gpu_equal_one_process(mode):
    """The product path under data parallelism: two ranks (gloo, both on this GPU), each marching half
    of the batch with the HIP kernels, reach the parameters of one process on the whole batch -- `dense`: reduce-scatter of
    the grid gradients, Adam on the owned X-slab, all-gather of the parameters; `sharded_tv`: the same with the sparse
    total-variation gradient added per slab; `allreduce`: the unsharded fallback; `touched`: the compacted
    touched-voxel reduction."""
    import socket
    import torch.multiprocessing as mp
    ref_params, ref_losses = _dp_run(0, 1, mode)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    params, losses = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # rank 0 reports its share of the global loss; both shares add up to the single-process loss only in sum, so compare
    # parameters (the thing that must match) and check the loss is finite
    assert np.isfinite(losses).all()
    for k, v in ref_params.items():
        _assert_same_up_to_adam_noise(torch.from_numpy(params[k]), v.cpu(), k)


def test_adam_from_gradient_rows_equals_dense_path():
    """TrainStep(rows_adam=True) -- MaskedAdam reading the combined gradient rows of the fused backward -- against the
    split-into-dense-gradients path: same parameters and optimizer state after three steps (up to float-atomic order),
    same set of voxels touched (the masked rule), and the two grids' .grad stay None."""
    from directvoxgo_amd import fused as fused_mod
    from directvoxgo_amd.dvgo import DirectVoxGO
    from directvoxgo_amd.scenes import synthetic_scene
    from directvoxgo_amd.train import FINE_TRAIN, TrainStep
    sc = synthetic_scene(world=48, n_rays=4096, seed=12, device='cuda')
    outs = []
    saved = fused_mod.COMBINED_MIN_RATIO
    fused_mod.COMBINED_MIN_RATIO = 1e9           # the combined-rows backward on this small scene
    fused_mod.BRICK_SCATTER = False              # (the default brick scatter fuses Adam itself: test_gpu_brick.py)
    try:
        for rows in (True, False):
            torch.manual_seed(4)
            m = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=48 ** 3, num_voxels_base=48 ** 3, alpha_init=1e-2,
                            fast_color_thres=1e-4, rgbnet_dim=12, rgbnet_width=128, rgbnet_direct=True).cuda()
            with torch.no_grad():
                m.density.copy_(sc['density']); m.k0.copy_(sc['k0']); m.mask_cache.mask.copy_(sc['mask'])
            step = TrainStep(m, dict(FINE_TRAIN), dict(near=sc['near'], far=sc['far'], bg=1, stepsize=0.5), rows_adam=rows)
            for s in range(3):
                step(sc['rays_o'], sc['rays_d'], sc['viewdirs'], sc['target'], global_step=s)
            assert (m.k0.grad is None and m.density.grad is None) == rows
            st = step.optimizer.state
            outs.append([m.density.detach().clone(), m.k0.detach().clone(), st[m.k0]['exp_avg'].clone(),
                         st[m.k0]['exp_avg_sq'].clone(), st[m.density]['exp_avg'].clone(), st[m.k0]['step'], st[m.density]['step']])
    finally:
        fused_mod.COMBINED_MIN_RATIO = saved
        fused_mod.BRICK_SCATTER = True
    a, b = outs
    assert a[5] == b[5] == 3 and a[6] == b[6] == 3
    assert torch.equal(a[2] != 0, b[2] != 0)                           # same voxels ever touched
    for x, y in zip(a[:5], b[:5]):
        _assert_same_up_to_adam_noise(x, y)


@pytest.mark.parametrize('fused', [True, False])
def test_resume_from_a_reference_written_checkpoint(fused):
    """N5 end to end: tests/golden/ref_checkpoint.tar (written by the imported reference after one optimisation step:
    contiguous grids and Adam moments, numpy kwargs) -> load_model + load_checkpoint -> ONE TrainStep on the fixture's
    batch must land on the parameters the reference reached with its own second step (ref_checkpoint_next.npz)."""
    import os
    from conftest import GOLDEN, load_golden
    from directvoxgo_amd.checkpoint import load_checkpoint, load_model
    from directvoxgo_amd.dvgo import DirectVoxGO
    from directvoxgo_amd.train import FINE_TRAIN, TrainStep, create_optimizer_or_freeze_model
    path = os.path.join(GOLDEN, 'ref_checkpoint.tar')
    g = load_golden('ref_checkpoint_next')
    m = load_model(DirectVoxGO, path, fused=fused).cuda()
    cfg = dict(FINE_TRAIN)
    opt = create_optimizer_or_freeze_model(m, cfg, global_step=0)
    _, _, start = load_checkpoint(m, opt, path)
    assert start == 1 and opt.state[m.k0]['step'] == 1
    assert abs(opt.param_groups[0]['lr'] - 0.1 * 0.1 ** (1 / 20000)) < 1e-9          # the decayed lr travels in the file
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rk = dict(near=float(g['near']), far=float(g['far']), bg=1, stepsize=float(g['stepsize']))
    step = TrainStep(m, cfg, rk, optimizer=opt)
    step(cu(g['rays_o']), cu(g['rays_d']), cu(g['viewdirs']), cu(g['target']), global_step=2)
    st = opt.state[m.k0]
    assert st['step'] == int(g['k0_step']) == 2 and st['exp_avg'].stride() == m.k0.stride()
    np.testing.assert_allclose(st['exp_avg'].cpu().numpy(), g['k0_exp_avg'], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(m.density.detach().cpu().numpy(), g['density'], atol=2e-3)
    np.testing.assert_allclose(m.k0.detach().cpu().numpy(), g['k0'], atol=2e-3)
    for k, p in m.rgbnet.named_parameters():
        np.testing.assert_allclose(p.detach().cpu().numpy(), g['rgbnet_' + k], atol=5e-5)


def test_captured_step_replays_to_the_same_parameters_as_eager_steps():
    """TrainStep.capture(): the whole step as one HIP graph (sample count on the device, Adam step sizes in device
    memory).  Five replayed steps on changing batches land on the parameters of five eager steps."""
    from directvoxgo_amd.dvgo import DirectVoxGO
    from directvoxgo_amd.scenes import lego_like_rays, synthetic_scene
    from directvoxgo_amd.train import FINE_TRAIN, TrainStep
    sc = synthetic_scene(world=48, n_rays=2048, seed=21, device='cuda')
    batches = []
    for b in range(8):
        gen = torch.Generator().manual_seed(100 + b)
        ro, rd, vd = lego_like_rays(2048, gen)
        batches.append(tuple(t.cuda() for t in (ro, rd, vd, torch.rand(2048, 3, generator=gen))))
    outs = []
    for graph in (True, False):
        torch.manual_seed(4)
        m = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=48 ** 3, num_voxels_base=48 ** 3, alpha_init=1e-2,
                        fast_color_thres=1e-4, rgbnet_dim=12, rgbnet_width=128, rgbnet_direct=True).cuda()
        with torch.no_grad():
            m.density.copy_(sc['density']); m.k0.copy_(sc['k0']); m.mask_cache.mask.copy_(sc['mask'])
        step = TrainStep(m, dict(FINE_TRAIN), dict(near=sc['near'], far=sc['far'], bg=1, stepsize=0.5))
        losses = []
        for s in range(3):                                          # warm-up steps: eager in both runs
            losses.append(float(step(*batches[s], global_step=1 + s)))
        if graph:
            assert step.can_capture()
            before = [p.detach().clone() for p in m.parameters()]
            steps_before = {id(p): st['step'] for p, st in step.optimizer.state.items()}
            assert step.capture(*batches[3], global_step=4, warmup=0)
            for p, q in zip(m.parameters(), before):               # capturing runs no kernel and counts no step
                assert torch.equal(p.detach(), q)
            assert steps_before == {id(p): st['step'] for p, st in step.optimizer.state.items()}
        for s in range(3, 8):
            losses.append(float(step(*batches[s], global_step=1 + s)))
        torch.cuda.synchronize()
        assert step.optimizer.state[m.k0]['step'] == 8
        outs.append((losses, [p.detach().clone() for p in m.parameters()], step.optimizer.param_groups[0]['lr']))
    (la, pa, lra), (lb, pb, lrb) = outs
    assert abs(lra - lrb) < 1e-12
    np.testing.assert_allclose(la, lb, rtol=2e-4)
    for x, y in zip(pa, pb):
        _assert_same_up_to_adam_noise(x, y)
