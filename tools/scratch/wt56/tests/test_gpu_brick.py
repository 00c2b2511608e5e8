"""GPU: the owner-computes brick scatter (csrc/brick.hip) against the atomic scatters and the CPU oracle.

What is held against what:
  * oracle (oracle/dvgo_oracle.c grid_sample_bwd = the reference's grid_sampler_3d_backward sums, scalar C) on
    seeded scenes the oracle finishes in seconds -- C in {12, 9, 3}, lattice sizes that are NOT multiples of the
    brick edge, samples on brick faces and on the upper bbox face;
  * the three scatter variants against each other at BASELINE config-2 full size (160^3, 8192 rays x 256 samples):
    naive (one atomic per sample, corner, channel), per-wavefront de-duplicated atomics, bricks -- and a case built to
    overflow the de-duplication table (its direct-atomic fallback);
  * the Adam update fused into the brick kernel against dense gradients + MaskedAdam.step (bit-identical update rule,
    same set of updated voxels: adam_upd_kernel.cu:25-40).
Tolerance: gradient sums differ by summation order only -> rtol 1e-4.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(world, n_rays, C=12, seed=5, width=32, direct=False, fused=True, stepsize=0.5, scene='lego'):
    from directvoxgo_amd.dvgo import DirectVoxGO
    from directvoxgo_amd.scenes import roofline_scene, synthetic_scene
    if scene == 'roofline':
        sc = roofline_scene(world=world, n_rays=n_rays, seed=seed, device='cuda')
    else:
        sc = synthetic_scene(world=world, n_rays=n_rays, seed=seed, device='cuda', k0_dim=C, stepsize=stepsize)
    torch.manual_seed(1)
    kw = dict(rgbnet_dim=C, rgbnet_width=width, rgbnet_direct=direct) if C != 3 else dict(rgbnet_dim=0)
    m = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=world ** 3, num_voxels_base=world ** 3, alpha_init=1e-2,
                    fast_color_thres=1e-4, fused=fused, **kw).cuda()
    with torch.no_grad():
        m.density.copy_(sc['density']); m.k0.copy_(sc['k0']); m.mask_cache.mask.copy_(sc['mask'])
    return sc, m


def _grads(m, sc, variant, stepsize=None):
    """grid gradients of one forward + backward with the chosen scatter: 'brick' | 'dedup' | 'naive' | 'rows'"""
    from directvoxgo_amd import _lib as L, fused as F
    from directvoxgo_amd._lib import _int
    F.BRICK_SCATTER = variant == 'brick'
    F.COMBINED_GRID_GRAD, F.COMBINED_MIN_RATIO = variant == 'rows', 1e9
    L.call('dvgo_set_tuning', _int(0), _int(0 if variant == 'naive' else 1))
    L.call('dvgo_set_tuning', _int(1), _int(0 if variant == 'naive' else 1))
    try:
        m.zero_grad(set_to_none=True)
        res = m(sc['rays_o'], sc['rays_d'], sc['viewdirs'], near=sc['near'], far=sc['far'], bg=1,
                stepsize=stepsize or sc['stepsize'])
        loss = (res['rgb_marched'] - sc['target']).pow(2).mean() + 0.01 * res['alphainv_last'].clamp(1e-6, 1 - 1e-6).log().mean()
        loss.backward()
        torch.cuda.synchronize()
        return res, m.density.grad.clone(), m.k0.grad.clone()
    finally:
        F.BRICK_SCATTER, F.COMBINED_GRID_GRAD, F.COMBINED_MIN_RATIO = True, True, 6
        L.call('dvgo_set_tuning', _int(0), _int(1))
        L.call('dvgo_set_tuning', _int(1), _int(1))


def _close(a, b, rtol=1e-4):
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= rtol * scale + 1e-9, (float((a - b).abs().max()), scale)


@pytest.mark.parametrize('world,C', [(20, 12), (23, 12), (17, 3), (26, 9)])
def test_brick_gradients_match_oracle_scatter(world, C, oracle):
    """Lattices that are not multiples of 8 (partial bricks), C = 12 / 3 / 9: the brick path's dense gradients vs the
    oracle's scatter of the SAME per-sample gradients (taken from the op-by-op path's autograd graph)."""
    sc, m = _model(world, 700, C=C)
    res, gd, gk = _grads(m, sc, 'brick')
    assert gk.stride() == m.k0.stride()
    # the same scene through the op-by-op path with the naive atomic kernels = the reference's sums in another order
    m.fused = False
    _, gd_ref, gk_ref = _grads(m, sc, 'naive')
    _close(gd, gd_ref)
    _close(gk, gk_ref)
    assert torch.equal(gd != 0, gd_ref != 0) and torch.equal(gk != 0, gk_ref != 0)      # same voxels touched (masked Adam)

    # oracle: scatter the per-sample feature gradient the model produced
    m.fused = True
    from directvoxgo_amd import fused as F
    m.zero_grad(set_to_none=True)
    w, alpha, last, feat, ray_id, step_id, off3 = F.fused_march(m.density, m.k0, sc['rays_o'], sc['rays_d'],
                                                                m._march_cfg(sc['near'], sc['far'], sc['stepsize']))
    g_feat = torch.randn_like(feat)
    feat.backward(g_feat)
    mn, mx = sc['xyz_min'].cpu().numpy(), sc['xyz_max'].cpu().numpy()
    stepdist = float(np.float32(sc['stepsize']) * m.voxel_size.numpy())
    pts, mo, rid, sid, *_ = oracle.sample_pts_on_rays(sc['rays_o'].cpu().numpy(), sc['rays_d'].cpu().numpy(), mn, mx,
                                                      sc['near'], sc['far'], stepdist)
    key = {(int(r), int(s)): i for i, (r, s) in enumerate(zip(rid, sid))}
    sel = np.array([key[(int(r), int(s))] for r, s in zip(ray_id.cpu().numpy(), step_id.cpu().numpy())], np.int64)
    ref = oracle.grid_sample_bwd(g_feat.cpu().numpy(), tuple(m.k0.shape[1:]), pts[sel], mn, mx)
    got = m.k0.grad[0].cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-6 * np.abs(ref).max())


def test_scatter_variants_agree_at_full_size():
    """BASELINE config 2 at full size (160^3, 8192 rays x 256 samples, every sample kept): naive atomics, the
    de-duplicated atomics (per grid and as combined 64-byte rows) and the brick scatter produce the same gradients."""
    sc, m = _model(160, 8192, width=128, direct=True, scene='roofline')
    res, gd_n, gk_n = _grads(m, sc, 'naive')
    assert res['weights'].numel() == 8192 * 256
    for variant in ('dedup', 'rows', 'brick'):
        _, gd, gk = _grads(m, sc, variant)
        _close(gd, gd_n)
        _close(gk, gk_n)
        assert int((gk != 0).sum()) == int((gk_n != 0).sum()) and int((gd != 0).sum()) == int((gd_n != 0).sum())


def test_dedup_table_overflow_fallback_and_bricks_on_scattered_samples():
    """Steps of ~3.7 voxels: consecutive samples share no corner, so the 256-slot table of the de-duplicating kernel
    meets 256 distinct voxels per pass and its probe bound overflows into the direct-atomic fallback (march.hip);
    every brick list is made of single-sample runs and most samples straddle brick faces."""
    sc, m = _model(64, 4096, stepsize=3.7)
    _, gd_n, gk_n = _grads(m, sc, 'naive')
    assert int((gk_n != 0).sum()) > 0
    for variant in ('dedup', 'brick'):
        _, gd, gk = _grads(m, sc, variant)
        _close(gd, gd_n)
        _close(gk, gk_n)
        assert torch.equal(gk != 0, gk_n != 0)


@pytest.mark.parametrize('world,scene,n_rays', [(48, 'lego', 2048), (160, 'roofline', 8192)])
def test_adam_fused_into_the_brick_kernel_equals_dense_gradients_plus_masked_adam(world, scene, n_rays):
    """Three steps of TrainStep with the Adam update applied inside the brick kernel vs the same steps with dense
    gradients + MaskedAdam.step (rows_adam=False): same rule (adam_upd_kernel.cu:25-40) on gradients that differ by
    summation order only; the set of voxels that moved is identical."""
    from directvoxgo_amd.train import FINE_TRAIN, TrainStep
    outs = []
    for fused_adam in (True, False):
        sc, m = _model(world, n_rays, width=128, direct=True, scene=scene)
        p0 = (m.density.detach().clone(), m.k0.detach().clone())
        step = TrainStep(m, dict(FINE_TRAIN), dict(near=sc['near'], far=sc['far'], bg=1, stepsize=sc['stepsize']),
                         rows_adam=fused_adam)
        for it in range(3):
            step(sc['rays_o'], sc['rays_d'], sc['viewdirs'], sc['target'], global_step=5000 + it)
        torch.cuda.synchronize()
        if fused_adam:
            assert m.density.grad is None and m.k0.grad is None            # never materialised
        st = step.optimizer.state[m.k0]
        assert st['step'] == 3
        outs.append((m.density.detach().clone(), m.k0.detach().clone(), st['exp_avg'].clone(), st['exp_avg_sq'].clone(), p0))
    (d_a, k_a, m_a, v_a, p0), (d_b, k_b, m_b, v_b, _) = outs
    # the masked rule updates an element iff its gradient is not exactly 0: the two runs sum the same contributions in
    # different orders, so the sets agree except where a sum of a few terms cancels to exactly 0.0f in one order and to
    # a last-bit residue in the other (a 2^-24-ish event per element: a handful out of 53 M at full size, none at 48^3)
    n_d, n_k = int(((d_a != p0[0]) != (d_b != p0[0])).sum()), int(((k_a != p0[1]) != (k_b != p0[1])).sum())
    assert n_d <= 1e-6 * d_a.numel() + (0 if world < 100 else 2) and n_k <= 1e-6 * k_a.numel() + (0 if world < 100 else 2), (n_d, n_k)
    assert int((k_a != p0[1]).sum()) > 0
    # Adam normalises the step (lr 0.1 * m / sqrt(v)): where a gradient nearly cancels, summation-order noise is
    # amplified, so compare the moments tightly and the parameters with the usual Adam allowance
    _close(m_a, m_b, rtol=2e-4)
    _close(v_a, v_b, rtol=2e-4)
    assert float((k_a - k_b).abs().max()) <= 2e-3 and float((d_a - d_b).abs().max()) <= 2e-3


def test_brick_lists_cover_every_sample_corner_exactly_once():
    """Structure of the lists themselves: the number of entries equals the number of (sample, brick) incidences
    computed independently on the host, for a lattice with partial bricks."""
    from directvoxgo_amd import fused as F
    sc, m = _model(23, 500)
    cfg = m._march_cfg(sc['near'], sc['far'], sc['stepsize'])
    w, alpha, last, feat, ray_id, step_id, off3 = F.fused_march(m.density, m.k0, sc['rays_o'], sc['rays_d'], cfg)
    node = feat.grad_fn
    brick_off, _, _, _, _, E = node.bricks
    off = brick_off[0].cpu().numpy()
    assert off[0] == 0 and off[-1] == E and np.all(np.diff(off) >= 0)
    assert E >= w.numel()                         # every kept sample is listed at least once (plus alpha-only ones)


def test_heavy_bricks_are_split_into_slices_that_meet_in_scratch_tiles(monkeypatch):
    """A thin slab of matter crossed by 16384 rays: a few bricks collect far more than the slice length, so their
    lists are summed by several workgroups that meet through the scratch tiles and the arrival counters.  Dense gradients
    and the fused Adam update must match the atomic scatter / the dense path as everywhere else."""
    from directvoxgo_amd import _lib as L, fused as F
    from directvoxgo_amd.train import FINE_TRAIN, TrainStep
    assert L.lib().dvgo_brick_slice() >= 256
    slice_len = 1024
    monkeypatch.setattr(F, 'BRICK_SLICE', slice_len)
    sc, m = _model(48, 16384, width=128, direct=True)
    with torch.no_grad():                                    # matter only in a 6-voxel slab: every ray's samples pile up there
        d = torch.full_like(m.density, -20.0)
        d[:, :, 20:26] = 6.0
        m.density.copy_(d)
        m.mask_cache.mask.fill_(True)
    _, gd_n, gk_n = _grads(m, sc, 'naive')
    _, gd, gk = _grads(m, sc, 'brick')
    _close(gd, gd_n)
    _close(gk, gk_n)
    assert torch.equal(gk != 0, gk_n != 0)
    # the lists really were sliced
    cfg = m._march_cfg(sc['near'], sc['far'], sc['stepsize'])
    w, alpha, last, feat, ray_id, step_id, off3 = F.fused_march(m.density, m.k0, sc['rays_o'], sc['rays_d'], cfg)
    tables = feat.grad_fn.bricks[0].cpu().numpy()
    counts = np.diff(tables[0])
    assert counts.max() > 2 * slice_len
    assert tables[1][-1] >= 2                                              # extra work items in use
    assert np.array_equal(tables[2][:tables[2][-1]], np.nonzero(counts)[0])   # the list of non-empty bricks
    assert np.array_equal(np.diff(tables[1]), np.maximum(1, -(-counts // slice_len)) - 1)
    # and the fused Adam epilogue of a sliced brick == dense gradients + MaskedAdam.step
    outs = []
    for fused_adam in (True, False):
        sc2, m2 = _model(48, 16384, width=128, direct=True)
        with torch.no_grad():
            m2.density.copy_(d); m2.mask_cache.mask.fill_(True)
        step = TrainStep(m2, dict(FINE_TRAIN), dict(near=sc2['near'], far=sc2['far'], bg=1, stepsize=sc2['stepsize']),
                         rows_adam=fused_adam)
        step(sc2['rays_o'], sc2['rays_d'], sc2['viewdirs'], sc2['target'], global_step=5000)
        outs.append((m2.density.detach().clone(), m2.k0.detach().clone(), step.optimizer.state[m2.k0]['exp_avg'].clone()))
    _close(outs[0][2], outs[1][2], rtol=2e-4)
    assert float((outs[0][1] - outs[1][1]).abs().max()) <= 2e-3 and float((outs[0][0] - outs[1][0]).abs().max()) <= 2e-3
