"""GPU parity tests proper: every op of the drop-in surface, called through the C ABI
(directvoxgo_amd.render_utils / ops -> libdvgo_hip.so), against the CPU oracle on the same
seeded inputs and against the committed golden fixtures.

Bars: integer ids, counts and masks exact; positions and everything
that does not pass through libm bit-exact; activation values rtol 1e-5 / atol 1e-6;
grid gradients rtol 1e-4 / atol 1e-6 (float atomics, summation order).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ru():
    from directvoxgo_amd import render_utils
    return render_utils


@pytest.fixture(scope='module')
def ops():
    from directvoxgo_amd import ops
    return ops


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def make_rays(rng, n, special=True):
    """camera-like rays towards a unit-ish box, plus the edge cases the slab test cares about."""
    o = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    o = (o / np.linalg.norm(o, axis=1, keepdims=True) * rng.uniform(2.5, 4, (n, 1))).astype(np.float32)
    tgt = rng.uniform(-0.9, 0.9, (n, 3)).astype(np.float32)
    d = (tgt - o) * rng.uniform(0.3, 1.5, (n, 1)).astype(np.float32)       # |d| != 1 on purpose
    if special and n >= 8:
        o[0], d[0] = [-3, 0.2, 0.1], [1, 0, 0]          # zero components
        o[1], d[1] = [0.3, -3, 0.2], [0, 2, 0]
        o[2], d[2] = [0.1, 0.2, 3.0], [0, 0, -0.5]
        o[3], d[3] = [3, 3, 3], [1, 0.1, 0.1]           # misses the box
        o[4], d[4] = [0.1, -0.2, 0.3], [0.3, 0.5, -0.2]  # starts inside
        o[5], d[5] = [0, 0, 5], [0, 0, 1]               # points away
    return o, d.astype(np.float32)


BOX = (np.array([-1.0, -0.9, -1.1], np.float32), np.array([1.0, 1.1, 0.9], np.float32))


# ------------------------------------------------------------------ K1-K6
@pytest.mark.parametrize('n', [1, 7, 300, 8192])
def test_sampling_helpers_bit_exact(ru, oracle, n):
    rng = np.random.default_rng(10 + n)
    o, d = make_rays(rng, n)
    mn, mx = BOX
    near, far, stepdist = 0.2, 6.0, np.float32(0.013)
    t_min, t_max = ru.infer_t_minmax(cu(o), cu(d), cu(mn), cu(mx), near, far)
    et_min, et_max = oracle.infer_t_minmax(o, d, mn, mx, near, far)
    assert np.array_equal(t_min.cpu().numpy(), et_min) and np.array_equal(t_max.cpu().numpy(), et_max)
    ns = ru.infer_n_samples(t_min, t_max, stepdist)
    assert ns.dtype == torch.int64
    assert np.array_equal(ns.cpu().numpy(), oracle.infer_n_samples(et_min, et_max, stepdist))
    start, dirs = ru.infer_ray_start_dir(cu(o), cu(d), t_min)
    es, ed = oracle.infer_ray_start_dir(o, d, et_min)
    assert np.array_equal(start.cpu().numpy(), es) and np.array_equal(dirs.cpu().numpy(), ed)


@pytest.mark.parametrize('n', [1, 9, 513, 8192])
def test_sample_pts_on_rays_exact(ru, oracle, n):
    rng = np.random.default_rng(20 + n)
    o, d = make_rays(rng, n)
    mn, mx = BOX
    near, far, stepdist = 0.2, 6.0, np.float32(0.02)
    out = ru.sample_pts_on_rays(cu(o), cu(d), cu(mn), cu(mx), near, far, stepdist)
    exp = oracle.sample_pts_on_rays(o, d, mn, mx, near, far, stepdist)
    names = ['rays_pts', 'mask_outbbox', 'ray_id', 'step_id', 'N_steps', 't_min', 't_max']
    assert out[1].dtype == torch.bool and out[2].dtype == torch.int64 and out[4].dtype == torch.int64
    for name, a, b in zip(names, out, exp):
        assert np.array_equal(a.cpu().numpy(), b), name
    assert out[0].shape == (int(exp[4].sum()), 3)


def test_sample_pts_zero_rays_and_errors(ru):
    mn, mx = cu(BOX[0]), cu(BOX[1])
    e = torch.zeros((0, 3), device='cuda')
    out = ru.sample_pts_on_rays(e, e, mn, mx, 0.2, 6.0, 0.02)     # run.py:91 produces empty chunks
    assert out[0].shape == (0, 3) and out[2].numel() == 0 and out[4].numel() == 0
    o = torch.zeros((4, 3))
    with pytest.raises(RuntimeError, match='must be a CUDA tensor'):
        ru.sample_pts_on_rays(o, o, mn, mx, 0.2, 6.0, 0.02)
    nc = torch.zeros((3, 4), device='cuda').t()
    with pytest.raises(RuntimeError, match='must be contiguous'):
        ru.sample_pts_on_rays(nc, nc, mn, mx, 0.2, 6.0, 0.02)


def test_sample_ndc_exact(ru, oracle):
    rng = np.random.default_rng(30)
    n = 777
    o = np.concatenate([rng.uniform(-1.2, 1.2, (n, 2)), -np.ones((n, 1))], 1).astype(np.float32)
    d = np.concatenate([rng.uniform(-0.5, 0.5, (n, 2)), 2 * np.ones((n, 1))], 1).astype(np.float32)
    mn, mx = np.array([-1, -1, -1], np.float32), np.array([1, 1, 1], np.float32)
    for S in (2, 17, 255):
        pts, m = ru.sample_ndc_pts_on_rays(cu(o), cu(d), cu(mn), cu(mx), S)
        ep, em = oracle.sample_ndc_pts_on_rays(o, d, mn, mx, S)
        assert np.array_equal(pts.cpu().numpy(), ep) and np.array_equal(m.cpu().numpy(), em)
    pts, m = ru.sample_ndc_pts_on_rays(cu(o[:0]), cu(d[:0]), cu(mn), cu(mx), 5)
    assert pts.shape == (0, 5, 3)


# ------------------------------------------------------------------ K8
def test_maskcache_lookup_exact(ru, oracle):
    rng = np.random.default_rng(40)
    world = rng.random((23, 17, 31)) < 0.3
    mn, mx = BOX
    scale = ((np.array(world.shape, np.float32) - 1) / (mx - mn)).astype(np.float32)
    shift = (-mn * scale).astype(np.float32)
    xyz = rng.uniform(-1.3, 1.3, (100001, 3)).astype(np.float32)
    out = ru.maskcache_lookup(cu(world), cu(xyz), cu(scale), cu(shift))
    assert out.dtype == torch.bool
    assert np.array_equal(out.cpu().numpy(), oracle.maskcache_lookup(world, xyz, scale, shift))
    assert ru.maskcache_lookup(cu(world), cu(xyz[:0]), cu(scale), cu(shift)).numel() == 0


# ------------------------------------------------------------------ K9 / K10
@pytest.mark.parametrize('n', [0, 1, 3, 4, 1023, 100003])
def test_raw2alpha(ru, oracle, n):
    rng = np.random.default_rng(50 + n)
    d = (rng.standard_normal(n) * 8).astype(np.float32)
    if n > 10:
        d[:4] = [200, -200, 88.5, 0]
    for shift, interval in [(-4.5951, 0.5), (-13.8155, 1.0), (0.0, 2.0)]:
        e, a = ru.raw2alpha(cu(d), shift, interval)
        ee, ea = oracle.raw2alpha(d, shift, interval)
        np.testing.assert_allclose(e.cpu().numpy(), ee, rtol=1e-5)
        np.testing.assert_allclose(a.cpu().numpy(), ea, rtol=1e-5, atol=1e-6)
        gb = rng.standard_normal(n).astype(np.float32)
        g = ru.raw2alpha_backward(cu(ee), cu(gb), interval)
        np.testing.assert_allclose(g.cpu().numpy(), oracle.raw2alpha_backward(ee, gb, interval), rtol=1e-5, atol=1e-7)


def test_raw2alpha_unaligned_and_tensor_scalars(ru, oracle):
    d = torch.randn(1001, device='cuda')[1:]                # 4-byte aligned only -> scalar kernel
    e, a = ru.raw2alpha(d.contiguous() if not d.is_contiguous() else d, torch.tensor(-4.5951), torch.tensor(0.5))
    ee, ea = oracle.raw2alpha(d.cpu().numpy(), -4.5951, 0.5)
    np.testing.assert_allclose(a.cpu().numpy(), ea, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ K11-K13
def ragged(rng, n_rays, max_len, empty_every=5):
    lens = rng.integers(1, max_len, n_rays)
    lens[::empty_every] = 0
    return lens, np.repeat(np.arange(n_rays), lens)


@pytest.mark.parametrize('n_rays,max_len,amax', [(1, 40, 0.3), (40, 150, 0.5), (300, 700, 0.05), (64, 300, 1.0)])
def test_alpha2weight_bit_exact(ru, oracle, n_rays, max_len, amax):
    rng = np.random.default_rng(60 + n_rays)
    lens, ray_id = ragged(rng, n_rays, max_len)
    alpha = (rng.random(ray_id.shape[0]) ** 2 * amax).astype(np.float32)
    if amax == 1.0 and alpha.size:
        alpha[rng.integers(0, alpha.size, 20)] = 1.0         # fully opaque samples -> T = 1e-10 * T
    out = ru.alpha2weight(cu(alpha), cu(ray_id), n_rays)
    exp = oracle.alpha2weight(alpha, ray_id, n_rays)
    for name, a, b in zip(['weights', 'T', 'alphainv_last', 'i_start', 'i_end'], out, exp):
        assert np.array_equal(a.cpu().numpy(), b), name
    gw = rng.standard_normal(alpha.shape).astype(np.float32)
    gl = rng.standard_normal(n_rays).astype(np.float32)
    g = ru.alpha2weight_backward(cu(alpha), out[0], out[1], out[2], out[3], out[4], n_rays, cu(gw), cu(gl))
    eg = oracle.alpha2weight_backward(alpha, *exp, n_rays, gw, gl, fma=True)
    assert np.array_equal(g.cpu().numpy(), eg)


def test_alpha2weight_empty(ru):
    w, T, last, i_s, i_e = ru.alpha2weight(torch.zeros(0, device='cuda'), torch.zeros(0, dtype=torch.int64, device='cuda'), 5)
    assert w.numel() == 0 and torch.all(last == 1) and torch.all(i_s == 0) and torch.all(i_e == 0)


def test_alphas2weights_autograd(ops, oracle):
    rng = np.random.default_rng(61)
    lens, ray_id = ragged(rng, 50, 90)
    alpha = (rng.random(ray_id.shape[0]) * 0.3).astype(np.float32)
    a = cu(alpha).requires_grad_()
    w, last = ops.Alphas2Weights.apply(a, cu(ray_id), 50)
    gw = rng.standard_normal(alpha.shape).astype(np.float32); gl = rng.standard_normal(50).astype(np.float32)
    (w * cu(gw)).sum().add((last * cu(gl)).sum()).backward()
    ew, eT, el, es, ee = oracle.alpha2weight(alpha, ray_id, 50)
    eg = oracle.alpha2weight_backward(alpha, ew, eT, el, es, ee, 50, gw, gl)
    assert np.array_equal(a.grad.cpu().numpy(), eg)


# ------------------------------------------------------------------ A4 / A8 trilinear
def grid_layouts(grid_np):
    """[C,X,Y,Z] numpy -> {'cf': channel-first tensor, 'cl': channels-last tensor}, both [1,C,X,Y,Z]"""
    g = cu(grid_np)[None]
    out = {'cf': g.contiguous()}
    if grid_np.shape[0] > 1:
        out['cl'] = g.contiguous(memory_format=torch.channels_last_3d)
    return out


@pytest.mark.parametrize('C', [1, 3, 12])
def test_grid_sample_golden_and_oracle(ops, oracle, C):
    g = load_golden('grid_sampler')
    grid = g[f'grid_c{C}'][0]
    xyz, mn, mx = g['xyz'], g['xyz_min'], g['xyz_max']
    exp = oracle.grid_sample_fwd(grid, xyz, mn, mx, use_fma=True)
    for name, gt in grid_layouts(grid).items():
        gt.requires_grad_()
        out = ops.grid_sample(gt, cu(xyz), cu(mn), cu(mx))
        o = out.detach().cpu().numpy().reshape(-1, C)
        assert np.array_equal(o, exp), name                                     # bit-exact vs oracle
        np.testing.assert_allclose(o, g[f'out_c{C}'].reshape(-1, C), rtol=1e-5, atol=1e-6)   # vs reference run
        out.reshape(-1, C).backward(cu(g[f'gout_c{C}']))
        assert gt.grad.stride() == gt.stride()
        np.testing.assert_allclose(gt.grad.cpu().numpy()[0], g[f'ggrid_c{C}'][0], rtol=1e-4, atol=1e-6)


def test_grid_sample_shapes_and_empty(ops):
    grid = torch.randn(1, 12, 8, 9, 10, device='cuda').contiguous(memory_format=torch.channels_last_3d)
    mn, mx = cu(BOX[0]), cu(BOX[1])
    assert ops.grid_sample(grid, torch.rand(5, 7, 3, device='cuda'), mn, mx).shape == (5, 7, 12)
    assert ops.grid_sample(grid[:, :1].contiguous(), torch.rand(5, 7, 3, device='cuda'), mn, mx).shape == (5, 7)
    assert ops.grid_sample(grid, torch.rand(0, 3, device='cuda'), mn, mx).shape == (0, 12)


def test_grid_sample_full_size_properties(ops):
    """160^3 x 12 channels, 2M points: linearity in the grid, partition of unity, scatter mass."""
    torch.manual_seed(0)
    ws = (160, 160, 160)
    mn, mx = cu(np.array([-1.575] * 3, np.float32)), cu(np.array([1.575] * 3, np.float32))
    M = 2_097_152
    xyz = (torch.rand(M, 3, device='cuda') * 3.1 - 1.55)
    ga = torch.randn(1, 12, *ws, device='cuda').contiguous(memory_format=torch.channels_last_3d)
    gb = torch.randn_like(ga)
    fa, fb = ops.grid_sample(ga, xyz, mn, mx), ops.grid_sample(gb, xyz, mn, mx)
    fab = ops.grid_sample(ga * 2 + gb, xyz, mn, mx)
    assert torch.allclose(fab, 2 * fa + fb, rtol=1e-4, atol=1e-4)
    ones = torch.ones(1, 12, *ws, device='cuda').contiguous(memory_format=torch.channels_last_3d).requires_grad_()
    f1 = ops.grid_sample(ones, xyz, mn, mx)
    assert torch.allclose(f1, torch.ones_like(f1), atol=1e-5)                 # weights sum to 1 inside the box
    go = torch.randn(M, 12, device='cuda')
    f1.backward(go)
    assert torch.allclose(ones.grad.sum((0, 2, 3, 4)).double(), go.double().sum(0).float().double(), rtol=1e-3, atol=0.5)


# ------------------------------------------------------------------ A9
@pytest.mark.parametrize('C', [1, 3])
def test_segment_coo(ops, oracle, C):
    rng = np.random.default_rng(70 + C)
    lens, idx = ragged(rng, 500, 300)
    src = rng.standard_normal((idx.shape[0], C)).astype(np.float32)
    s = cu(src if C > 1 else src[:, 0]).requires_grad_()
    zeros = torch.zeros((500, C) if C > 1 else (500,), device='cuda')
    out = ops.segment_coo(src=s, index=cu(idx), out=zeros, reduce='sum')
    exp = oracle.segment_sum(src if C > 1 else src[:, 0], idx, 500)
    np.testing.assert_allclose(out.detach().cpu().numpy(), exp, rtol=1e-5, atol=1e-5)
    go = torch.randn_like(out)
    out.backward(go)
    assert torch.equal(s.grad, go[cu(idx)])


# ------------------------------------------------------------------ N1 / N2
def test_adam_kernels_vs_fixture(oracle):
    from directvoxgo_amd.masked_adam import MaskedAdam
    g = load_golden('masked_adam')
    for tag, skip in [('plain', False), ('masked', True), ('perlr', False)]:
        p = torch.nn.Parameter(cu(g[f'{tag}_p0']))
        opt = MaskedAdam([{'params': [p], 'lr': 0.1, 'skip_zero_grad': skip}])
        if tag == 'perlr':
            opt.set_pervoxel_lr(cu(g[f'{tag}_count']))
        for s in range(3):
            p.grad = cu(g[f'{tag}_g{s}'])
            opt.step()
            np.testing.assert_allclose(p.detach().cpu().numpy(), g[f'{tag}_p{s + 1}'], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(opt.state[p]['exp_avg'].cpu().numpy(), g[f'{tag}_exp_avg'], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize('shape', [(1, 12, 9, 10, 11), (1, 1, 7, 5, 300), (1, 9, 3, 4, 70)])
@pytest.mark.parametrize('dense', [True, False])
def test_total_variation(ops, oracle, dense, shape):
    rng = np.random.default_rng(80)
    p = (rng.standard_normal(shape) * 2).astype(np.float32)
    grad = rng.standard_normal(p.shape).astype(np.float32)
    grad[rng.random(p.shape) < 0.6] = 0
    exp = grad.copy()
    oracle.total_variation_add_grad(p, exp, 0.3, 0.6, 1.2, dense)
    for fmt in (torch.contiguous_format, torch.channels_last_3d):
        pt = cu(p).contiguous(memory_format=fmt)
        gt = cu(grad).contiguous(memory_format=fmt)
        ops.total_variation_add_grad(pt, gt, 0.3, 0.6, 1.2, dense)
        np.testing.assert_array_equal(gt.cpu().numpy(), exp)


@pytest.mark.parametrize('n', [1, 8192, 16384, 16385, 100000, 1000003])
def test_scans_short_and_long(n):
    """One-workgroup scan up to 16384 items, three-launch block scan above (full-image ray chunks)."""
    from directvoxgo_amd import _lib as L
    from directvoxgo_amd._lib import _flt, _i64, ptr, stream_of
    g = torch.Generator(device='cuda').manual_seed(n)
    counts = torch.randint(0, 300, (n,), device='cuda', dtype=torch.int32, generator=g)
    off = torch.empty(n + 1, dtype=torch.int64, device='cuda')
    L.call('dvgo_exclusive_scan_i32', ptr(counts), _i64(n), ptr(off), stream_of(counts))
    ref = torch.cat([torch.zeros(1, dtype=torch.int64, device='cuda'), counts.long().cumsum(0)])
    assert torch.equal(off, ref)
    # inclusive int64 scan: the n_steps cumsum of dvgo_sample_pts_prepare
    ro = torch.rand(n, 3, device='cuda', generator=g) * 0.2 - 3.0
    rd = torch.nn.functional.normalize(torch.rand(n, 3, device='cuda', generator=g) + 0.2, dim=-1)
    mn, mx = torch.tensor([-1.0, -1.0, -1.0], device='cuda'), torch.tensor([1.0, 1.0, 1.0], device='cuda')
    t_min, t_max = torch.empty(n, device='cuda'), torch.empty(n, device='cuda')
    n_steps, cum = torch.empty(n, dtype=torch.int64, device='cuda'), torch.empty(n, dtype=torch.int64, device='cuda')
    start, dirs = torch.empty(n, 3, device='cuda'), torch.empty(n, 3, device='cuda')
    L.call('dvgo_sample_pts_prepare', ptr(ro), ptr(rd), ptr(mn), ptr(mx), _flt(0.2), _flt(8.0), _flt(0.01), _i64(n), ptr(t_min),
           ptr(t_max), ptr(n_steps), ptr(cum), ptr(start), ptr(dirs), stream_of(ro))
    assert torch.equal(cum, n_steps.cumsum(0)) and int(n_steps.max()) > 1


# ------------------------------------------------------------------ N3 fused colour head
@pytest.mark.parametrize('M,diffuse,width,C,E', [
    (1, True, 128, 12, 27), (31, True, 128, 12, 27), (1000, False, 128, 12, 27), (70001, True, 128, 12, 27),
    (40000, False, 128, 12, 27),                       # the rgbnet_direct head of configs/default.py (d_in 39)
    (33, False, 64, 9, 3), (50001, False, 64, 9, 3),   # the LLFF head (configs/llff, lib/dmpigo.py): width 64, d_in 12
    (3000, True, 64, 12, 27), (3000, False, 128, 9, 3)])
@pytest.mark.parametrize('variant', [0, 7])
def test_shade_matches_torch_modules(M, diffuse, width, C, E, variant):
    """The colour head kernels vs the torch modules they replace (lib/dvgo.py:516-541), values and grads: csrc/shade.hip
    (fp32 MFMA, variant 0) and csrc/shade_x3.hip (bf16 matrix cores on exactly 3-way-split fp32 operands) -- the SAME
    tolerances for both."""
    from directvoxgo_amd import _lib as L
    from directvoxgo_amd.dvgo import make_rgbnet
    from directvoxgo_amd.shade import shade
    prev = L.lib().dvgo_shade_variant(variant)
    try:
        _shade_vs_torch(M, diffuse, width, C, E)
    finally:
        L.lib().dvgo_shade_variant(prev)


def _shade_vs_torch(M, diffuse, width, C, E):
    from directvoxgo_amd.dvgo import make_rgbnet
    from directvoxgo_amd.shade import shade
    torch.manual_seed(M)
    N = 50
    d_in = (C - 3 if diffuse else C) + E
    net = make_rgbnet(d_in, width, 3).cuda()
    feat = torch.randn(M, C, device='cuda', requires_grad=True)
    emb = torch.randn(N, E, device='cuda')
    ray_id = torch.sort(torch.randint(N, (M,), device='cuda'))[0]
    rgb = shade(net, feat, emb, ray_id, diffuse)
    assert rgb is not None and rgb.shape == (M, 3)
    x = torch.cat([feat[:, 3:] if diffuse else feat, emb[ray_id]], -1)
    ref = torch.sigmoid(net(x) + (feat[:, :3] if diffuse else 0))
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=2e-6)
    go = torch.randn_like(ref)
    gr = torch.autograd.grad(ref, [feat] + list(net.parameters()), go)
    gs = torch.autograd.grad(rgb, [feat] + list(net.parameters()), go)
    # a pre-activation within rounding of zero may land on the other side of the ReLU than in torch's GEMM (different
    # summation order): that sample's feature gradient then differs as a whole row -- allow a handful of such rows
    a, b = gs[0].cpu().numpy(), gr[0].cpu().numpy()
    bad_rows = (~np.isclose(a, b, rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(b).max())))).any(1)
    if bad_rows.sum() > max(1, M // 10000):
        # which side flipped?  torch's fp32 GEMM picks its algorithm (and summation order) per call; a float64 evaluation
        # of the same modules is the arbiter: a row only counts against the kernel if it is off against that as well
        import copy
        net64 = copy.deepcopy(net).double()
        f64 = feat.detach().double().requires_grad_()
        x64 = torch.cat([f64[:, 3:] if diffuse else f64, emb.double()[ray_id]], -1)
        ref64 = torch.sigmoid(net64(x64) + (f64[:, :3] if diffuse else 0))
        b64 = torch.autograd.grad(ref64, f64, go.double())[0].cpu().numpy()
        bad_rows &= (~np.isclose(a, b64, rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(b64).max())))).any(1)
    assert bad_rows.sum() <= max(1, M // 10000), f'{bad_rows.sum()} rows of the feature gradient differ'
    for a, b in zip(gs[1:], gr[1:]):
        a, b = a.cpu().numpy(), b.cpu().numpy()
        if bad_rows.sum() == 0:
            np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(b).max())))
        else:       # the flipped sample(s) enter every weight gradient with a whole-sample contribution
            assert np.linalg.norm(a - b) <= 2e-2 * np.linalg.norm(b)


def test_shade_falls_back_for_other_heads():
    from directvoxgo_amd.dvgo import make_rgbnet
    from directvoxgo_amd.shade import shade
    for net in (make_rgbnet(36, 96, 3), make_rgbnet(36, 128, 4), make_rgbnet(36, 256, 3)):
        assert shade(net.cuda(), torch.randn(5, 12, device='cuda'), torch.randn(2, 27, device='cuda'),
                     torch.zeros(5, dtype=torch.int64, device='cuda'), True) is None


# ------------------------------------------------------------------ H3 glue kernels
def test_fused_loss_matches_reference_formula():
    from directvoxgo_amd.train import FINE_TRAIN, COARSE_TRAIN, fused_render_loss, render_loss
    torch.manual_seed(0)
    N, M = 777, 20011
    for cfg in (dict(FINE_TRAIN), dict(COARSE_TRAIN), dict(FINE_TRAIN, weight_rgbper=0.0, weight_entropy_last=0.0)):
        leaves = [torch.rand(N, 3, device='cuda').requires_grad_(), torch.rand(N, device='cuda').requires_grad_(),
                  torch.rand(M, 3, device='cuda').requires_grad_(), torch.rand(M, device='cuda').requires_grad_()]
        with torch.no_grad():
            leaves[1][:5] = torch.tensor([0.0, 1.0, 1e-7, 1 - 1e-7, 0.5], device='cuda')     # clamp edges
        rid = torch.sort(torch.randint(N, (M,), device='cuda'))[0]
        tgt = torch.rand(N, 3, device='cuda')
        res = {'rgb_marched': leaves[0], 'alphainv_last': leaves[1], 'raw_rgb': leaves[2], 'weights': leaves[3], 'ray_id': rid}
        a = render_loss(res, tgt, 2 * N, cfg)
        ga = torch.autograd.grad(a, leaves[:3], allow_unused=True)
        b = fused_render_loss(res, tgt, 2 * N, cfg)
        gb = torch.autograd.grad(b, leaves[:3], allow_unused=True)
        np.testing.assert_allclose(float(b), float(a), rtol=2e-5)
        for x, y in zip(gb, ga):
            if y is None:
                assert x is None or float(x.abs().max()) == 0.0
            else:
                np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=1e-5, atol=1e-10)


def test_viewdir_embed_matches_torch_expression():
    from directvoxgo_amd.shade import viewdir_embed
    v = torch.nn.functional.normalize(torch.randn(1000, 3, device='cuda'), dim=-1)
    for F in (4, 0, 2):
        freq = torch.tensor([2.0 ** i for i in range(F)], device='cuda')
        e = (v.unsqueeze(-1) * freq).flatten(-2)
        ref = torch.cat([v, e.sin(), e.cos()], -1)
        np.testing.assert_allclose(viewdir_embed(v, freq).cpu().numpy(), ref.cpu().numpy(), rtol=1e-6, atol=1e-6)


def test_multi_tensor_adam_equals_single_launches():
    from directvoxgo_amd.masked_adam import MaskedAdam
    torch.manual_seed(1)
    shapes = [(128, 36), (128,), (128, 128), (128,), (3, 128), (3,)]
    pa = [torch.nn.Parameter(torch.randn(s, device='cuda')) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    big = torch.nn.Parameter(torch.randn(300000, device='cuda'))          # above the batching threshold
    oa = MaskedAdam([{'params': pa + [big], 'lr': 1e-3, 'skip_zero_grad': False}])
    ob = torch.optim.Adam(pb, lr=1e-3, betas=(0.9, 0.99), eps=1e-8)
    for _ in range(3):
        for x, y in zip(pa, pb):
            g = torch.randn_like(x); x.grad = g.clone(); y.grad = g.clone()
        big.grad = torch.randn_like(big)
        oa.step(); ob.step()
    for x, y in zip(pa, pb):
        np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=1e-5, atol=1e-7)


def test_deferred_wgrad_gives_identical_parameter_gradients():
    """shade.defer_wgrad (used by TrainStep to overlap the grid all-reduce with the weight-gradient kernel)."""
    from directvoxgo_amd.dvgo import make_rgbnet
    from directvoxgo_amd.shade import defer_wgrad, shade
    torch.manual_seed(5)
    M, N = 50000, 64
    net = make_rgbnet(36, 128, 3).cuda()
    feat = torch.randn(M, 12, device='cuda', requires_grad=True)
    emb = torch.randn(N, 27, device='cuda')
    ray_id = torch.sort(torch.randint(N, (M,), device='cuda'))[0]
    go = torch.randn(M, 3, device='cuda')
    shade(net, feat, emb, ray_id, True).backward(go)
    ref = [p.grad.clone() for p in net.parameters()]; gf = feat.grad.clone()
    net.zero_grad(set_to_none=True); feat.grad = None
    with defer_wgrad() as d:
        shade(net, feat, emb, ray_id, True).backward(go)
        assert all(p.grad is None for p in net.parameters()) and len(d.pending) == 1
    d.flush()
    assert torch.equal(feat.grad, gf)
    for p, r in zip(net.parameters(), ref):
        # the per-workgroup partial sums are combined with float atomics: two runs differ by summation order, i.e. by
        # rounding relative to the LARGEST terms of a sum, not to a result that may have cancelled to near zero
        assert float((p.grad - r).abs().max()) <= 1e-5 * float(r.abs().max())


@pytest.mark.parametrize('cl,C', [(True, 12), (True, 9), (False, 1)])
@pytest.mark.parametrize('dense', [True, False])
def test_total_variation_slab_by_slab_equals_whole_grid(cl, C, dense):
    """dvgo_total_variation_add_grad_slab: the TV gradient added plane-range by plane-range (what the data-parallel ranks
    do, each on the slab it owns) is the TV gradient of the whole grid, bit for bit."""
    from directvoxgo_amd.ops import total_variation_add_grad
    torch.manual_seed(2)
    p = torch.randn(1, C, 13, 9, 11, device='cuda')
    g = torch.randn(1, C, 13, 9, 11, device='cuda') * (torch.rand(1, C, 13, 9, 11, device='cuda') < 0.5)
    if cl:
        p, g = p.contiguous(memory_format=torch.channels_last_3d), g.contiguous(memory_format=torch.channels_last_3d)
    whole = g.clone()
    total_variation_add_grad(p, whole, 0.3, 0.3, 0.7, dense)
    parts = g.clone()
    for lo, hi in ((0, 4), (4, 5), (5, 13)):
        total_variation_add_grad(p, parts, 0.3, 0.3, 0.7, dense, x_range=(lo, hi))
    assert torch.equal(whole, parts)
    only = g.clone()
    total_variation_add_grad(p, only, 0.3, 0.3, 0.7, dense, x_range=(4, 5))
    assert torch.equal(only[:, :, :4], g[:, :, :4]) and torch.equal(only[:, :, 5:], g[:, :, 5:])
    assert torch.equal(only[:, :, 4:5], whole[:, :, 4:5])
