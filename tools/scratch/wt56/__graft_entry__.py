"""Driver entry points.

build(): compile every HIP source for gfx950 into directvoxgo_amd/csrc/libdvgo_hip.so (hipcc
         cross-compiles without a GPU), compile the CPU oracle (test infrastructure; building the
         checker is not using it) and import the package.  oracle/_ref (a build of the reference's own
         sources) does not exist for this reference: its native path is CUDA-only and unbuildable
         here (see oracle/Makefile).
smoke(): one small forward + backward of the fused ray-march path on cuda:0, checked against the
         CPU oracle.
"""
import os
import sys

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def build():
    from directvoxgo_amd import build as hip_build
    so = hip_build.build(verbose=True)
    from oracle import oracle as O
    O.build()
    import directvoxgo_amd  # noqa: F401
    from directvoxgo_amd import _lib
    _lib.lib()            # dlopen + ABI version check (no GPU needed)
    print('built', so)


def smoke():
    import numpy as np
    import torch
    from directvoxgo_amd.dvgo import DirectVoxGO
    from directvoxgo_amd.scenes import synthetic_scene
    from oracle import oracle as O

    assert torch.cuda.is_available(), 'smoke() needs cuda:0'
    torch.cuda.set_device(0)
    sc = synthetic_scene(world=32, n_rays=256, seed=777, device='cuda')
    m = DirectVoxGO(sc['xyz_min'], sc['xyz_max'], num_voxels=32 ** 3, num_voxels_base=32 ** 3, alpha_init=1e-2,
                    fast_color_thres=1e-4, rgbnet_dim=12, rgbnet_width=32, fused=True).cuda()
    with torch.no_grad():
        m.density.copy_(sc['density']); m.k0.copy_(sc['k0']); m.mask_cache.mask.copy_(sc['mask'])
    res = m(sc['rays_o'], sc['rays_d'], sc['viewdirs'], near=sc['near'], far=sc['far'], bg=1, stepsize=0.5)
    loss = (res['rgb_marched'] - sc['target']).pow(2).mean()
    loss.backward()
    torch.cuda.synchronize()

    # oracle check of the march (sampling -> mask -> density -> alpha -> weights), reference op order
    mn, mx = sc['xyz_min'].cpu().numpy(), sc['xyz_max'].cpu().numpy()
    stepdist = np.float32(0.5) * m.voxel_size.numpy()
    pts, mo, rid, sid, *_ = O.sample_pts_on_rays(sc['rays_o'].cpu().numpy(), sc['rays_d'].cpu().numpy(), mn, mx,
                                                 sc['near'], sc['far'], stepdist)
    pts, rid = pts[~mo], rid[~mo]
    mask = sc['mask'].cpu().numpy()
    scale = (np.array(mask.shape, np.float32) - 1) / (mx - mn)
    k = O.maskcache_lookup(mask, pts, scale, -mn * scale)
    pts, rid = pts[k], rid[k]
    dens = O.grid_sample_fwd(sc['density'][0].cpu().numpy(), pts, mn, mx)[:, 0]
    _, alpha = O.raw2alpha(dens, m.act_shift, np.float32(0.5) * m.voxel_size_ratio.numpy())
    k = alpha > 1e-4
    alpha, rid = alpha[k], rid[k]
    w, T, last, i_s, i_e = O.alpha2weight(alpha, rid, 256)
    k = w > 1e-4
    # index outputs exact; values to the activation's tolerance (device expf/powf vs glibc)
    assert int(k.sum()) == res['weights'].numel(), (int(k.sum()), res['weights'].numel())
    assert np.array_equal(res['ray_id'].cpu().numpy(), rid[k])
    np.testing.assert_allclose(res['alphainv_last'].detach().cpu().numpy(), last, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(res['weights'].detach().cpu().numpy(), w[k], rtol=1e-5, atol=1e-6)
    assert torch.isfinite(m.density.grad).all() and torch.isfinite(m.k0.grad).all()
    assert m.k0.grad.abs().sum() > 0 and m.density.grad.abs().sum() > 0
    print('smoke ok: rays=256 samples=%d loss=%.5f' % (res['weights'].numel(), float(loss)))


if __name__ == '__main__':
    build()
    if '--smoke' in sys.argv:
        smoke()
