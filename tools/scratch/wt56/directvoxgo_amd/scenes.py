"""Synthetic inputs for tests, smoke and bench (no dataset exists in the image).

Follows SURVEY.md section 8d:
  * lego-like cameras: pose_spherical(theta ~ U(-180,180), phi = -30, r = 4) and pixel-centre pinhole
    rays (conventions of /root/reference/lib/load_blender.py:14-42 and lib/ray_utils.py:9-47,80-85;
    pinned against the reference run in tests/golden/rays.npz);
  * a smooth blob density field with low-passed noise, random features, occupancy mask from the
    max-pooled activated density;
  * the "roofline case": unit-direction rays crossing the box face to face with near = 0 and
    far = 255.5 * stepdist, so that every ray has exactly 256 in-box samples, a density for which no
    sample is culled by either threshold and no ray terminates early (M = N * 256).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def pose_spherical(theta, phi, radius):
    """camera-to-world of a camera on a sphere looking at the origin (blender convention)."""
    th, ph = math.radians(theta), math.radians(phi)
    trans = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], dtype=torch.float32)
    rphi = torch.tensor([[1, 0, 0, 0], [0, math.cos(ph), -math.sin(ph), 0], [0, math.sin(ph), math.cos(ph), 0],
                         [0, 0, 0, 1]], dtype=torch.float32)
    rth = torch.tensor([[math.cos(th), 0, -math.sin(th), 0], [0, 1, 0, 0], [math.sin(th), 0, math.cos(th), 0],
                        [0, 0, 0, 1]], dtype=torch.float32)
    flip = torch.tensor([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=torch.float32)
    return flip @ (rth @ (rphi @ trans))


def camera_rays(H, W, focal, c2w, pix_i=None, pix_j=None):
    """Pixel-centre pinhole rays (non-inverse-y convention): returns rays_o, rays_d, viewdirs for
    the given pixel columns/rows (all pixels, row-major, when omitted)."""
    if pix_i is None:
        jj, ii = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
        pix_i, pix_j = ii.reshape(-1), jj.reshape(-1)
    i = pix_i + 0.5
    j = pix_j + 0.5
    dirs = torch.stack([(i - 0.5 * W) / focal, -(j - 0.5 * H) / focal, -torch.ones_like(i)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, 3].expand(rays_d.shape)
    viewdirs = rays_d / rays_d.norm(dim=-1, keepdim=True)
    return rays_o.contiguous(), rays_d.contiguous(), viewdirs.contiguous()


def lego_like_rays(n_rays, gen, n_views=100, H=800, W=800, focal=1111.11, radius=4.0, phi=-30.0):
    """n_rays rays drawn uniformly from the pixels of n_views random lego-like cameras."""
    thetas = torch.rand(n_views, generator=gen) * 360 - 180
    view = torch.randint(n_views, (n_rays,), generator=gen)
    pi = torch.randint(W, (n_rays,), generator=gen).float()
    pj = torch.randint(H, (n_rays,), generator=gen).float()
    ro = torch.empty(n_rays, 3); rd = torch.empty(n_rays, 3); vd = torch.empty(n_rays, 3)
    for v in range(n_views):
        sel = (view == v).nonzero().squeeze(1)
        if sel.numel() == 0:
            continue
        o, d, u = camera_rays(H, W, focal, pose_spherical(float(thetas[v]), phi, radius), pi[sel], pj[sel])
        ro[sel], rd[sel], vd[sel] = o, d, u
    return ro, rd, vd


def blob_density(ws, xyz_min, xyz_max, gen, amp=16.0, bias=-9.0, noise=1.5, radius=0.9):
    ax = [torch.linspace(float(xyz_min[a]), float(xyz_max[a]), ws[a]) for a in range(3)]
    xx, yy, zz = torch.meshgrid(*ax, indexing='ij')
    r = torch.sqrt(xx ** 2 + yy ** 2 + zz ** 2) / radius
    n = torch.randn(ws, generator=gen)
    n = F.avg_pool3d(n[None, None], 3, 1, 1)[0, 0]
    return amp * torch.exp(-r ** 4) + bias + noise * n


def activate(density, act_shift, interval):
    return 1 - torch.exp(-F.softplus(density + act_shift) * interval)


def synthetic_scene(world=160, n_rays=8192, seed=777, device='cpu', k0_dim=12, alpha_init=1e-2,
                    fast_color_thres=1e-4, bbox=1.5, bound_scale=1.05, stepsize=0.5):
    """config-2-like scene (lego fine stage): cubic bbox [-1.5,1.5]^3 * 1.05, world^3 grid, blob density
    with ~15-30 % occupied voxels, N(0, 0.3^2) features, lego-like rays, near/far = 2/6."""
    gen = torch.Generator().manual_seed(seed)
    mn = torch.full((3,), -bbox * bound_scale); mx = torch.full((3,), bbox * bound_scale)
    ws = (world, world, world)
    act_shift = math.log(1 / (1 - alpha_init) - 1)
    density = blob_density(ws, mn, mx, gen)
    k0 = torch.randn((1, k0_dim, *ws), generator=gen) * 0.3
    alpha = F.max_pool3d(activate(density, act_shift, 1.0)[None, None], 3, 1, 1)[0, 0]
    mask = alpha > fast_color_thres
    ro, rd, vd = lego_like_rays(n_rays, gen)
    target = torch.rand((n_rays, 3), generator=gen)
    sc = dict(xyz_min=mn, xyz_max=mx, density=density[None, None], k0=k0, mask=mask, rays_o=ro, rays_d=rd,
              viewdirs=vd, target=target)
    sc = {k: v.to(device) for k, v in sc.items()}
    sc.update(near=2.0, far=6.0, stepsize=stepsize, world=world, occupancy=float(mask.float().mean()))
    return sc


def roofline_rays(n_rays, gen, half):
    """Unit-direction chords from one face of the cube [-half, half]^3 (1e-4 inside) to the opposite
    face, every other coordinate strictly inside: chord length >= 2*half - 2e-4."""
    axis = torch.randint(3, (n_rays,), generator=gen)
    side = torch.randint(2, (n_rays,), generator=gen).float() * 2 - 1
    inset = half * (1 - 2e-3)
    o = (torch.rand((n_rays, 3), generator=gen) * 2 - 1) * inset
    t = (torch.rand((n_rays, 3), generator=gen) * 2 - 1) * inset
    idx = torch.arange(n_rays)
    o[idx, axis] = -side * (half - 1e-4)
    t[idx, axis] = side * (half - 1e-4)
    d = t - o
    d = d / d.norm(dim=-1, keepdim=True)
    return o.contiguous(), d.contiguous()


def roofline_scene(world=160, n_rays=8192, n_samples=256, seed=777, device='cpu', k0_dim=12, bbox=1.5,
                   bound_scale=1.05, stepsize=0.5):
    """The 8192 x 256 roofline case of SURVEY.md section 8d.  Every ray is a unit-direction chord
    from one box face (1e-4 inside) to the opposite face; near = 0, far = (n_samples - 0.5) * stepdist;
    density ~ N(0, 0.1^2) (alpha ~ 0.005 per sample: above the 1e-4 threshold, total opacity far from
    the 1e-3 early-stop), mask all true.  Every ray therefore yields exactly n_samples samples that all
    survive to the feature lookup."""
    gen = torch.Generator().manual_seed(seed)
    half = bbox * bound_scale
    mn = torch.full((3,), -half); mx = torch.full((3,), half)
    ws = (world, world, world)
    voxel_size = ((mx - mn).prod() / world ** 3).pow(1 / 3)
    stepdist = float(stepsize * voxel_size)
    o, d = roofline_rays(n_rays, gen, half)
    far = (n_samples - 0.5) * stepdist
    assert far < 2 * half - 1e-2, 'chord shorter than the sampled span'
    density = torch.randn((1, 1, *ws), generator=gen) * 0.1
    k0 = torch.randn((1, k0_dim, *ws), generator=gen) * 0.3
    mask = torch.ones(ws, dtype=torch.bool)
    target = torch.rand((n_rays, 3), generator=gen)
    sc = dict(xyz_min=mn, xyz_max=mx, density=density, k0=k0, mask=mask, rays_o=o.contiguous(),
              rays_d=d.contiguous(), viewdirs=d.clone().contiguous(), target=target)
    sc = {k: v.to(device) for k, v in sc.items()}
    sc.update(near=0.0, far=far, stepsize=stepsize, world=world, n_samples=n_samples, occupancy=1.0)
    return sc
