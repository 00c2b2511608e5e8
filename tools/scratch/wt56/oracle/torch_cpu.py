"""Pure-PyTorch CPU restatement of the hot path -- TEST / BASELINE INFRASTRUCTURE, never imported by the product.

The reference has no runnable CPU path (every op of lib/dvgo.py is CUDA-only); what BASELINE.md section 3
calls "the reference's pure-PyTorch CPU fallback" is this restatement, assembled from the pure-PyTorch fragments the
reference still carries:

  slab test + fixed-length uniform sampling + out-of-box mask   /root/reference/lib/dvgo.py:282-289 (voxel_count_views),
                                                                lib/multiscene_dvgo.py:493-515 (sample_ray_py)
  trilinear lookup through F.grid_sample                        lib/dvgo.py:312-328 (grid_sampler)
  softplus form of the activation                               lib/dvgo.py:590, docstring :621-626
  compositing as a cumulative product                           docstring lib/dvgo.py:651-656 (what Alphas2Weights
                                                                replaced), early stop omitted as in that form
  per-ray sums                                                  torch_scatter.segment_coo == index_add on sorted ids
  loss                                                          run.py:377-386
  Adam (plain / masked / per-voxel lr)                          lib/masked_adam.py:39-71 + adam_upd_kernel.cu:8-58

It runs on however many threads torch is given (bench.py: all host cores) and is timed by bench.py's `cpu_baseline`
leg next to the GPU number; tests/test_oracle_golden.py holds it against the C oracle.
"""
import math

import torch
import torch.nn.functional as F


def grid_sampler(grid, xyz, xyz_min, xyz_max):
    """lib/dvgo.py:312-328"""
    shape = xyz.shape[:-1]
    ind_norm = ((xyz.reshape(1, 1, 1, -1, 3) - xyz_min) / (xyz_max - xyz_min)).flip((-1,)) * 2 - 1
    out = F.grid_sample(grid, ind_norm, mode='bilinear', align_corners=True)
    out = out.reshape(grid.shape[1], -1).T.reshape(*shape, grid.shape[1])
    return out.squeeze(-1) if grid.shape[1] == 1 else out


def sample_ray(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist, n_samples):
    """Fixed-length form of K1-K6 (lib/multiscene_dvgo.py:493-515): -> ray_pts [M,3], ray_id [M], step_id [M] of the
    in-box samples."""
    vec = torch.where(rays_d == 0, torch.full_like(rays_d, 1e-6), rays_d)
    rate_a = (xyz_max - rays_o) / vec
    rate_b = (xyz_min - rays_o) / vec
    t_min = torch.minimum(rate_a, rate_b).amax(-1).clamp(min=near, max=far)
    t_max = torch.maximum(rate_a, rate_b).amin(-1).clamp(min=near, max=far)
    rng = torch.arange(n_samples, dtype=torch.float32)[None]
    step = stepdist * rng
    interpx = t_min[..., None] + step / rays_d.norm(dim=-1, keepdim=True)
    rays_pts = rays_o[..., None, :] + rays_d[..., None, :] * interpx[..., None]
    mask_outbbox = ((xyz_min > rays_pts) | (rays_pts > xyz_max)).any(dim=-1) | (interpx > t_max[..., None])
    inb = ~mask_outbbox
    N = rays_o.shape[0]
    ray_id = torch.arange(N).view(-1, 1).expand(N, n_samples)[inb]
    step_id = torch.arange(n_samples).view(1, -1).expand(N, n_samples)[inb]
    return rays_pts[inb], ray_id, step_id


def render(density, k0, rgbnet, viewfreq, rays_o, rays_d, viewdirs, xyz_min, xyz_max, near, far, stepdist, n_samples,
           act_shift, interval, thres, bg, mask=None, rgbnet_direct=True):
    """DirectVoxGO.forward (lib/dvgo.py:450-577) in plain torch ops; returns the reference's result dict."""
    N = rays_o.shape[0]
    ray_pts, ray_id, step_id = sample_ray(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist, n_samples)
    if mask is not None:                                   # nearest-voxel occupancy (lib/dvgo.py:600-611)
        scale = (torch.tensor(mask.shape, dtype=torch.float32) - 1) / (xyz_max - xyz_min)
        ijk = torch.round(ray_pts * scale - xyz_min * scale).long()
        ok = ((ijk >= 0) & (ijk < torch.tensor(mask.shape))).all(-1)
        ijk = ijk.clamp(min=0)
        ijk = torch.minimum(ijk, torch.tensor(mask.shape) - 1)
        keep = ok & mask[ijk[:, 0], ijk[:, 1], ijk[:, 2]]
        ray_pts, ray_id, step_id = ray_pts[keep], ray_id[keep], step_id[keep]
    dens = grid_sampler(density, ray_pts, xyz_min, xyz_max)
    alpha = 1 - torch.exp(-F.softplus(dens + act_shift) * interval)
    if thres > 0:
        keep = alpha > thres
        ray_pts, ray_id, step_id, alpha = ray_pts[keep], ray_id[keep], step_id[keep], alpha[keep]
    # compositing: T_i = prod_{j<i} (1 - alpha_j + 1e-10) per ray, via a dense [N, S] scatter + cumprod
    dense = torch.zeros(N, n_samples + 1)
    dense = dense.index_put((ray_id, step_id + 1), alpha)
    T_all = torch.cumprod(1 - dense + 1e-10, dim=-1)
    T = T_all[ray_id, step_id]
    weights = alpha * T
    alphainv_last = T_all[:, -1]
    if thres > 0:
        keep = weights > thres
        ray_pts, ray_id, step_id, alpha, weights = ray_pts[keep], ray_id[keep], step_id[keep], alpha[keep], weights[keep]
    feat = grid_sampler(k0, ray_pts, xyz_min, xyz_max)
    if rgbnet is None:
        rgb = torch.sigmoid(feat)
    else:
        emb = (viewdirs.unsqueeze(-1) * viewfreq).flatten(-2)
        emb = torch.cat([viewdirs, emb.sin(), emb.cos()], -1)[ray_id]
        if rgbnet_direct:
            rgb = torch.sigmoid(rgbnet(torch.cat([feat, emb], -1)))
        else:
            rgb = torch.sigmoid(rgbnet(torch.cat([feat[:, 3:], emb], -1)) + feat[:, :3])
    rgb_marched = torch.zeros(N, 3).index_add(0, ray_id, weights.unsqueeze(-1) * rgb) + alphainv_last.unsqueeze(-1) * bg
    return {'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched, 'raw_alpha': alpha,
            'raw_rgb': rgb, 'ray_id': ray_id}


def loss_fn(res, target, w_main=1.0, w_ent=0.001, w_per=0.01):
    """run.py:377-386"""
    loss = w_main * F.mse_loss(res['rgb_marched'], target)
    pout = res['alphainv_last'].clamp(1e-6, 1 - 1e-6)
    loss = loss + w_ent * (-(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout)).mean())
    rgbper = (res['raw_rgb'] - target[res['ray_id']]).pow(2).sum(-1)
    return loss + w_per * ((rgbper * res['weights'].detach()).sum() / len(target))


@torch.no_grad()
def adam_step(p, g, m, v, step, lr, mode=0, perlr=None, beta1=0.9, beta2=0.99, eps=1e-8):
    """lib/masked_adam.py:39-71 over adam_upd_kernel.cu:8-58, vectorised: mode 0 plain, 1 masked, 2 per-voxel lr."""
    step_size = lr * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    if mode == 1:
        sel = g != 0
        m_new = beta1 * m + (1 - beta1) * g
        v_new = beta2 * v + (1 - beta2) * g * g
        m.copy_(torch.where(sel, m_new, m)); v.copy_(torch.where(sel, v_new, v))
        p.copy_(torch.where(sel, p - step_size * m / (v.sqrt() + eps), p))
        return
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    upd = m / (v.sqrt() + eps)
    p.sub_(upd * (step_size * perlr) if mode == 2 else upd * step_size)
