"""Ray generation and chunked full-image rendering (rows N4 / config 5 of SURVEY.md section 8).

  get_rays / ndc_rays / get_rays_of_a_view   /root/reference/lib/ray_utils.py:9-85 (same argument
                                             meaning; built on the device the pose lives on)
  render_viewpoints                          /root/reference/run.py:57-143 without the PNG / metric
                                             side: 8192-ray chunks under no_grad, `render_depth` on;
                                             the last chunk may be empty (run.py:91) and is accepted.
Multi-GPU inference (section 8e): images are embarrassingly parallel -- rank r renders poses
r, r+P, ... and the results are gathered; no all-reduce.
"""
import numpy as np
import torch
import torch.distributed as dist


def get_rays(H, W, K, c2w, inverse_y=False, flip_x=False, flip_y=False, mode='center'):
    dev = c2w.device
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W, device=dev), torch.linspace(0, H - 1, H, device=dev),
                          indexing='ij')
    i, j = i.t().float(), j.t().float()
    if mode == 'center':
        i, j = i + 0.5, j + 0.5
    elif mode == 'random':
        i, j = i + torch.rand_like(i), j + torch.rand_like(j)
    elif mode != 'lefttop':
        raise NotImplementedError
    if flip_x:
        i = i.flip((1,))
    if flip_y:
        j = j.flip((0,))
    fx, fy, cx, cy = float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])
    if inverse_y:
        dirs = torch.stack([(i - cx) / fx, (j - cy) / fy, torch.ones_like(i)], -1)
    else:
        dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, 3].expand(rays_d.shape)
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """lib/ray_utils.py:60-77"""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    o0 = -1. / (W / (2. * focal)) * rays_o[..., 0] / rays_o[..., 2]
    o1 = -1. / (H / (2. * focal)) * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1. + 2. * near / rays_o[..., 2]
    d0 = -1. / (W / (2. * focal)) * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = -1. / (H / (2. * focal)) * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2. * near / rays_o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


def get_rays_of_a_view(H, W, K, c2w, ndc, inverse_y, flip_x, flip_y, mode='center'):
    rays_o, rays_d = get_rays(H, W, K, c2w, inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y, mode=mode)
    viewdirs = rays_d / rays_d.norm(dim=-1, keepdim=True)
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, float(K[0][0]), 1., rays_o, rays_d)
    return rays_o, rays_d, viewdirs


@torch.no_grad()
def render_viewpoints(model, render_poses, HW, Ks, ndc, render_kwargs, flip_x=False, flip_y=False, chunk=8192,
                      distributed=False):
    """-> (rgbs [n,H,W,3], depths [n,H,W,1]) as numpy arrays (every rank gets all images when
    ``distributed``)."""
    assert len(render_poses) == len(HW) and len(HW) == len(Ks)
    world = dist.get_world_size() if distributed else 1
    rank = dist.get_rank() if distributed else 0
    dev = next(model.parameters()).device
    kwargs = dict(render_kwargs, render_depth=True)
    mine = {}
    for i in range(rank, len(render_poses), world):
        H, W = int(HW[i][0]), int(HW[i][1])
        c2w = torch.as_tensor(np.asarray(render_poses[i]), dtype=torch.float32, device=dev)
        rays_o, rays_d, viewdirs = get_rays_of_a_view(H, W, Ks[i], c2w, ndc, inverse_y=kwargs.get('inverse_y', False),
                                                      flip_x=flip_x, flip_y=flip_y)
        rays_o, rays_d, viewdirs = (t.flatten(0, -2).contiguous() for t in (rays_o, rays_d, viewdirs))
        out_rgb, out_depth = [], []
        n_chunks = rays_o.shape[0] // chunk + 1                    # run.py:91, last chunk may be empty
        for c in range(n_chunks):
            sl = slice(chunk * c, chunk * (c + 1))
            res = model(rays_o[sl], rays_d[sl], viewdirs[sl], global_step=c, **kwargs)
            out_rgb.append(res['rgb_marched']); out_depth.append(res['depth'])
        mine[i] = (torch.cat(out_rgb).reshape(H, W, 3), torch.cat(out_depth).reshape(H, W, 1))
    if distributed and world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: (a.cpu(), b.cpu()) for k, (a, b) in mine.items()})
        mine = {k: v for part in gathered for k, v in part.items()}
    idx = sorted(mine)
    rgbs = np.stack([mine[i][0].cpu().numpy() for i in idx]) if idx else np.zeros((0,))
    depths = np.stack([mine[i][1].cpu().numpy() for i in idx]) if idx else np.zeros((0,))
    return rgbs, depths
