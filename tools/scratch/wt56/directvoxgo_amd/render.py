"""Ray generation and chunked full-image rendering (rows N4 / config 5 of SURVEY.md section 8).

  get_rays / ndc_rays / get_rays_of_a_view   /root/reference/lib/ray_utils.py:9-85 (same argument
                                             meaning; built on the device the pose lives on)
  render_viewpoints                          /root/reference/run.py:57-143 without the PNG / metric
                                             side: chunks of rays under no_grad, `render_depth` on
                                             (the reference uses 8192; rays are independent, so the
                                             image is identical for any chunk and 65536 halves the
                                             per-view time on MI355X: 43 -> 21 ms at 800x800);
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
def render_viewpoints(model, render_poses, HW, Ks, ndc, render_kwargs, flip_x=False, flip_y=False, chunk=65536,
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


# ----------------------------------------------------------------------------------------------
# Training-ray gathering (lib/ray_utils.py:88-183, 283-290)
# ----------------------------------------------------------------------------------------------
@torch.no_grad()
def get_training_rays(rgb_tr, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y):
    """Per-image ray tensors [n_img,H,W,3] for same-sized images (lib/ray_utils.py:88-110)."""
    assert len(np.unique(HW, axis=0)) == 1 and len(rgb_tr) == len(train_poses) == len(Ks) == len(HW)
    H, W = int(HW[0][0]), int(HW[0][1])
    dev = rgb_tr.device
    per_view = [get_rays_of_a_view(H, W, Ks[0], torch.as_tensor(np.asarray(c2w), dtype=torch.float32, device=dev),
                                   ndc, inverse_y, flip_x, flip_y) for c2w in train_poses]
    rays_o, rays_d, viewdirs = (torch.stack([v[k] for v in per_view]) for k in range(3))
    return rgb_tr, rays_o, rays_d, viewdirs, [1] * len(rgb_tr)


@torch.no_grad()
def get_training_rays_flatten(rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y):
    """All pixels of all (possibly differently sized) images as flat [N,3] tensors (lib/ray_utils.py:113-142)."""
    assert len(rgb_tr_ori) == len(train_poses) == len(Ks) == len(HW)
    dev = rgb_tr_ori[0].device
    rgb, ro, rd, vd, imsz = [], [], [], [], []
    for c2w, img, (H, W), K in zip(train_poses, rgb_tr_ori, HW, Ks):
        assert tuple(img.shape[:2]) == (int(H), int(W))
        o, d, v = get_rays_of_a_view(int(H), int(W), K, torch.as_tensor(np.asarray(c2w), dtype=torch.float32, device=dev),
                                     ndc, inverse_y, flip_x, flip_y)
        rgb.append(img.flatten(0, 1)); ro.append(o.flatten(0, 1)); rd.append(d.flatten(0, 1)); vd.append(v.flatten(0, 1))
        imsz.append(int(H) * int(W))
    return torch.cat(rgb), torch.cat(ro), torch.cat(rd), torch.cat(vd), imsz


@torch.no_grad()
def get_training_rays_in_maskcache_sampling(rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y, model,
                                            render_kwargs, rows_per_call=64):
    """Only the rays that hit known-occupied space (lib/ray_utils.py:145-183): `model.hit_coarse_geo`
    (sampler + mask lookup kernels) over 64 image rows at a time."""
    assert len(rgb_tr_ori) == len(train_poses) == len(Ks) == len(HW)
    dev = rgb_tr_ori[0].device
    rgb, ro, rd, vd, imsz = [], [], [], [], []
    n_all = 0
    for c2w, img, (H, W), K in zip(train_poses, rgb_tr_ori, HW, Ks):
        o, d, v = get_rays_of_a_view(int(H), int(W), K, torch.as_tensor(np.asarray(c2w), dtype=torch.float32, device=dev),
                                     ndc, inverse_y, flip_x, flip_y)
        hit = torch.cat([model.hit_coarse_geo(rays_o=o[i:i + rows_per_call], rays_d=d[i:i + rows_per_call], **render_kwargs)
                         for i in range(0, int(H), rows_per_call)])
        rgb.append(img[hit]); ro.append(o[hit]); rd.append(d[hit]); vd.append(v[hit])
        imsz.append(int(hit.sum()))
        n_all += int(H) * int(W)
    return torch.cat(rgb), torch.cat(ro), torch.cat(rd), torch.cat(vd), imsz


def batch_indices_generator(N, BS, seed=None):
    """Endless stream of index batches from a NumPy permutation, reshuffled when exhausted
    (lib/ray_utils.py:283-290)."""
    rng = np.random if seed is None else np.random.RandomState(seed)
    idx, top = torch.from_numpy(rng.permutation(N)), 0
    while True:
        if top + BS > N:
            idx, top = torch.from_numpy(rng.permutation(N)), 0
        yield idx[top:top + BS]
        top += BS
