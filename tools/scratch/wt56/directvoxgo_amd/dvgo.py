"""DirectVoxGO scene model, host side, on the MI355X kernels.

Build-side counterpart of /root/reference/lib/dvgo.py:30-577 for the rows of SURVEY.md
section 8a (H1 sample_ray / hit_coarse_geo, H2 forward): same constructor arguments, same
``state_dict`` keys ('density', 'k0', 'rgbnet.*', 'mask_cache.*', 'xyz_min', 'xyz_max',
'viewfreq'), same ``forward(rays_o, rays_d, viewdirs, global_step, **render_kwargs)`` contract and
result dict, so the training / rendering loops of run.py consume it unchanged.

Two execution paths produce the same dict:
  fused=True   (default) csrc/march.hip: 4 kernels, 1 host sync per forward;
  fused=False  the reference's own op-by-op orchestration on the drop-in ops of
               render_utils.py / ops.py (what a maintainer gets by only swapping the bindings).
The fork-specific LIIF / positional-encoding experiments of the reference model
(implicit_voxel_feat, posbase_pe, rgbnet_full_implicit; lib/dvgo.py:40-41,100-122,329-410) are
outside the north-star path and raise NotImplementedError.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import render_utils as render_utils_hip
from ._lib import _flt, _i64, _int, f3, ptr, stream_of
from .fused import MarchConfig, composite, composite_depth, fused_hit, fused_march
from .ops import Alphas2Weights, MaskCache, Raw2Alpha, grid_sample, segment_coo, total_variation_add_grad
from .shade import shade, viewdir_embed


def _as_f32(x):
    return torch.as_tensor(np.asarray(x, dtype=np.float32) if not isinstance(x, torch.Tensor) else x.detach().cpu(),
                           dtype=torch.float32)


class _LinearSplitK(torch.autograd.Function):
    """y = x @ W^T + b for tall-skinny x [M, K] (M ~ 10^6 samples, K, N <= 128).

    Same maths as nn.Linear; only the weight gradient is evaluated differently: dW = g^T x is a
    reduction over the M samples into a tiny [N, K] output, for which the stock GEMM picks a
    3-ms single-pass kernel at M = 2 M (profiles/r1).  Here the samples are cut into chunks that are
    reduced as one batched GEMM (parallel over chunks) and then summed."""
    CHUNK = 8192

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ weight if ctx.needs_input_grad[0] else None
        M, chunk = x.shape[0], _LinearSplitK.CHUNK
        main = (M // chunk) * chunk
        gw = None
        if main:
            S = main // chunk
            gw = torch.bmm(g[:main].view(S, chunk, -1).transpose(1, 2), x[:main].view(S, chunk, -1)).sum(0)
        if main < M:
            tail = g[main:].t() @ x[main:]
            gw = tail if gw is None else gw + tail
        return gx, gw, g.sum(0)


def mlp_forward(net, x):
    """Run an rgbnet (nn.Sequential of Linear / ReLU / nested Sequential) with the split-K linear for
    large sample counts; identical module tree and parameters."""
    for mod in net:
        if isinstance(mod, nn.Sequential):
            x = mlp_forward(mod, x)
        elif isinstance(mod, nn.Linear) and x.shape[0] >= 4 * _LinearSplitK.CHUNK and x.requires_grad | mod.weight.requires_grad:
            x = _LinearSplitK.apply(x.contiguous(), mod.weight, mod.bias)
        else:
            x = mod(x)
    return x


def make_rgbnet(dim0, width, depth):
    """Same module tree (hence state_dict keys) as lib/dvgo.py:123-131."""
    net = nn.Sequential(
        nn.Linear(dim0, width), nn.ReLU(inplace=True),
        *[nn.Sequential(nn.Linear(width, width), nn.ReLU(inplace=True)) for _ in range(depth - 2)],
        nn.Linear(width, 3))
    nn.init.constant_(net[-1].bias, 0)
    return net


class DirectVoxGO(nn.Module):
    def __init__(self, xyz_min, xyz_max, num_voxels=0, num_voxels_base=0, alpha_init=None,
                 mask_cache_path=None, mask_cache_thres=1e-3, fast_color_thres=0,
                 rgbnet_dim=0, rgbnet_direct=False, rgbnet_full_implicit=False,
                 rgbnet_depth=3, rgbnet_width=128, viewbase_pe=4,
                 posbase_pe=0, implicit_voxel_feat=False,
                 channels_last=True, fused=True, verbose=False, **kwargs):
        super().__init__()
        if posbase_pe > 0 or implicit_voxel_feat or rgbnet_full_implicit:
            raise NotImplementedError('fork-specific LIIF / posbase_pe / full-implicit variants are out of scope')
        self.verbose = verbose
        self.fused = bool(fused)
        self.fused_shade = True          # fp32-MFMA colour head (csrc/shade.hip) when the rgbnet has the default shape
        self.channels_last = bool(channels_last)
        xyz_min, xyz_max = _as_f32(xyz_min), _as_f32(xyz_max)
        self.register_buffer('xyz_min', xyz_min.clone())
        self.register_buffer('xyz_max', xyz_max.clone())
        # host copies: sizing maths runs on the CPU in float32 exactly like the reference's
        # tensor expressions, and never costs a device sync afterwards
        self._xyz_min_cpu, self._xyz_max_cpu = xyz_min.clone(), xyz_max.clone()
        self.fast_color_thres = fast_color_thres

        # lib/dvgo.py:55-62
        self.num_voxels_base = num_voxels_base
        self.voxel_size_base = ((self._xyz_max_cpu - self._xyz_min_cpu).prod() / self.num_voxels_base).pow(1 / 3)
        self.alpha_init = alpha_init
        self.act_shift = np.log(1 / (1 - alpha_init) - 1)
        self._set_grid_resolution(num_voxels)

        ws = [int(v) for v in self.world_size]
        self.density = nn.Parameter(torch.zeros([1, 1, *ws]))
        self.rgbnet_kwargs = {
            'rgbnet_dim': rgbnet_dim, 'rgbnet_direct': rgbnet_direct,
            'rgbnet_full_implicit': rgbnet_full_implicit,
            'rgbnet_depth': rgbnet_depth, 'rgbnet_width': rgbnet_width, 'viewbase_pe': viewbase_pe,
        }
        if rgbnet_dim <= 0:
            self.k0_dim = 3                 # colour grid, coarse stage (lib/dvgo.py:83-87)
            self.rgbnet = None
        else:
            self.k0_dim = rgbnet_dim        # feature grid + shallow MLP (lib/dvgo.py:88-131)
            self.rgbnet_direct = rgbnet_direct
            self.register_buffer('viewfreq', torch.FloatTensor([(2 ** i) for i in range(viewbase_pe)]))
            dim0 = (3 + 3 * viewbase_pe * 2) + (self.k0_dim if rgbnet_direct else self.k0_dim - 3)
            self.rgbnet = make_rgbnet(dim0, rgbnet_width, rgbnet_depth)
        self.k0 = nn.Parameter(self._alloc_k0(ws))

        # occupancy grid (lib/dvgo.py:135-153)
        self.mask_cache_path = mask_cache_path
        self.mask_cache_thres = mask_cache_thres
        if mask_cache_path:
            coarse = MaskCache(path=mask_cache_path, mask_cache_thres=mask_cache_thres)
            mask = self._lookup_on_own_grid(coarse, ws)
        else:
            mask = torch.ones(ws, dtype=torch.bool)
        self.mask_cache = MaskCache(path=None, mask=mask, xyz_min=self._xyz_min_cpu, xyz_max=self._xyz_max_cpu)
        self._cfg_cache = {}

    # ------------------------------------------------------------------ sizing / bookkeeping
    def _alloc_k0(self, ws, device=None):
        g = torch.zeros([1, self.k0_dim, *ws], device=device)
        if self.channels_last and self.k0_dim > 1:
            g = g.contiguous(memory_format=torch.channels_last_3d)
        return g

    def _set_grid_resolution(self, num_voxels):
        """lib/dvgo.py:155-165 (float32 tensor maths on the host)."""
        self.num_voxels = num_voxels
        ext = self._xyz_max_cpu - self._xyz_min_cpu
        self.voxel_size = (ext.prod() / num_voxels).pow(1 / 3)
        self.world_size = (ext / self.voxel_size).long()
        self.voxel_size_ratio = self.voxel_size / self.voxel_size_base
        self._cfg_cache = {}
        if self.verbose:
            print('dvgo_amd: world_size', self.world_size.tolist(), 'voxel_size', float(self.voxel_size),
                  'voxel_size_ratio', float(self.voxel_size_ratio))

    def get_kwargs(self):
        return {
            'xyz_min': self._xyz_min_cpu.numpy(), 'xyz_max': self._xyz_max_cpu.numpy(),
            'num_voxels': self.num_voxels, 'num_voxels_base': self.num_voxels_base,
            'alpha_init': self.alpha_init, 'act_shift': self.act_shift,
            'voxel_size_ratio': self.voxel_size_ratio,
            'mask_cache_path': self.mask_cache_path, 'mask_cache_thres': self.mask_cache_thres,
            'fast_color_thres': self.fast_color_thres,
            **self.rgbnet_kwargs,
        }

    def _grid_xyz(self, ws, device):
        return torch.stack(torch.meshgrid(
            torch.linspace(float(self._xyz_min_cpu[0]), float(self._xyz_max_cpu[0]), ws[0], device=device),
            torch.linspace(float(self._xyz_min_cpu[1]), float(self._xyz_max_cpu[1]), ws[1], device=device),
            torch.linspace(float(self._xyz_min_cpu[2]), float(self._xyz_max_cpu[2]), ws[2], device=device),
            indexing='ij'), -1)

    def _lookup_on_own_grid(self, coarse, ws):
        """Evaluate a coarse MaskCache at this model's voxel centres (lib/dvgo.py:143-148).
        Needs the GPU (the lookup is a HIP op)."""
        dev = torch.device('cuda', torch.cuda.current_device())
        return coarse.to(dev)(self._grid_xyz(ws, dev)).cpu()

    # ------------------------------------------------------------------ grid maintenance (N4)
    @torch.no_grad()
    def maskout_near_cam_vox(self, cam_o, near):
        """lib/dvgo.py:215-226: density = -100 wherever a training camera is within `near` (one kernel over the voxels,
        csrc/maintain.hip; the voxel centres are the reference's torch.linspace coordinates)."""
        dev = self.density.device
        X, Y, Z = (int(v) for v in self.density.shape[2:])
        gx, gy, gz = (torch.linspace(float(self._xyz_min_cpu[a]), float(self._xyz_max_cpu[a]), n, device=dev)
                      for a, n in enumerate((X, Y, Z)))
        cams = torch.as_tensor(cam_o, dtype=torch.float32).reshape(-1, 3).to(dev).contiguous()
        with L.device_of(self.density):
            L.call('dvgo_maskout_near_cam', ptr(self.density), ptr(gx), ptr(gy), ptr(gz), _int(X), _int(Y), _int(Z),
                   ptr(cams), _int(cams.shape[0]), _flt(float(near)), _flt(-100.0), stream_of(self.density))

    @torch.no_grad()
    def scale_volume_grid(self, num_voxels):
        """Progressive up-scaling (lib/dvgo.py:228-263)."""
        self._set_grid_resolution(num_voxels)
        ws = tuple(int(v) for v in self.world_size)
        self.density = nn.Parameter(F.interpolate(self.density.data, size=ws, mode='trilinear', align_corners=True))
        if self.k0_dim > 0:
            k0 = F.interpolate(self.k0.data.contiguous(), size=ws, mode='trilinear', align_corners=True)
            if self.channels_last and self.k0_dim > 1:
                k0 = k0.contiguous(memory_format=torch.channels_last_3d)
            self.k0 = nn.Parameter(k0)
        else:
            self.k0 = nn.Parameter(self._alloc_k0(ws, device=self.density.device))
        self_alpha = F.max_pool3d(self.activate_density(self.density), kernel_size=3, padding=1, stride=1)[0, 0]
        mask = self_alpha > self.fast_color_thres
        if self.mask_cache_path:
            coarse = MaskCache(path=self.mask_cache_path, mask_cache_thres=self.mask_cache_thres).to(self.density.device)
            mask = coarse(self._grid_xyz(ws, self.density.device)) & mask
        self.mask_cache = MaskCache(path=None, mask=mask.cpu(), xyz_min=self._xyz_min_cpu,
                                    xyz_max=self._xyz_max_cpu).to(self.density.device)
        self._cfg_cache = {}

    @torch.no_grad()
    def voxel_count_views(self, rays_o_tr, rays_d_tr, imsz, near, far, stepsize, downrate=1, irregular_shape=False):
        """How many training views see each voxel (lib/dvgo.py:265-295; drives the coarse stage's per-voxel learning
        rate, run.py:311-320).  The reference pushes ones through grid_sample and reads `ones.grad > 1` per view; here
        the per-view weight sums are accumulated by one kernel (one wavefront per ray) and committed by another
        (csrc/maintain.hip) -- same argument meaning, returns count [1,1,X,Y,Z] float."""
        dev = self.density.device
        X, Y, Z = (int(v) for v in self.density.shape[2:])
        n_samples = int(np.linalg.norm(np.array([X, Y, Z]) + 1) / stepsize) + 1
        step = float(np.float32(stepsize) * self.voxel_size.numpy().astype(np.float32))
        count = torch.zeros_like(self.density.detach())
        acc = torch.zeros(X * Y * Z, dtype=torch.float32, device=dev)
        mn, mx = f3(self._xyz_min_cpu), f3(self._xyz_max_cpu)
        with L.device_of(self.density):
            st = stream_of(self.density)
            for rays_o_, rays_d_ in zip(rays_o_tr.split(imsz), rays_d_tr.split(imsz)):
                if not irregular_shape:
                    rays_o_, rays_d_ = rays_o_[::downrate, ::downrate], rays_d_[::downrate, ::downrate]
                ro = rays_o_.to(dev).reshape(-1, 3).float().contiguous()
                rd = rays_d_.to(dev).reshape(-1, 3).float().contiguous()
                L.call('dvgo_view_weight_accumulate', ptr(ro), ptr(rd), _i64(ro.shape[0]), mn, mx, _flt(float(near)),
                       _flt(float(far)), _flt(step), _int(n_samples), _int(X), _int(Y), _int(Z), ptr(acc), st)
                L.call('dvgo_view_count_commit', ptr(acc), ptr(count), _i64(X * Y * Z), st)
        return count

    def density_total_variation_add_grad(self, weight, dense_mode, x_range=None):
        """lib/dvgo.py:297-300"""
        w = weight * float(self.world_size.max()) / 128
        total_variation_add_grad(self.density, self.density.grad, w, w, w, dense_mode, x_range)

    def k0_total_variation_add_grad(self, weight, dense_mode, x_range=None):
        """lib/dvgo.py:302-305"""
        w = weight * float(self.world_size.max()) / 128
        total_variation_add_grad(self.k0, self.k0.grad, w, w, w, dense_mode, x_range)

    # ------------------------------------------------------------------ op wrappers
    def activate_density(self, density, interval=None):
        """lib/dvgo.py:307-310"""
        interval = interval if interval is not None else self.voxel_size_ratio
        shape = density.shape
        return Raw2Alpha.apply(density.flatten().contiguous(), self.act_shift, interval).reshape(shape)

    def grid_sampler(self, xyz, *grids, **_unused):
        """lib/dvgo.py:312-328 (bilinear branch)."""
        ret = [grid_sample(g, xyz, self.xyz_min, self.xyz_max) for g in grids]
        return ret[0] if len(ret) == 1 else ret

    def hit_coarse_geo(self, rays_o, rays_d, near, far, stepsize, **render_kwargs):
        """Rays with at least one sample in known-occupied space (lib/dvgo.py:412-423)."""
        shape = rays_o.shape[:-1]
        rays_o = rays_o.reshape(-1, 3).contiguous()
        rays_d = rays_d.reshape(-1, 3).contiguous()
        if self.fused and self.mask_cache is not None:
            return fused_hit(rays_o, rays_d, self._march_cfg(near, far, stepsize)).reshape(shape)
        stepdist = stepsize * self.voxel_size
        ray_pts, mask_outbbox, ray_id = render_utils_hip.sample_pts_on_rays(
            rays_o, rays_d, self.xyz_min, self.xyz_max, near, far, stepdist)[:3]
        mask_inbbox = ~mask_outbbox
        hit = torch.zeros([len(rays_o)], dtype=torch.bool, device=rays_o.device)
        hit[ray_id[mask_inbbox][self.mask_cache(ray_pts[mask_inbbox])]] = 1
        return hit.reshape(shape)

    def sample_ray(self, rays_o, rays_d, near, far, stepsize, is_train=0, **render_kwargs):
        """lib/dvgo.py:425-448 -> (ray_pts, ray_id, step_id) of the in-box samples, near to far."""
        rays_o = rays_o.contiguous()
        rays_d = rays_d.contiguous()
        stepdist = stepsize * self.voxel_size
        ray_pts, mask_outbbox, ray_id, step_id, N_steps, t_min, t_max = render_utils_hip.sample_pts_on_rays(
            rays_o, rays_d, self.xyz_min, self.xyz_max, near, far, stepdist)
        mask_inbbox = ~mask_outbbox
        return ray_pts[mask_inbbox], ray_id[mask_inbbox], step_id[mask_inbbox]

    # ------------------------------------------------------------------ colour head
    def _shade(self, k0, viewdirs, ray_id, m_dev=None):
        """lib/dvgo.py:512-541 (bilinear / non-implicit branches)."""
        if self.rgbnet is None:
            return torch.sigmoid(k0)
        if self.rgbnet_direct:
            k0_view = k0
        else:
            k0_view = k0[:, 3:]
            k0_diffuse = k0[:, :3]
        if self.fused and self.fused_shade and viewdirs.is_cuda and viewdirs.dim() == 2:
            rgb = shade(self.rgbnet, k0, viewdir_embed(viewdirs, self.viewfreq), ray_id, diffuse=not self.rgbnet_direct,
                        m_dev=m_dev)
            if rgb is not None:
                return rgb
        assert m_dev is None, 'capacity mode needs the fused colour head'      # (torch ops would run over undefined rows)
        viewdirs_emb = (viewdirs.unsqueeze(-1) * self.viewfreq).flatten(-2)
        viewdirs_emb = torch.cat([viewdirs, viewdirs_emb.sin(), viewdirs_emb.cos()], -1)
        viewdirs_emb = viewdirs_emb.flatten(0, -2)[ray_id]
        rgb_logit = mlp_forward(self.rgbnet, torch.cat([k0_view, viewdirs_emb], -1))
        if self.rgbnet_direct:
            return torch.sigmoid(rgb_logit)
        return torch.sigmoid(rgb_logit + k0_diffuse)

    # ------------------------------------------------------------------ forward (H2)
    def _march_cfg(self, near, far, stepsize):
        key = (float(near), float(far), float(stepsize))
        cfg = self._cfg_cache.get(key)
        if cfg is None or cfg.mask is not (self.mask_cache.mask if self.mask_cache is not None else None):
            mc = self.mask_cache
            cfg = MarchConfig(self.xyz_min, self.xyz_max, stepdist=float(stepsize * self.voxel_size),
                              act_shift=self.act_shift, interval=float(stepsize * self.voxel_size_ratio),
                              fast_color_thres=self.fast_color_thres, near=near, far=far,
                              mask=None if mc is None else mc.mask,
                              xyz2ijk_scale=None if mc is None else mc.xyz2ijk_scale,
                              xyz2ijk_shift=None if mc is None else mc.xyz2ijk_shift)
            self._cfg_cache[key] = cfg
        return cfg

    def forward(self, rays_o, rays_d, viewdirs, global_step=None, **render_kwargs):
        """Volume rendering (lib/dvgo.py:450-577).  Returns the reference's dict:
        alphainv_last [N], weights [M], rgb_marched [N,3], raw_alpha [M], raw_rgb [M,3], ray_id [M]
        (+ depth [N] when render_kwargs['render_depth'])."""
        assert len(rays_o.shape) == 2 and rays_o.shape[-1] == 3, 'Only suuport point queries in [N, 3] format'
        if self.fused:
            return self._forward_fused(rays_o, rays_d, viewdirs, **render_kwargs)
        return self._forward_unfused(rays_o, rays_d, viewdirs, global_step, **render_kwargs)

    def can_keep_count_on_device(self):
        """True when `forward(..., _capacity=True)` is available: the fused march with the fused colour head."""
        from .shade import head_layers
        return bool(self.fused and self.fused_shade and self.rgbnet is not None and head_layers(self.rgbnet) is not None)

    def _forward_fused(self, rays_o, rays_d, viewdirs, near, far, stepsize, bg, render_depth=False, _capacity=False,
                       **_unused):
        """`_capacity` (training step only, train.py): no host synchronisation -- the per-sample outputs are allocated at
        their upper bound, only their first `ret['n_samples']` rows (a device scalar) are defined, and every kernel
        downstream reads that count from the device."""
        N = len(rays_o)
        cfg = self._march_cfg(near, far, stepsize)
        _capacity = bool(_capacity) and self.can_keep_count_on_device() and viewdirs.is_cuda and viewdirs.dim() == 2
        weights, alpha, alphainv_last, k0, ray_id, step_id, off3 = fused_march(
            self.density, self.k0, rays_o, rays_d, cfg, capacity=_capacity)
        m_dev = off3[N:] if _capacity else None
        rgb = self._shade(k0, viewdirs, ray_id, m_dev)
        rgb_marched = composite(weights, rgb, alphainv_last, ray_id, off3, bg, m_dev)
        ret = {'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched,
               'raw_alpha': alpha, 'raw_rgb': rgb, 'ray_id': ray_id}
        if _capacity:
            ret['n_samples'] = m_dev
        if render_depth:
            ret['depth'] = composite_depth(weights.detach(), step_id, off3, N)
        return ret

    def _forward_unfused(self, rays_o, rays_d, viewdirs, global_step=None, **render_kwargs):
        """The reference's op sequence (lib/dvgo.py:458-577) on the drop-in ops."""
        N = len(rays_o)
        ray_pts, ray_id, step_id = self.sample_ray(rays_o=rays_o, rays_d=rays_d,
                                                   is_train=global_step is not None, **render_kwargs)
        interval = render_kwargs['stepsize'] * self.voxel_size_ratio
        if self.mask_cache is not None:                      # skip known free space
            mask = self.mask_cache(ray_pts)
            ray_pts, ray_id, step_id = ray_pts[mask], ray_id[mask], step_id[mask]
        density = self.grid_sampler(ray_pts, self.density)  # post-activated alpha
        alpha = self.activate_density(density, interval)
        if self.fast_color_thres > 0:
            mask = alpha > self.fast_color_thres
            ray_pts, ray_id, step_id = ray_pts[mask], ray_id[mask], step_id[mask]
            alpha = alpha[mask]
        weights, alphainv_last = Alphas2Weights.apply(alpha, ray_id, N)
        if self.fast_color_thres > 0:
            mask = weights > self.fast_color_thres
            weights, alpha = weights[mask], alpha[mask]
            ray_pts, ray_id, step_id = ray_pts[mask], ray_id[mask], step_id[mask]
        k0 = self.grid_sampler(ray_pts, self.k0)
        rgb = self._shade(k0, viewdirs, ray_id)
        rgb_marched = segment_coo(src=(weights.unsqueeze(-1) * rgb), index=ray_id,
                                  out=torch.zeros([N, 3], device=rays_o.device), reduce='sum')
        rgb_marched = rgb_marched + alphainv_last.unsqueeze(-1) * render_kwargs['bg']
        ret = {'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched,
               'raw_alpha': alpha, 'raw_rgb': rgb, 'ray_id': ray_id}
        if render_kwargs.get('render_depth', False):
            with torch.no_grad():
                ret['depth'] = segment_coo(src=(weights * step_id), index=ray_id,
                                           out=torch.zeros([N], device=rays_o.device), reduce='sum')
        return ret
