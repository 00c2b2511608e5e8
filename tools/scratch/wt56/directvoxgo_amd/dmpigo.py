"""DirectMPIGO (forward-facing / NDC scenes, BASELINE config 4) on the MI355X kernels.

Host-side counterpart of /root/reference/lib/dmpigo.py:17-290: a multi-plane-image shaped grid
(`mpi_depth` planes along z), a fixed number of samples per ray placed by
``sample_ndc_pts_on_rays`` (K7), ``act_shift = 0`` and ``voxel_size_ratio = 256 / mpi_depth``.
Everything downstream of the sampler is the same op set as DirectVoxGO; the fused path runs the
same four kernels with NDC spacing (include/dvgo_hip.h, `stepdist < 0`).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import render_utils as render_utils_hip
from .dvgo import make_rgbnet, mlp_forward
from .shade import shade, viewdir_embed
from .fused import MarchConfig, composite, composite_depth, fused_march
from .ops import Alphas2Weights, MaskCache, Raw2Alpha, grid_sample, segment_coo, total_variation_add_grad


class DirectMPIGO(nn.Module):
    def __init__(self, xyz_min, xyz_max, num_voxels=0, mpi_depth=0, mask_cache_path=None, mask_cache_thres=1e-3,
                 fast_color_thres=0, rgbnet_dim=0, rgbnet_depth=3, rgbnet_width=128, viewbase_pe=0,
                 channels_last=True, fused=True, **kwargs):
        super().__init__()
        self.fused, self.channels_last = bool(fused), bool(channels_last)
        self.fused_shade = True          # fp32-MFMA colour head (csrc/shade.hip) when the rgbnet has a built shape
        xyz_min = torch.as_tensor(np.asarray(xyz_min, dtype=np.float32))
        xyz_max = torch.as_tensor(np.asarray(xyz_max, dtype=np.float32))
        self.register_buffer('xyz_min', xyz_min.clone())
        self.register_buffer('xyz_max', xyz_max.clone())
        self._xyz_min_cpu, self._xyz_max_cpu = xyz_min.clone(), xyz_max.clone()
        self.fast_color_thres = fast_color_thres
        self.act_shift = 0
        self._set_grid_resolution(num_voxels, mpi_depth)
        ws = [int(v) for v in self.world_size]

        # density initialised so that every plane has the same stop probability (lib/dmpigo.py:35-44)
        self.density = nn.Parameter(torch.zeros([1, 1, *ws]))
        with torch.no_grad():
            g = np.full([mpi_depth], 1. / mpi_depth - 1e-6)
            p = [1 - g[0]]
            for i in range(1, len(g)):
                p.append((1 - g[:i + 1].sum()) / (1 - g[:i].sum()))
            for i in range(len(p)):
                self.density[..., i].fill_(np.log(p[i] ** (-1 / self.voxel_size_ratio) - 1))
            self.density[..., -1].fill_(10)

        self.rgbnet_kwargs = {'rgbnet_dim': rgbnet_dim, 'rgbnet_depth': rgbnet_depth, 'rgbnet_width': rgbnet_width,
                              'viewbase_pe': viewbase_pe}
        if rgbnet_dim <= 0:
            self.k0_dim, self.rgbnet = 3, None
        else:
            self.k0_dim = rgbnet_dim
            self.register_buffer('viewfreq', torch.FloatTensor([(2 ** i) for i in range(viewbase_pe)]))
            self.rgbnet = make_rgbnet((3 + 3 * viewbase_pe * 2) + self.k0_dim, rgbnet_width, rgbnet_depth)
        self.k0 = nn.Parameter(self._alloc_k0(ws))

        self.mask_cache_path, self.mask_cache_thres = mask_cache_path, mask_cache_thres
        if mask_cache_path:
            coarse = MaskCache(path=mask_cache_path, mask_cache_thres=mask_cache_thres)
            dev = torch.device('cuda', torch.cuda.current_device())
            mask = coarse.to(dev)(self._grid_xyz(ws, dev)).cpu()
        else:
            mask = torch.ones(ws, dtype=torch.bool)
        self.mask_cache = MaskCache(path=None, mask=mask, xyz_min=self._xyz_min_cpu, xyz_max=self._xyz_max_cpu)
        self._cfg_cache = {}

    def _alloc_k0(self, ws, device=None):
        g = torch.zeros([1, self.k0_dim, *ws], device=device)
        if self.channels_last and self.k0_dim > 1:
            g = g.contiguous(memory_format=torch.channels_last_3d)
        return g

    def _set_grid_resolution(self, num_voxels, mpi_depth):
        """lib/dmpigo.py:97-107"""
        self.num_voxels, self.mpi_depth = num_voxels, mpi_depth
        ext = self._xyz_max_cpu - self._xyz_min_cpu
        r = (num_voxels / self.mpi_depth / ext[:2].prod()).sqrt()
        self.world_size = torch.zeros(3, dtype=torch.long)
        self.world_size[:2] = (ext[:2] * r).long()
        self.world_size[2] = self.mpi_depth
        self.voxel_size_ratio = 256. / mpi_depth
        self._cfg_cache = {}

    def get_kwargs(self):
        return {'xyz_min': self._xyz_min_cpu.numpy(), 'xyz_max': self._xyz_max_cpu.numpy(),
                'num_voxels': self.num_voxels, 'mpi_depth': self.mpi_depth, 'act_shift': self.act_shift,
                'voxel_size_ratio': self.voxel_size_ratio, 'mask_cache_path': self.mask_cache_path,
                'mask_cache_thres': self.mask_cache_thres, 'fast_color_thres': self.fast_color_thres,
                **self.rgbnet_kwargs}

    def _grid_xyz(self, ws, device):
        return torch.stack(torch.meshgrid(
            *[torch.linspace(float(self._xyz_min_cpu[a]), float(self._xyz_max_cpu[a]), ws[a], device=device)
              for a in range(3)], indexing='ij'), -1)

    @torch.no_grad()
    def scale_volume_grid(self, num_voxels, mpi_depth):
        """lib/dmpigo.py:123-146"""
        self._set_grid_resolution(num_voxels, mpi_depth)
        ws = tuple(int(v) for v in self.world_size)
        self.density = nn.Parameter(F.interpolate(self.density.data, size=ws, mode='trilinear', align_corners=True))
        k0 = F.interpolate(self.k0.data.contiguous(), size=ws, mode='trilinear', align_corners=True)
        if self.channels_last and self.k0_dim > 1:
            k0 = k0.contiguous(memory_format=torch.channels_last_3d)
        self.k0 = nn.Parameter(k0)
        self_alpha = F.max_pool3d(self.activate_density(self.density), kernel_size=3, padding=1, stride=1)[0, 0]
        self.mask_cache = MaskCache(path=None, mask=(self_alpha > self.fast_color_thres).cpu(),
                                    xyz_min=self._xyz_min_cpu, xyz_max=self._xyz_max_cpu).to(self.density.device)
        self._cfg_cache = {}

    def density_total_variation_add_grad(self, weight, dense_mode, x_range=None):
        """lib/dmpigo.py:147-151"""
        wxy = weight * float(self.world_size[:2].max()) / 128
        wz = weight * self.mpi_depth / 128
        total_variation_add_grad(self.density, self.density.grad, wxy, wxy, wz, dense_mode, x_range)

    def k0_total_variation_add_grad(self, weight, dense_mode, x_range=None):
        """lib/dmpigo.py:153-157"""
        wxy = weight * float(self.world_size[:2].max()) / 128
        wz = weight * self.mpi_depth / 128
        total_variation_add_grad(self.k0, self.k0.grad, wxy, wxy, wz, dense_mode, x_range)

    def activate_density(self, density, interval=None):
        interval = interval if interval is not None else self.voxel_size_ratio
        shape = density.shape
        return Raw2Alpha.apply(density.flatten().contiguous(), 0, interval).reshape(shape)

    def grid_sampler(self, xyz, grid):
        return grid_sample(grid, xyz, self.xyz_min, self.xyz_max)

    def n_samples(self, stepsize):
        return int((self.mpi_depth - 1) / stepsize) + 1          # lib/dmpigo.py:188

    def sample_ray(self, rays_o, rays_d, near, far, stepsize, is_train=False, **render_kwargs):
        """lib/dmpigo.py:173-198"""
        assert near == 0 and far == 1
        N_samples = self.n_samples(stepsize)
        ray_pts, mask_outbbox = render_utils_hip.sample_ndc_pts_on_rays(
            rays_o.contiguous(), rays_d.contiguous(), self.xyz_min, self.xyz_max, N_samples)
        mask_inbbox = ~mask_outbbox
        ray_pts = ray_pts[mask_inbbox]
        dev = rays_o.device
        ray_id = torch.arange(mask_inbbox.shape[0], device=dev).view(-1, 1).expand_as(mask_inbbox)[mask_inbbox]
        step_id = torch.arange(mask_inbbox.shape[1], device=dev).view(1, -1).expand_as(mask_inbbox)[mask_inbbox]
        return ray_pts, ray_id, step_id

    def _shade(self, vox_emb, viewdirs, ray_id):
        """lib/dmpigo.py:246-257"""
        if self.rgbnet is None:
            return torch.sigmoid(vox_emb)
        if self.fused and self.fused_shade and viewdirs.is_cuda and viewdirs.dim() == 2:
            rgb = shade(self.rgbnet, vox_emb, viewdir_embed(viewdirs, self.viewfreq), ray_id, diffuse=False)
            if rgb is not None:              # fp32-MFMA colour head (csrc/shade.hip): width 64 / 128, d_in <= 40
                return rgb
        viewdirs_emb = (viewdirs.unsqueeze(-1) * self.viewfreq).flatten(-2)
        viewdirs_emb = torch.cat([viewdirs, viewdirs_emb.sin(), viewdirs_emb.cos()], -1)[ray_id]
        return torch.sigmoid(mlp_forward(self.rgbnet, torch.cat([vox_emb, viewdirs_emb], -1)))

    def forward(self, rays_o, rays_d, viewdirs, global_step=None, **render_kwargs):
        """lib/dmpigo.py:200-283; same result dict as DirectVoxGO.forward."""
        assert len(rays_o.shape) == 2 and rays_o.shape[-1] == 3, 'Only suuport point queries in [N, 3] format'
        N = len(rays_o)
        stepsize, bg = render_kwargs['stepsize'], render_kwargs['bg']
        interval = stepsize * self.voxel_size_ratio
        if self.fused:
            assert render_kwargs['near'] == 0 and render_kwargs['far'] == 1
            key = float(stepsize)
            cfg = self._cfg_cache.get(key)
            if cfg is None or cfg.mask is not self.mask_cache.mask:
                mc = self.mask_cache
                cfg = MarchConfig(self.xyz_min, self.xyz_max, stepdist=1.0, act_shift=0.0, interval=float(interval),
                                  fast_color_thres=self.fast_color_thres, near=0.0, far=1.0, mask=mc.mask,
                                  xyz2ijk_scale=mc.xyz2ijk_scale, xyz2ijk_shift=mc.xyz2ijk_shift,
                                  ndc_samples=self.n_samples(stepsize))
                self._cfg_cache[key] = cfg
            weights, alpha, alphainv_last, vox_emb, ray_id, step_id, off3 = fused_march(
                self.density, self.k0, rays_o, rays_d, cfg)
            rgb = self._shade(vox_emb, viewdirs, ray_id)
            rgb_marched = composite(weights, rgb, alphainv_last, ray_id, off3, bg)
            ret = {'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched,
                   'raw_alpha': alpha, 'raw_rgb': rgb, 'ray_id': ray_id}
            if render_kwargs.get('render_depth', False):
                ret['depth'] = composite_depth(weights.detach(), step_id, off3, N)
            return ret

        ray_pts, ray_id, step_id = self.sample_ray(rays_o=rays_o, rays_d=rays_d,
                                                   is_train=global_step is not None, **render_kwargs)
        if self.mask_cache is not None:
            mask = self.mask_cache(ray_pts)
            ray_pts, ray_id, step_id = ray_pts[mask], ray_id[mask], step_id[mask]
        density = self.grid_sampler(ray_pts, self.density)
        alpha = self.activate_density(density, interval)
        if self.fast_color_thres > 0:
            mask = alpha > self.fast_color_thres
            ray_pts, ray_id, step_id, alpha = ray_pts[mask], ray_id[mask], step_id[mask], alpha[mask]
        weights, alphainv_last = Alphas2Weights.apply(alpha, ray_id, N)
        if self.fast_color_thres > 0:
            mask = weights > self.fast_color_thres
            ray_pts, ray_id, step_id = ray_pts[mask], ray_id[mask], step_id[mask]
            alpha, weights = alpha[mask], weights[mask]
        vox_emb = self.grid_sampler(ray_pts, self.k0)
        rgb = self._shade(vox_emb, viewdirs, ray_id)
        rgb_marched = segment_coo(src=(weights.unsqueeze(-1) * rgb), index=ray_id,
                                  out=torch.zeros([N, 3], device=rays_o.device), reduce='sum')
        rgb_marched = rgb_marched + alphainv_last.unsqueeze(-1) * bg
        ret = {'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched,
               'raw_alpha': alpha, 'raw_rgb': rgb, 'ray_id': ray_id}
        if render_kwargs.get('render_depth', False):
            with torch.no_grad():
                ret['depth'] = segment_coo(src=(weights * step_id), index=ray_id,
                                           out=torch.zeros([N], device=rays_o.device), reduce='sum')
        return ret
