"""Fused ray march: the MI355X-native replacement of the op-by-op pipeline in
DirectVoxGO.forward (/root/reference/lib/dvgo.py:450-577).

Two autograd Functions wrap the kernels of csrc/march.hip:

  fused_march(...)   sampling + mask + density interp + activation + compositing weights +
                     both threshold filters + feature interp, returning exactly the tensors the
                     reference has after its 4th boolean compaction (lib/dvgo.py:488-509);
                     backward scatters into the density and feature grids.
  composite(...)     per-ray weighted colour sum + background (lib/dvgo.py:554-559), optional
                     depth (lib/dvgo.py:569-576).

One host sync per forward (the survivor count M3, needed to size the outputs); the reference
needs five.
"""
import ctypes
import math

import torch

from . import _lib as L
from ._lib import _flt, _i64, _int, check_f32, check_input, f3, ptr, stream_of
from .ops import _grid_geom

# scratch budget for fixed-stride records (16 B per potential sample)
_MAX_STRIDE_SCRATCH_BYTES = 4 << 30


class MarchConfig:
    """Host-side constants of one model/render setting (model constants travel to the fused
    kernels as kernel arguments, so they are kept as host arrays here)."""

    def __init__(self, xyz_min, xyz_max, stepdist, act_shift, interval, fast_color_thres, near, far,
                 mask=None, xyz2ijk_scale=None, xyz2ijk_shift=None, ndc_samples=0):
        self.xyz_min_t, self.xyz_max_t = xyz_min, xyz_max          # device tensors (prepare kernel)
        self.xyz_min_h, self.xyz_max_h = f3(xyz_min), f3(xyz_max)  # host copies (fused kernels)
        self.stepdist = float(stepdist)
        self.act_shift = float(act_shift)
        self.interval = float(interval)
        self.thres = float(fast_color_thres)
        self.near, self.far = float(near), float(far)
        # > 0: forward-facing / MPI sampling (lib/dmpigo.py:173-198): every ray has exactly ndc_samples
        # samples at o + d * s/(ndc_samples-1); encoded for the kernels as stepdist = -(ndc_samples-1)
        self.ndc_samples = int(ndc_samples)
        if self.ndc_samples > 0:
            self.stepdist = -float(self.ndc_samples - 1)
        self.mask = mask
        self.scale_h = f3(xyz2ijk_scale) if mask is not None else f3([0, 0, 0])
        self.shift_h = f3(xyz2ijk_shift) if mask is not None else f3([0, 0, 0])


def _rec_stride(cfg, n_rays):
    """Upper bound of N_steps: t_min,t_max are clamped to [near,far] (render_utils_kernel.cu:32-33)
    so N_steps <= ceil((far-near)/stepdist) (+1 for the division rounding)."""
    span = (cfg.far - cfg.near) / cfg.stepdist
    if not math.isfinite(span) or span < 0:
        return 0
    stride = int(math.ceil(span)) + 2
    if stride * n_rays * 16 > _MAX_STRIDE_SCRATCH_BYTES:
        return 0
    return stride


# Owner-computes gradient scatter (csrc/brick.hip): samples are listed per 8x8x8 brick, one workgroup sums a brick;
# no float atomics, no zero-fill.  The default whenever both grids want a gradient, share the lattice and the
# feature grid is channels-last with a built channel count; the atomic scatters below remain as A/B variants.
BRICK_SLICE = None          # entries per work item of the brick kernel (None: the library's default)
BRICK_SCATTER = True
BRICK_CHANNELS = (3, 4, 9, 12)

# A/B switch (tests, tools): scatter both grid gradients through one buffer of 64-byte voxel rows
COMBINED_GRID_GRAD = True
COMBINED_MIN_RATIO = 6          # use it when kept samples * ratio >= voxels (tests set 1e9 to force it)


class grid_rows_capture:
    """Context manager for a training step that owns the optimizer: inside it the march backward does NOT produce
    `k0.grad` / `density.grad` (those two stay `None`).  Either
      * `adam` is given (a callable returning the argument tail of dvgo_brick_accumulate, see
        `MaskedAdam.grid_step_args`) and the brick scatter applies the masked Adam update of both grids in place, from
        the brick's gradient tile in LDS -- the gradient never exists in memory (`.stepped` is set); or
      * the atomic scatter into combined gradient rows hands them over (`.G`: [n_vox, 16] = 12 feature channels, the
        density gradient, pad) so that the optimizer can update both grids straight from the rows
        (`MaskedAdam.step_grid_rows`).
    Only for the (density, k0) parameters given."""
    _active = None

    def __init__(self, density, k0, adam=None):
        self.density, self.k0, self.G, self.adam, self.stepped = density, k0, None, adam, False

    def __enter__(self):
        self.G, self.stepped = None, False
        grid_rows_capture._active = self
        return self

    def __exit__(self, *exc):
        grid_rows_capture._active = None
        return False


def split_grid_rows(G, density, k0):
    """Combined gradient rows -> (density.grad, k0.grad) in the parameters' own layouts."""
    gk = torch.empty_like(k0, memory_format=torch.preserve_format)
    gd = torch.empty_like(density)
    with L.device_of(G):
        L.call('dvgo_grid_grad_split', ptr(G), _i64(density.numel()), _int(16), _int(k0.shape[1]), ptr(gk), ptr(gd), stream_of(G))
    return gd, gk


class _FusedMarch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, density, k0, rays_o, rays_d, cfg, capacity=False):
        for x, n in ((rays_o, 'rays_o'), (rays_d, 'rays_d')):
            check_input(x, n); check_f32(x, n)
        if not (density.is_cuda and k0.is_cuda):
            raise RuntimeError('density must be a CUDA tensor')
        check_f32(density, 'density'); check_f32(k0, 'k0')
        if not density.is_contiguous():
            raise RuntimeError('density must be contiguous')
        dev = rays_o.device
        N = rays_o.shape[0]
        _, X, Y, Z, _, _, _, _ = _grid_geom(density)
        C, kX, kY, kZ, sC, sX, sY, sZ = _grid_geom(k0)
        assert (kX, kY, kZ) == (X, Y, Z), 'density and k0 must share world_size'
        st = stream_of(rays_o)
        ndc = cfg.ndc_samples > 0
        stride = cfg.ndc_samples if ndc else _rec_stride(cfg, N)

        if ndc:
            n_steps = torch.full((N,), cfg.ndc_samples, dtype=torch.int64, device=dev)
            start, dirs = rays_o, rays_d
        else:
            t_min = torch.empty(N, dtype=torch.float32, device=dev)
            t_max = torch.empty_like(t_min)
            n_steps = torch.empty(N, dtype=torch.int64, device=dev)
            start = torch.empty((N, 3), dtype=torch.float32, device=dev)
            dirs = torch.empty((N, 3), dtype=torch.float32, device=dev)
        cum = torch.empty(N, dtype=torch.int64, device=dev) if stride == 0 else None
        n2 = torch.empty(N, dtype=torch.int32, device=dev)
        n3 = torch.empty(N, dtype=torch.int32, device=dev)
        last = torch.empty(N, dtype=torch.float32, device=dev)
        off3 = torch.empty(N + 1, dtype=torch.int64, device=dev)
        with L.device_of(rays_o):
            if not ndc:
                L.call('dvgo_sample_pts_prepare', ptr(rays_o), ptr(rays_d), ptr(cfg.xyz_min_t), ptr(cfg.xyz_max_t),
                       _flt(cfg.near), _flt(cfg.far), _flt(cfg.stepdist), _i64(N), ptr(t_min), ptr(t_max),
                       ptr(n_steps), ptr(cum), ptr(start), ptr(dirs), st)
            if stride == 0:
                cap = int(cum[-1].item()) if N > 0 else 0     # exact layout: one extra host read
            else:
                cap = stride * N
            rec2 = torch.empty((max(cap, 1), 4), dtype=torch.float32, device=dev)
            # training: count, per 8^3 brick, the samples the backward will list for it (csrc/brick.hip)
            bricks = (BRICK_SCATTER and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and C in BRICK_CHANNELS
                      and tuple(density.shape[2:]) == (X, Y, Z) and (sC, sZ, sY, sX) == (1, C, Z * C, Y * Z * C)
                      and k0.data_ptr() % 16 == 0 and N > 0 and 8 * cap < 1 << 31)       # int32 list offsets
            brick_cnt = brick_off = brick_cur = extra_brick = None
            n_extra_max = slice_len = 0
            if bricks:
                nb = L.lib().dvgo_n_bricks(X, Y, Z)
                brick_cnt = torch.zeros(nb, dtype=torch.int32, device=dev)
                brick_off = torch.empty((3, nb + 1), dtype=torch.int32, device=dev)    # list offsets, extra items, non-empty bricks
                brick_cur = torch.empty(nb, dtype=torch.int32, device=dev)
                # heavy bricks: extra work items <= entries / slice, entries <= 8 per record slot (the slice tables are
                # built for up to 2^28 entries; render-sized batches beyond that run one workgroup per brick)
                if 8 * cap < 1 << 28:
                    slice_len = BRICK_SLICE or L.lib().dvgo_brick_slice()
                    n_extra_max = 8 * max(cap, 1) // slice_len
                    extra_brick = torch.empty(max(n_extra_max, 1), dtype=torch.int32, device=dev)
            mask = cfg.mask
            mshape = mask.shape if mask is not None else (0, 0, 0)
            L.call('dvgo_march_density', ptr(start), ptr(dirs), ptr(n_steps), ptr(cum), _i64(stride), _i64(N),
                   cfg.xyz_min_h, cfg.xyz_max_h, _flt(cfg.stepdist), ptr(mask), _int(mshape[0]), _int(mshape[1]),
                   _int(mshape[2]), cfg.scale_h, cfg.shift_h, ptr(density), _int(X), _int(Y), _int(Z),
                   _flt(cfg.act_shift), _flt(cfg.interval), _flt(cfg.thres), ptr(rec2), ptr(n2), ptr(n3),
                   ptr(last), ptr(brick_cnt), st)
            n_entries = 0
            if N <= 16384:              # both scans in one launch (one workgroup each)
                L.call('dvgo_march_scans', ptr(n3), _i64(N), ptr(off3), ptr(brick_cnt), _int(nb if bricks else 0),
                       ptr(brick_off[0] if bricks else None), ptr(brick_cur),
                       ptr(brick_off[1] if extra_brick is not None else None),
                       ptr(brick_off[2] if extra_brick is not None else None), ptr(extra_brick), _int(n_extra_max),
                       _int(slice_len), st)
            else:                       # render-sized batches: the multi-workgroup scan
                L.call('dvgo_exclusive_scan_i32', ptr(n3), _i64(N), ptr(off3), st)
                if bricks:
                    L.call('dvgo_brick_scan', ptr(brick_cnt), _int(nb), ptr(brick_off[0]), ptr(brick_cur),
                           ptr(brick_off[1] if extra_brick is not None else None),
                           ptr(brick_off[2] if extra_brick is not None else None), ptr(extra_brick), _int(n_extra_max),
                           _int(slice_len), st)
            if capacity and stride > 0 and bricks:
                # training step (train.py): the surviving-sample count stays on the device.  The outputs are sized by
                # their upper bound -- every step of every ray -- and every consumer is handed off3[N] as a device
                # pointer (`m_dev`): no host synchronisation in the forward at all.  Rows past the count are garbage.
                M3 = stride * N
                n_entries = 8 * M3                              # a sample touches at most 2 x 2 x 2 bricks
            elif bricks:
                M3, n_entries = torch.stack((off3[-1], brick_off[0, -1].long())).tolist()   # the one host sync
            else:
                M3 = int(off3[-1].item())                      # the one host sync of the fused forward
            ray_id = torch.empty(M3, dtype=torch.int64, device=dev)
            step_id = torch.empty(M3, dtype=torch.int64, device=dev)
            weights = torch.empty(M3, dtype=torch.float32, device=dev)
            alpha = torch.empty(M3, dtype=torch.float32, device=dev)
            feat = torch.empty((M3, C), dtype=torch.float32, device=dev)
            L.call('dvgo_march_gather', ptr(rec2), ptr(n2), ptr(n_steps), ptr(cum), _i64(stride), ptr(off3), _i64(N), _i64(M3),
                   ptr(start), ptr(dirs), _flt(cfg.stepdist), cfg.xyz_min_h, cfg.xyz_max_h, ptr(k0), _int(C),
                   _int(X), _int(Y), _int(Z), _i64(sC), _i64(sX), _i64(sY), _i64(sZ), ptr(ray_id), ptr(step_id),
                   ptr(weights), ptr(alpha), ptr(feat), st)
        ctx.cfg = cfg
        ctx.geom = (X, Y, Z, C, sC, sX, sY, sZ, stride, N)
        ctx.bricks = (brick_off, brick_cur, brick_cnt, extra_brick, slice_len, n_entries) if bricks else None
        ctx.padded = bool(capacity and stride > 0 and bricks)
        ctx.density_meta, ctx.k0_meta = density, k0
        ctx.save_for_backward(rec2, n2, n_steps, cum if cum is not None else n_steps, off3, start, dirs, last,
                              ray_id, step_id)
        ctx.mark_non_differentiable(alpha, ray_id, step_id, off3)
        ctx.set_materialize_grads(False)     # no zero-filled [M3] int64 'gradients' for the id outputs (33 MB per step)
        return weights, alpha, last, feat, ray_id, step_id, off3

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_w, _g_alpha, g_last, g_feat, _g_rid, _g_sid, _g_off):
        rec2, n2, n_steps, cum, off3, start, dirs, last, ray_id, step_id = ctx.saved_tensors
        X, Y, Z, C, sC, sX, sY, sZ, stride, N = ctx.geom
        cfg = ctx.cfg
        cum_p = ptr(cum) if stride == 0 else ptr(None)
        M3 = ray_id.shape[0]
        st = stream_of(start)
        grad_density = grad_k0 = None
        dev = start.device
        with L.device_of(start):
            want_k0 = ctx.needs_input_grad[1] and g_feat is not None and C > 0
            want_d = ctx.needs_input_grad[0]
            gw = gl = None
            if want_d:
                gw = g_w.contiguous() if g_w is not None else torch.zeros(M3, dtype=torch.float32, device=dev)
                gl = g_last.contiguous() if g_last is not None else None

            def density_bwd(dst, dst_stride, kept, cursor=None, recs=None):
                L.call('dvgo_march_density_bwd', ptr(rec2), ptr(n2), ptr(n_steps), cum_p, _i64(stride), ptr(off3),
                       _i64(N), ptr(start), ptr(dirs), _flt(cfg.stepdist), cfg.xyz_min_h, cfg.xyz_max_h, ptr(last),
                       _flt(cfg.interval), ptr(gw), ptr(gl), _int(X), _int(Y), _int(Z), ptr(dst), _i64(dst_stride),
                       ptr(kept), ptr(cursor), ptr(recs), st)

            if ctx.padded and not (ctx.bricks is not None and want_k0 and want_d and BRICK_SCATTER):
                raise RuntimeError('capacity-mode forward (device-side sample count) needs the brick scatter backward for both grids')
            if ctx.bricks is not None and want_k0 and want_d and BRICK_SCATTER:
                # owner-computes scatter: list every sample per brick, then one workgroup sums each brick
                brick_off, brick_cur, arrive, extra_brick, slice_len, E = ctx.bricks
                ctx.bricks = None                               # the fill cursors are consumed: one backward per forward
                recs = torch.empty((max(E, 1), 4), dtype=torch.int32, device=dev)
                density_bwd(None, 1, None, brick_cur, recs)
                g_feat = g_feat.contiguous()
                # heavy bricks run as several work items (slices of the list) that meet in scratch tiles
                if extra_brick is not None:
                    n_extra_max = min(extra_brick.shape[0], E // slice_len)
                    tiles = torch.empty((2 * n_extra_max + 1, 512 * ((C + 4) // 4 * 4)), dtype=torch.float32, device=dev)
                    items = (ptr(brick_off[0]), ptr(brick_off[1]), ptr(brick_off[2]), ptr(extra_brick), ptr(arrive),
                             ptr(tiles), _i64(n_extra_max), _int(slice_len))
                else:
                    items = (ptr(brick_off[0]), ptr(None), ptr(None), ptr(None), ptr(None), ptr(None), _i64(0), _int(0))
                cap = grid_rows_capture._active
                fuse = (cap is not None and cap.adam is not None and not cap.stepped and cap.density is ctx.density_meta
                        and cap.k0 is ctx.k0_meta)
                if fuse:
                    tail = cap.adam()
                    L.call('dvgo_brick_accumulate', *items, ptr(recs), ptr(start), ptr(dirs), _flt(cfg.stepdist),
                           cfg.xyz_min_h, cfg.xyz_max_h, ptr(g_feat), _int(C), _int(X), _int(Y), _int(Z), ptr(None), ptr(None),
                           *tail, st)
                    cap.stepped = True
                    return None, None, None, None, None, None
                grad_k0 = torch.empty_like(ctx.k0_meta, memory_format=torch.preserve_format)
                grad_density = torch.empty_like(ctx.density_meta)
                assert grad_k0.stride() == ctx.k0_meta.stride() and grad_density.is_contiguous()
                L.call('dvgo_brick_accumulate', *items, ptr(recs), ptr(start), ptr(dirs), _flt(cfg.stepdist),
                       cfg.xyz_min_h, cfg.xyz_max_h, ptr(g_feat), _int(C), _int(X), _int(Y), _int(Z), ptr(grad_k0),
                       ptr(grad_density), ptr(None), ptr(None), ptr(None), _flt(0), _int(0),
                       ptr(None), ptr(None), ptr(None), _flt(0), _int(0), _flt(0), _flt(0), _flt(0), ptr(None), st)
                return grad_density, grad_k0, None, None, None, None

            # worth its two extra full-grid passes (zero 64 B, split 116 B per voxel) from ~1 kept sample per 6 voxels
            combined = (COMBINED_GRID_GRAD and want_k0 and want_d and C == 12 and M3 * COMBINED_MIN_RATIO >= X * Y * Z and tuple(ctx.density_meta.shape[2:]) == (X, Y, Z)
                        and (sC, sZ, sY, sX) == (1, C, Z * C, Y * Z * C))
            if combined:
                # both grids share the voxel lattice: one scatter into 64-byte rows (12 feature channels + the
                # density gradient), then a streaming split into the two dense gradients
                G = torch.zeros((X * Y * Z, 16), dtype=torch.float32, device=dev)
                kept = torch.empty(M3, dtype=torch.float32, device=dev)
                density_bwd(G[:, 12:], 16, kept)
                L.call('dvgo_march_feat_bwd', ptr(g_feat.contiguous()), ptr(kept), ptr(ray_id), ptr(step_id), _i64(M3),
                       ptr(start), ptr(dirs), _flt(cfg.stepdist), cfg.xyz_min_h, cfg.xyz_max_h, _int(C), _int(X),
                       _int(Y), _int(Z), _i64(1), _i64(Y * Z * 16), _i64(Z * 16), _i64(16), ptr(G), st)
                cap = grid_rows_capture._active
                if cap is not None and cap.G is None and cap.density is ctx.density_meta and cap.k0 is ctx.k0_meta:
                    cap.G = G                  # the optimizer consumes the rows; no dense gradients are produced
                    return None, None, None, None, None, None
                grad_k0 = torch.empty_like(ctx.k0_meta, memory_format=torch.preserve_format)
                grad_density = torch.empty_like(ctx.density_meta)
                assert grad_k0.stride() == ctx.k0_meta.stride() and grad_density.is_contiguous()
                L.call('dvgo_grid_grad_split', ptr(G), _i64(X * Y * Z), _int(16), _int(C), ptr(grad_k0),
                       ptr(grad_density), st)
            else:
                if want_k0:
                    grad_k0 = torch.zeros_like(ctx.k0_meta, memory_format=torch.preserve_format)
                    assert grad_k0.stride() == ctx.k0_meta.stride()
                    L.call('dvgo_march_feat_bwd', ptr(g_feat.contiguous()), ptr(None), ptr(ray_id), ptr(step_id),
                           _i64(M3), ptr(start), ptr(dirs), _flt(cfg.stepdist), cfg.xyz_min_h, cfg.xyz_max_h, _int(C),
                           _int(X), _int(Y), _int(Z), _i64(sC), _i64(sX), _i64(sY), _i64(sZ), ptr(grad_k0), st)
                if want_d:
                    grad_density = torch.zeros_like(ctx.density_meta)
                    density_bwd(grad_density, 1, None)
        return grad_density, grad_k0, None, None, None, None


@torch.no_grad()
def fused_hit(rays_o, rays_d, cfg):
    """bool [N]: rays with at least one in-box sample in occupied space (lib/dvgo.py:412-423), no sample
    list materialised."""
    rays_o, rays_d = rays_o.contiguous(), rays_d.contiguous()
    for x, n in ((rays_o, 'rays_o'), (rays_d, 'rays_d')):
        check_input(x, n); check_f32(x, n)
    N, dev = rays_o.shape[0], rays_o.device
    t_min = torch.empty(N, dtype=torch.float32, device=dev)
    t_max = torch.empty_like(t_min)
    n_steps = torch.empty(N, dtype=torch.int64, device=dev)
    start = torch.empty((N, 3), dtype=torch.float32, device=dev)
    dirs = torch.empty((N, 3), dtype=torch.float32, device=dev)
    hit = torch.empty(N, dtype=torch.bool, device=dev)
    mask = cfg.mask
    with L.device_of(rays_o):
        st = stream_of(rays_o)
        L.call('dvgo_sample_pts_prepare', ptr(rays_o), ptr(rays_d), ptr(cfg.xyz_min_t), ptr(cfg.xyz_max_t), _flt(cfg.near),
               _flt(cfg.far), _flt(cfg.stepdist), _i64(N), ptr(t_min), ptr(t_max), ptr(n_steps), ptr(None), ptr(start),
               ptr(dirs), st)
        L.call('dvgo_march_hit', ptr(start), ptr(dirs), ptr(n_steps), _i64(N), cfg.xyz_min_h, cfg.xyz_max_h,
               _flt(cfg.stepdist), ptr(mask), _int(mask.shape[0]), _int(mask.shape[1]), _int(mask.shape[2]), cfg.scale_h,
               cfg.shift_h, ptr(hit), st)
    return hit


def fused_march(density, k0, rays_o, rays_d, cfg, capacity=False):
    """-> weights [M3], raw_alpha [M3], alphainv_last [N], k0 features [M3,C], ray_id, step_id [M3],
    off3 [N+1] (exclusive offsets of each ray's samples in the M3 arrays).
    `capacity=True`: no host synchronisation; the M3-sized outputs are allocated at their upper bound and only their
    first off3[N] rows are defined (pass `off3[N:]` as `m_dev` to the consumers)."""
    return _FusedMarch.apply(density, k0, rays_o.contiguous(), rays_d.contiguous(), cfg, capacity)


class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, rgb, alphainv_last, ray_id, off3, bg, m_dev=None):
        N = alphainv_last.shape[0]
        rgb = rgb.contiguous()
        weights = weights.contiguous()
        out = torch.empty((N, 3), dtype=torch.float32, device=weights.device)
        with L.device_of(weights):
            L.call('dvgo_march_composite', ptr(weights), ptr(rgb), ptr(None), ptr(off3), _i64(N),
                   ptr(alphainv_last.contiguous()), _flt(float(bg)), ptr(out), ptr(None), stream_of(weights))
        ctx.save_for_backward(weights, rgb, ray_id)
        ctx.bg = float(bg)
        ctx.N = N
        ctx.m_dev = m_dev
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        weights, rgb, ray_id = ctx.saved_tensors
        g = g.contiguous()
        M3 = weights.shape[0]
        gw = torch.empty_like(weights) if ctx.needs_input_grad[0] else None
        grgb = torch.empty_like(rgb) if ctx.needs_input_grad[1] else None
        glast = torch.empty(ctx.N, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[2] else None
        with L.device_of(weights):
            L.call('dvgo_march_composite_bwd', ptr(g), ptr(weights), ptr(rgb), ptr(ray_id), _i64(M3), ptr(ctx.m_dev), _i64(ctx.N),
                   _flt(ctx.bg), ptr(gw), ptr(grgb), ptr(glast), stream_of(weights))
        return gw, grgb, glast, None, None, None, None


def composite(weights, rgb, alphainv_last, ray_id, off3, bg, m_dev=None):
    """rgb_marched = segment_sum(weights * rgb) + alphainv_last * bg   (lib/dvgo.py:554-559)"""
    return _Composite.apply(weights, rgb, alphainv_last, ray_id, off3, bg, m_dev)


@torch.no_grad()
def composite_depth(weights, step_id, off3, n_rays):
    """depth = segment_sum(weights * step_id)   (lib/dvgo.py:569-576, no_grad in the reference)"""
    dev = weights.device
    dummy_rgb = torch.zeros((weights.shape[0], 3), dtype=torch.float32, device=dev)
    zeros = torch.zeros(n_rays, dtype=torch.float32, device=dev)
    out = torch.empty((n_rays, 3), dtype=torch.float32, device=dev)
    depth = torch.empty(n_rays, dtype=torch.float32, device=dev)
    with L.device_of(weights):
        L.call('dvgo_march_composite', ptr(weights.contiguous()), ptr(dummy_rgb), ptr(step_id), ptr(off3),
               _i64(n_rays), ptr(zeros), _flt(0.0), ptr(out), ptr(depth), stream_of(weights))
    return depth
