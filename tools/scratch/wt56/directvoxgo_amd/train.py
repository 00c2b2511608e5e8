"""Training step of the hot path (harness row H3 of SURVEY.md section 8a) and its ray-parallel
data-parallel form (section 8e).

Counterpart of /root/reference/run.py:348-407: loss terms and weights (:377-386), gradient step
with ``zero_grad(set_to_none=True)`` before backward (:376), optional TV add-grad (:389-395),
MaskedAdam step (:397) and the per-step lr decay (:401-406); optimizer construction follows
lib/utils.py:20-48.

Data parallelism (the reference has none): one process per GPU, rays sharded,
grids / MLP / optimizer state replicated.  Every loss term is normalised by the GLOBAL ray
count so that the sum of the per-rank gradients equals the single-process gradient; the grid
gradients are summed with one all-reduce each (RCCL over xGMI; `backend='nccl'` on ROCm) and
the small MLP gradients travel in one flat bucket.  TV and the masked Adam run after the
reduction because both branch on ``grad != 0`` (total_variation_kernel.cu:21,
adam_upd_kernel.cu:35) and must see the reduced gradient.
"""
import contextlib

import torch
import torch.distributed as dist
import torch.nn as nn

from .masked_adam import MaskedAdam
from .fused import grid_rows_capture, split_grid_rows
from .shade import defer_wgrad

def flat_view(t):
    """1-D view of a dense tensor's memory (no copy): collectives want plain contiguous buffers, and the
    feature grid / its gradient are stored channels-last."""
    if t.is_contiguous():
        return t.view(-1)
    if t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d):
        v = t.permute(0, 2, 3, 4, 1).reshape(-1)
        assert v.data_ptr() == t.data_ptr()
        return v
    return None


COARSE_TRAIN = dict(
    N_iters=5000, N_rand=8192, lrate_density=1e-1, lrate_k0=1e-1, lrate_rgbnet=1e-3, lrate_decay=20,
    pervoxel_lr=True, weight_main=1.0, weight_entropy_last=0.01, weight_rgbper=0.1,
    tv_every=1, tv_after=0, tv_before=0, tv_dense_before=0, weight_tv_density=0.0, weight_tv_k0=0.0,
    pg_scale=[], skip_zero_grad_fields=[])            # configs/default.py:36-57
FINE_TRAIN = dict(COARSE_TRAIN, N_iters=20000, pervoxel_lr=False, weight_entropy_last=0.001, weight_rgbper=0.01,
                  pg_scale=[1000, 2000, 3000, 4000], skip_zero_grad_fields=['density', 'k0'])   # :59-68


def create_optimizer_or_freeze_model(model, cfg_train, global_step):
    """lib/utils.py:20-48: one param group per `lrate_<name>` whose attribute exists on the model."""
    decay_steps = cfg_train['lrate_decay'] * 1000
    decay_factor = 0.1 ** (global_step / decay_steps)
    groups = []
    for key in cfg_train:
        if not key.startswith('lrate_'):
            continue
        name = key[len('lrate_'):]
        if not hasattr(model, name):
            continue
        param = getattr(model, name)
        if param is None:
            continue
        lr = cfg_train[key] * decay_factor
        if lr > 0:
            if isinstance(param, nn.Module):
                param = param.parameters()
            groups.append({'params': param, 'lr': lr, 'skip_zero_grad': name in cfg_train['skip_zero_grad_fields']})
        elif not isinstance(param, dict):
            param.requires_grad = False
    return MaskedAdam(groups)


def render_loss(render_result, target, n_rays_global, cfg_train):
    """run.py:377-386 with every mean written as sum / global count (identical for one rank)."""
    d = render_result['rgb_marched'] - target
    loss = cfg_train['weight_main'] * d.pow(2).sum() / (3 * n_rays_global)
    if cfg_train['weight_entropy_last'] > 0:
        pout = render_result['alphainv_last'].clamp(1e-6, 1 - 1e-6)
        ent = -(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout)).sum() / n_rays_global
        loss = loss + cfg_train['weight_entropy_last'] * ent
    if cfg_train['weight_rgbper'] > 0:
        rgbper = (render_result['raw_rgb'] - target[render_result['ray_id']]).pow(2).sum(-1)
        loss = loss + cfg_train['weight_rgbper'] * ((rgbper * render_result['weights'].detach()).sum() / n_rays_global)
    return loss


class _FusedLoss(torch.autograd.Function):
    """render_loss in one pass (csrc/loss.hip): the value and d/d{rgb_marched, alphainv_last, raw_rgb}."""
    unit_grad = False        # set by TrainStep around its own loss.backward() (saves three scaling launches)

    @staticmethod
    def forward(ctx, rgb_marched, alphainv_last, raw_rgb, weights, ray_id, target, n_global, w_main, w_ent, w_per, m_dev=None):
        from . import _lib as L
        from ._lib import _flt, _i64, ptr, stream_of
        N, M = rgb_marched.shape[0], raw_rgb.shape[0]
        dev = rgb_marched.device
        rgb_marched, alphainv_last, raw_rgb = rgb_marched.contiguous(), alphainv_last.contiguous(), raw_rgb.contiguous()
        g_marched = torch.empty_like(rgb_marched)
        g_last = torch.empty_like(alphainv_last)
        g_raw = torch.empty_like(raw_rgb) if w_per > 0 else None
        loss = torch.empty((), dtype=torch.float32, device=dev)
        with L.device_of(rgb_marched):
            L.call('dvgo_loss_fwd_bwd', ptr(rgb_marched), ptr(alphainv_last), ptr(target.contiguous()), _i64(N), ptr(raw_rgb),
                   ptr(weights.contiguous()), ptr(ray_id), _i64(M), ptr(m_dev), _i64(int(n_global)), _flt(w_main), _flt(w_ent),
                   _flt(w_per), ptr(g_marched), ptr(g_last), ptr(g_raw), ptr(loss), stream_of(rgb_marched))
        ctx.save_for_backward(g_marched, g_last, g_raw if g_raw is not None else g_last)
        ctx.has_raw = g_raw is not None
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, go):
        g_marched, g_last, g_raw = ctx.saved_tensors
        if _FusedLoss.unit_grad:         # TrainStep calls loss.backward() itself: d loss / d loss = 1, nothing to scale
            return (g_marched, g_last, g_raw if ctx.has_raw else None, None, None, None, None, None, None, None, None)
        return (g_marched * go, g_last * go, (g_raw * go) if ctx.has_raw else None, None, None, None, None, None, None,
                None, None)


def fused_render_loss(render_result, target, n_rays_global, cfg_train):
    """Same value and gradients as `render_loss`, one kernel pair instead of ~40 framework launches."""
    return _FusedLoss.apply(render_result['rgb_marched'], render_result['alphainv_last'], render_result['raw_rgb'],
                            render_result['weights'].detach(), render_result['ray_id'], target, n_rays_global,
                            float(cfg_train['weight_main']), float(cfg_train['weight_entropy_last']),
                            float(cfg_train['weight_rgbper']), render_result.get('n_samples'))


class TrainStep:
    """One optimisation step on one batch of rays; ``world_size > 1`` shards the batch by rank."""

    def __init__(self, model, cfg_train, render_kwargs, optimizer=None, process_group=None, fused_loss=True,
                 overlap_wgrad=True, touched_reduce=True, rows_adam=True, track_mse=False, shard_grids=True, sync_free=False):
        self.model = model
        # keep the count of surviving samples on the device (model.forward(_capacity=True)): no host synchronisation in
        # the step.  That is what makes the step capturable (`capture()` switches it on); run eagerly it buys nothing --
        # the sparse step is bound by host work, not by the one read -- and costs capacity-sized temporaries, so it is
        # off by default
        self.sync_free = sync_free
        self._m3_seen = None                 # last sample count read back (asynchronously, a step or two late)
        self._m3_pin, self._m3_event = None, None
        # data parallel, dense scenes: reduce-scatter the grid gradients, update only the owned slab, all-gather the
        # parameters (see _sharded_*); False = plain all-reduce + full update on every rank
        self.shard_grids = shard_grids
        # weight_main * mse of the step, the quantity run.py:378 turns into the logged PSNR (before the entropy and
        # per-point terms are added); kept on the device, no sync
        self.track_mse = track_mse
        self.last_mse = None
        # one GPU, no TV this step: Adam reads the combined gradient rows of the fused backward directly
        # (fused.grid_rows_capture / MaskedAdam.step_grid_rows); density.grad / k0.grad then stay None
        self.rows_adam = rows_adam
        # data parallel, sparse scenes: all-reduce only the voxels some rank touched (see _reduce_touched)
        self.touched_reduce = touched_reduce
        self._touched_frac = None            # fraction of voxels in the last union; None: not probed yet
        self._steps_since_probe = 0
        self.overlap_wgrad = overlap_wgrad    # colour-head weight gradients on a second stream (shade.defer_wgrad)
        self.fused_loss = fused_loss
        self.cfg = cfg_train
        self.render_kwargs = render_kwargs
        self.optimizer = optimizer or create_optimizer_or_freeze_model(model, cfg_train, global_step=0)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.decay_factor = 0.1 ** (1 / (cfg_train['lrate_decay'] * 1000))
        self._small = [p for n, p in model.named_parameters() if n not in ('density', 'k0') and p.requires_grad]

    def reduce_grids_async(self):
        """Sparse scenes: start the compacted touched-voxel reduction (see _reduce_touched); returns its handle, or []
        when the dense path (sharded reduce-scatter / all-reduce) has to take the step."""
        works = []
        if self.world == 1:
            return works
        if self.touched_reduce:
            self._steps_since_probe += 1
            probe = self._touched_frac is None or self._steps_since_probe >= self.PROBE_EVERY
            if probe or self._touched_frac <= self.TOUCHED_MAX:
                rd = self._rows()
                pending = self._reduce_touched(*rd) if rd is not None else None
                if pending is not None:
                    return [pending]
        return works

    def _all_reduce_grids(self):
        """Plain sum of the full grid gradients on every rank (the fallback when the grids cannot be sharded)."""
        works = []
        for p in (self.model.density, self.model.k0):
            if p.grad is not None:
                flat = flat_view(p.grad)
                if flat is not None:
                    works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
                else:                                 # exotic strides: staged through a contiguous copy
                    tmp = p.grad.contiguous()
                    dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.pg)
                    p.grad.copy_(tmp)
        return works

    # A batch of rays touches the voxels along those rays only: on a trained scene a few per cent of the grid, while
    # the dense all-reduce always moves all of it (213 MB at 160^3 -- more than a whole step of compute on such
    # scenes).  The touched set differs per rank, so: OR-reduce a byte mask (4 MB), compact the union's rows
    # [n, C + 1] (features + density), all-reduce that, write back.  Untouched voxels stay exactly zero on every
    # rank, which is what the masked Adam and the sparse TV branch on.  Used while the union stays below
    # TOUCHED_MAX of the grid (decided from the previous union, identical on all ranks; re-probed every
    # PROBE_EVERY steps while the dense path is in use).
    TOUCHED_MAX = 0.35
    PROBE_EVERY = 64
    OVERLAP_MIN_SAMPLES = 600000

    def _rows(self):
        """(k0.grad as [n_vox, C] rows, density.grad as [n_vox]) when both share the lattice and are row-addressable."""
        d, k = self.model.density.grad, self.model.k0.grad
        if d is None or k is None or d.dim() != 5 or k.dim() != 5 or d.shape[2:] != k.shape[2:] or not d.is_contiguous():
            return None
        flat = flat_view(k) if k.is_contiguous(memory_format=torch.channels_last_3d) else None
        if flat is None:
            return None
        return flat.view(-1, k.shape[1]), d.view(-1)

    def _reduce_touched(self, rows, dflat):
        mask = (rows != 0).any(1) | (dflat != 0)
        m8 = mask.to(torch.uint8)
        dist.all_reduce(m8, op=dist.ReduceOp.MAX, group=self.pg)
        idx = m8.nonzero().flatten()                       # the union, identical on every rank (one host read)
        self._touched_frac = idx.numel() / max(m8.numel(), 1)
        self._steps_since_probe = 0
        if self._touched_frac > self.TOUCHED_MAX:
            return None                                     # dense scene: the caller falls back to the plain all-reduce
        C = rows.shape[1]
        compact = torch.empty((idx.numel(), C + 1), dtype=rows.dtype, device=rows.device)
        if idx.numel():
            compact[:, :C] = rows[idx]
            compact[:, C] = dflat[idx]
        work = dist.all_reduce(compact, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

        class _Pending:
            def wait(_self):
                work.wait()
                if idx.numel():
                    rows[idx] = compact[:, :C]
                    dflat[idx] = compact[:, C]
        return _Pending()

    # ------------------------------------------------------------------------------------------------------------
    # Dense scenes (every voxel has a gradient: the roofline case): a plain all-reduce moves 2 (P-1)/P x 213 MB per rank
    # AND leaves every rank sweeping all 53 M elements through Adam.  Instead (ZeRO-1 style):
    #   reduce_scatter   rank r receives the SUM of the gradient of the X-planes [r X/P, (r+1) X/P) -- in place, the slab
    #                    is a contiguous range of the gradient's memory (channels-last / C == 1: X is the outermost axis)
    #   TV + Adam        on that slab only (1/P of the optimizer's traffic; the TV stencil reads the replicated params)
    #   all_gather       the updated parameter slabs, in place in the parameters
    # Same bytes on the wire as the all-reduce ((P-1)/P x 213 MB out and in per rank and phase, spread over all xGMI
    # links by RCCL), 1/P of the optimizer work, and the parameters -- not the gradients -- are what ends up replicated.
    # ------------------------------------------------------------------------------------------------------------
    def _grid_shards(self):
        """[(param, flat param, flat grad, lo, hi, (x_lo, x_hi))] for the grids when the sharded update applies."""
        if not (self.shard_grids and self.world > 1 and hasattr(self.optimizer, 'step_shard')):
            return None
        rank = dist.get_rank(self.pg)
        out = []
        for p in (getattr(self.model, 'density', None), getattr(self.model, 'k0', None)):
            if not isinstance(p, nn.Parameter) or p.grad is None or p.dim() != 5:
                return None
            x_outermost = p.shape[1] == 1 and p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last_3d)
            fp, fg = flat_view(p.data), flat_view(p.grad)
            X = p.shape[2]
            if not x_outermost or fp is None or fg is None or p.grad.stride() != p.stride() or X % self.world != 0:
                return None
            n = fp.numel() // self.world
            out.append((p, fp, fg, rank * n, (rank + 1) * n, (rank * (X // self.world), (rank + 1) * (X // self.world))))
        return out

    def _sharded_reduce_start(self, shards):
        return [dist.reduce_scatter_tensor(fg[lo:hi], fg, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                for _, _, fg, lo, hi, _ in shards]

    def _sharded_update(self, shards):
        """Adam on the owned slabs, then the parameters travel.  Returns the all-gather handles."""
        works = []
        for p, fp, fg, lo, hi, _ in shards:
            self.optimizer.step_shard(p, fp, fg, lo, hi)
            p.grad = None                      # consumed: optimizer.step() below skips the grids
            works.append(dist.all_gather_into_tensor(fp, fp[lo:hi], group=self.pg, async_op=True))
        return works

    @torch.no_grad()
    def gather_optimizer_state(self):
        """Data-parallel runs with the sharded update: every rank has only ever updated the moments of the X-slab it
        owns.  Before `checkpoint.save_checkpoint` (or any other reader of `optimizer.state_dict()`), all-gather the slabs
        in place so that every rank holds the complete `exp_avg` / `exp_avg_sq` of both grids -- the state a single
        process would have written (run.py:420-437).  No-op on one rank or when the grids are not sharded."""
        if not (self.shard_grids and self.world > 1 and hasattr(self.optimizer, 'step_shard')):
            return False
        rank = dist.get_rank(self.pg)
        done = False
        for p in (getattr(self.model, 'density', None), getattr(self.model, 'k0', None)):
            st = self.optimizer.state.get(p) if isinstance(p, nn.Parameter) else None
            if not st or p.dim() != 5 or p.shape[2] % self.world != 0:
                continue
            for key in ('exp_avg', 'exp_avg_sq'):
                flat = flat_view(st[key])
                if flat is None or st[key].stride() != p.stride():
                    raise RuntimeError(f'gather_optimizer_state: {key} is not laid out like its parameter')
                n = flat.numel() // self.world
                dist.all_gather_into_tensor(flat, flat[rank * n:(rank + 1) * n].clone(), group=self.pg)
            done = True
        return done

    def _sample_count(self, res):
        """Number of surviving samples of the step, without waiting for it: exact when the forward read it back anyway,
        else the last value that has arrived from the device (copied asynchronously into pinned memory every step)."""
        n_dev = res.get('n_samples')
        if n_dev is None:
            return res['weights'].shape[0]
        if self._m3_event is not None and self._m3_event.query():
            self._m3_seen = int(self._m3_pin[0])
        if self._m3_pin is None:
            self._m3_pin = torch.empty(1, dtype=torch.int64).pin_memory()
            self._m3_event = torch.cuda.Event()
        if self._m3_event.query():                 # the previous copy has landed: start the next one
            self._m3_pin.copy_(n_dev, non_blocking=True)
            self._m3_event.record()
        return self._m3_seen if self._m3_seen is not None else res['weights'].shape[0]

    def reduce_small(self):
        """One flat bucket for the handful of MLP gradients."""
        if self.world == 1:
            return
        small = [p for p in self._small if p.grad is not None]
        if small:
            flat = torch.cat([p.grad.reshape(-1) for p in small])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg)
            off = 0
            for p in small:
                n = p.grad.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n

    # ------------------------------------------------------------------------------------------------------------
    # HIP-graph replay of the step.  A sparse-scene step is ~45 short kernels (0.55 ms of GPU work at 8192 rays on a
    # lego-like scene) behind ~0.8 ms of host work (Python, allocator, launches): with the sample count kept on the device
    # (`sync_free`) nothing in the step depends on a host read any more, so the whole of it -- forward, loss, backward,
    # grid update inside the brick kernel, MLP Adam -- is captured once and replayed.  What changes from step to step
    # travels through device memory: the batch (copied into the captured input tensors) and the bias-corrected Adam step
    # sizes (`MaskedAdam.hyper_begin`).  Captured: one GPU, fused model + fused colour head, masked Adam on both grids,
    # no total variation, fixed batch size and grid resolution; call `capture()` again after `scale_volume_grid` or an
    # occupancy-mask refresh (both replace tensors the graph holds).
    # ------------------------------------------------------------------------------------------------------------
    def can_capture(self):
        cfg, model = self.cfg, self.model
        density, k0 = getattr(model, 'density', None), getattr(model, 'k0', None)
        tv = (cfg['weight_tv_density'] > 0 or cfg['weight_tv_k0'] > 0) and cfg['tv_before'] > cfg['tv_after']
        return bool(self.world == 1 and self.fused_loss and self.rows_adam and not tv
                    and isinstance(self.optimizer, MaskedAdam) and isinstance(density, nn.Parameter) and density.is_cuda
                    and hasattr(model, 'can_keep_count_on_device') and model.can_keep_count_on_device()
                    and self.optimizer.can_fuse_grid_step(density, k0) and self.optimizer.per_lr is None)

    def capture(self, rays_o, rays_d, viewdirs, target, global_step=0, warmup=3):
        """Run `warmup` eager steps on the given batch, then capture one step; later calls with a batch of the same size
        replay it.  Returns False (and stays eager) when the step cannot be captured."""
        self._graph = None
        if not self.can_capture():
            return False
        model, opt = self.model, self.optimizer
        self.sync_free = True                                      # from here on the sample count stays on the device
        self._static = [t.detach().clone().contiguous() for t in (rays_o, rays_d, viewdirs, target)]
        opt.hyper_begin(model.density, model.k0)                  # creates the device-side step sizes; eager steps use them too
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                              # (torch: warm up on a side stream before capturing)
            for i in range(warmup):
                self._eager(*self._static, global_step + i)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # the capture pass runs the Python of one step without executing its kernels: put the host-side state back after
        steps = {p: st['step'] for p, st in opt.state.items()}
        lrs = [g['lr'] for g in opt.param_groups]
        opt.hyper_begin(model.density, model.k0)
        graph = torch.cuda.CUDAGraph()
        self._capturing = True
        try:
            with torch.cuda.graph(graph):
                self._static_loss = self._eager(*self._static, global_step + warmup)
        finally:
            self._capturing = False
        for p, n in steps.items():
            opt.state[p]['step'] = n
        for g, lr in zip(opt.param_groups, lrs):
            g['lr'] = lr
        self._graph = graph
        return True

    def _replay(self, rays_o, rays_d, viewdirs, target):
        for dst, src in zip(self._static, (rays_o, rays_d, viewdirs, target)):
            dst.copy_(src, non_blocking=True)
        self.optimizer.hyper_begin(self.model.density, self.model.k0, advance=True)    # this step's Adam step sizes
        self._graph.replay()
        for group in self.optimizer.param_groups:                                                  # run.py:401-406
            group['lr'] = group['lr'] * self.decay_factor
        return self._static_loss

    def __call__(self, rays_o, rays_d, viewdirs, target, global_step):
        """rays are this rank's shard; returns the (local share of the) loss as a 0-dim tensor (after `capture()`: a
        tensor that the next call overwrites)."""
        if getattr(self, '_graph', None) is not None and rays_o.shape == self._static[0].shape:
            return self._replay(rays_o, rays_d, viewdirs, target)
        return self._eager(rays_o, rays_d, viewdirs, target, global_step)

    def _eager(self, rays_o, rays_d, viewdirs, target, global_step):
        cfg, model = self.cfg, self.model
        if isinstance(self.optimizer, MaskedAdam) and self.optimizer.hyper_dev is not None and not getattr(self, '_capturing', False):
            self.optimizer.hyper_begin(model.density, model.k0)   # device-side step sizes of this step (see capture())
        n_global = rays_o.shape[0] * self.world
        keep_on_device = (self.sync_free and self.fused_loss and rays_o.is_cuda and hasattr(model, 'can_keep_count_on_device')
                          and model.can_keep_count_on_device())
        extra = {'_capacity': True} if keep_on_device else {}
        res = model(rays_o, rays_d, viewdirs, global_step=global_step, **self.render_kwargs, **extra)
        self.optimizer.zero_grad(set_to_none=True)
        loss_fn = fused_render_loss if (self.fused_loss and res['rgb_marched'].is_cuda) else render_loss
        loss = loss_fn(res, target, n_global, cfg)
        if self.track_mse:
            self.last_mse = cfg['weight_main'] * (res['rgb_marched'].detach() - target).pow(2).sum() / (3 * n_global)
        # backward order: ... colour-head data gradient -> grid scatters.  One GPU: the colour head's weight-gradient
        # kernel runs on a second stream beside the scatters.  Data parallel: it is postponed until the grid
        # all-reduce has been STARTED -- its persistent workgroups fill every CU, and RCCL's kernels, arriving
        # second, would sit behind them; arriving first they keep their CUs and the two overlap
        tv_now = (cfg['tv_after'] < global_step < cfg['tv_before'] and global_step % cfg['tv_every'] == 0 and
                  (cfg['weight_tv_density'] > 0 or cfg['weight_tv_k0'] > 0))
        density, k0 = getattr(model, 'density', None), getattr(model, 'k0', None)
        own = (self.rows_adam and self.world == 1 and not tv_now and isinstance(self.optimizer, MaskedAdam)
               and isinstance(density, nn.Parameter) and isinstance(k0, nn.Parameter) and density.is_cuda)
        fuse_adam = own and self.optimizer.can_fuse_grid_step(density, k0)
        use_rows = fuse_adam or (own and self.optimizer.can_step_grid_rows(density, k0))
        opt = self.optimizer
        rows = (grid_rows_capture(density, k0, adam=(lambda: opt.grid_step_args(density, k0)) if fuse_adam else None)
                if use_rows else contextlib.nullcontext())
        # the second stream pays on kernel-bound steps (the weight-gradient kernel beside the grid scatter: -0.3 ms at
        # 2 M samples) and costs on launch-bound ones (stream switches and event records on the host: +0.1 ms at 0.2 M)
        if getattr(self, '_capturing', False):       # no host reads while a graph is being captured: the last count seen
            n_samples = self._m3_seen if self._m3_seen is not None else res['weights'].shape[0]
        else:
            n_samples = self._sample_count(res)
        side = self.overlap_wgrad and self.world == 1 and n_samples >= self.OVERLAP_MIN_SAMPLES
        with defer_wgrad(side_stream=side) as deferred, rows as cap:
            _FusedLoss.unit_grad = True
            try:
                loss.backward()
            finally:
                _FusedLoss.unit_grad = False
        if use_rows and cap.stepped:
            assert density.grad is None and k0.grad is None    # both grids were updated inside the backward (csrc/brick.hip)
        elif use_rows and cap.G is not None:
            if density.grad is None and k0.grad is None:
                self.optimizer.step_grid_rows(density, k0, cap.G)  # (.grad of the two grids is None: step() below skips them)
            else:                                                 # more than one march in the graph: fold the rows back
                gd, gk = split_grid_rows(cap.G, density, k0)
                density.grad = gd if density.grad is None else density.grad + gd
                k0.grad = gk if k0.grad is None else k0.grad + gk
            cap.G = None
        works = self.reduce_grids_async()
        shards = None
        if self.world > 1 and not works:            # (the compacted touched-voxel reduction took the sparse case)
            shards = self._grid_shards()
            works = self._sharded_reduce_start(shards) if shards else self._all_reduce_grids()
        deferred.flush()
        self.reduce_small()
        for wk in works:
            wk.wait()
        if cfg['tv_after'] < global_step < cfg['tv_before'] and global_step % cfg['tv_every'] == 0:   # run.py:389-395
            dense = global_step < cfg['tv_dense_before']
            xr = {'x_range': shards[0][5]} if shards else {}      # a rank that owns a slab adds the TV gradient of that slab
            if cfg['weight_tv_density'] > 0:
                model.density_total_variation_add_grad(cfg['weight_tv_density'] / n_global, dense, **xr)
            if cfg['weight_tv_k0'] > 0:
                model.k0_total_variation_add_grad(cfg['weight_tv_k0'] / n_global, dense, **xr)
        gathers = self._sharded_update(shards) if shards else []
        self.optimizer.step()
        for wk in gathers:
            wk.wait()
        for group in self.optimizer.param_groups:                                                  # run.py:401-406
            group['lr'] = group['lr'] * self.decay_factor
        return loss.detach()
