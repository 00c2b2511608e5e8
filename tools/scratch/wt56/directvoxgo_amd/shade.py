"""Fused colour head (rgbnet) -- host side of csrc/shade.hip (row N3 of SURVEY.md section 8f).

`shade(...)` computes what /root/reference/lib/dvgo.py:516-541 computes with
`viewdirs_emb[ray_id]`, `torch.cat`, three `nn.Linear`s, two ReLUs and a sigmoid, in one fp32-MFMA
kernel.  It applies to the reference's default head, Sequential(Linear, ReLU, Sequential(Linear, ReLU),
Linear) with width 128 or 64; other shapes return None and the caller keeps the torch modules.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib as L
from ._lib import _i64, _int, ptr, stream_of

N_PARTS = 512     # max workgroups (= partial sums) of the weight-gradient kernel: two per CU (55 KB of LDS each)


class defer_wgrad:
    """Context manager: inside it, the colour head's backward only produces the data gradient and hands its
    weight-gradient kernel to this object; `flush()` accumulates the results into the parameters' `.grad`.

    The weight-gradient kernel (fp32 MFMA, ~1 ms on the roofline case) and what autograd runs next -- the
    grid-gradient scatters, bound by atomic requests, then in data-parallel runs the grid all-reduce -- use
    different parts of the machine, so with `side_stream=True` (default) the kernel is launched right away
    on a second HIP stream and runs CONCURRENTLY with them; `flush()` joins the streams.  With
    `side_stream=False` the kernel is merely postponed to `flush()` on the calling stream."""
    _active = None
    _streams = {}

    def __init__(self, side_stream=True):
        self.side_stream = side_stream

    def __enter__(self):
        self.pending = []
        defer_wgrad._active = self
        return self

    def __exit__(self, *exc):
        defer_wgrad._active = None
        return False

    def submit(self, params, fn, device):
        if not self.side_stream:
            self.pending.append((params, fn, None, None))
            return
        side = defer_wgrad._streams.get(device)
        if side is None:
            side = defer_wgrad._streams[device] = torch.cuda.Stream(device=device)
        main = torch.cuda.current_stream(device)
        side.wait_stream(main)                 # the data-gradient kernel's outputs are this kernel's inputs
        with torch.cuda.stream(side):
            grads = fn()                       # launched now; `fn` keeps its operand tensors alive until flush()
            done = torch.cuda.Event()
            done.record(side)
        self.pending.append((params, fn, grads, done))

    @torch.no_grad()
    def flush(self):
        for params, fn, grads, done in self.pending:
            if grads is None:
                grads = fn()
            else:
                torch.cuda.current_stream(grads[0].device).wait_event(done)
            for p, g in zip(params, grads):
                if p.requires_grad:
                    p.grad = g if p.grad is None else p.grad + g
        self.pending = []


@torch.no_grad()
def viewdir_embed(viewdirs, viewfreq):
    """cat([v, sin(v (x) freq), cos(v (x) freq)]) of lib/dvgo.py:524-525 in one launch -> [N, 3 + 6F]."""
    from ._lib import _flt  # noqa: F401
    N, F = viewdirs.shape[0], viewfreq.shape[0]
    emb = torch.empty((N, 3 + 6 * F), dtype=torch.float32, device=viewdirs.device)
    with L.device_of(viewdirs):
        L.call('dvgo_viewdir_embed', ptr(viewdirs.contiguous()), ptr(viewfreq), _int(F), _i64(N), ptr(emb),
               stream_of(viewdirs))
    return emb


def head_layers(rgbnet):
    """(lin1, lin2, lin3) when the module tree is the 3-layer head the kernel implements, else None."""
    if not isinstance(rgbnet, nn.Sequential) or len(rgbnet) != 4:
        return None
    a, r, mid, c = rgbnet
    if not (isinstance(a, nn.Linear) and isinstance(r, nn.ReLU) and isinstance(c, nn.Linear)):
        return None
    if not (isinstance(mid, nn.Sequential) and len(mid) == 2 and isinstance(mid[0], nn.Linear) and isinstance(mid[1], nn.ReLU)):
        return None
    width = a.out_features
    if width not in (128, 64) or mid[0].in_features != width or mid[0].out_features != width or c.in_features != width \
            or c.out_features != 3:
        return None          # built: 128 (configs/default.py) and 64 (configs/llff, lib/dmpigo.py)
    if a.in_features > 40:
        return None
    return a, mid[0], c


def _scratch(width, device):
    """Device scratch for the bf16-split kernels' weight image (csrc/shade_x3.hip): a fresh block per call from torch's
    caching allocator, so that calls on different streams never share one."""
    lib = L.lib()
    lib.dvgo_shade_scratch_bytes.restype = ctypes.c_int64
    n = int(lib.dvgo_shade_scratch_bytes(int(width)))
    return torch.empty(n, dtype=torch.uint8, device=device) if n > 0 else None


class _Shade(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, emb, ray_id, W1, b1, W2, b2, W3, b3, diffuse, train, m_dev):
        M, C = feat.shape
        E = emb.shape[1]
        width, d_in = W1.shape
        feat, emb = feat.contiguous(), emb.contiguous()
        # `train`: grad mode was on at the call (inside forward() it is always off, and needs_input_grad is set for
        # the parameters even under torch.no_grad()): rendering must not pay the 1 KB / sample of activation stores
        train = bool(train) and any(ctx.needs_input_grad)
        rgb = torch.empty((M, 3), dtype=torch.float32, device=feat.device)
        H1 = torch.empty((M, width), dtype=torch.float32, device=feat.device) if train else None
        H2 = torch.empty((M, width), dtype=torch.float32, device=feat.device) if train else None
        masks = torch.empty((M, 4), dtype=torch.int64, device=feat.device) if train else None
        scratch = _scratch(width, feat.device)
        with L.device_of(feat):
            L.call('dvgo_shade_fwd', ptr(feat), _int(C), ptr(emb), _int(E), ptr(ray_id), _i64(M), ptr(m_dev), ptr(W1.contiguous()),
                   ptr(b1.contiguous()), ptr(W2.contiguous()), ptr(b2.contiguous()), ptr(W3.contiguous()),
                   ptr(b3.contiguous()), _int(width), _int(d_in), _int(1 if diffuse else 0), ptr(rgb), ptr(H1), ptr(H2),
                   ptr(masks), ptr(scratch), stream_of(feat))
        if train:
            ctx.save_for_backward(feat, emb, ray_id, W1, W2, W3, rgb, H1, H2, masks)
            ctx.diffuse = diffuse
            ctx.m_dev = m_dev            # None, or the device-side sample count (arrays are then capacity-sized)
            ctx.params = (W1, b1, W2, b2, W3, b3)
        return rgb

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_rgb):
        feat, emb, ray_id, W1, W2, W3, rgb, H1, H2, masks = ctx.saved_tensors
        diffuse, m_dev = ctx.diffuse, ctx.m_dev
        M, C = feat.shape
        width, d_in = W1.shape
        g_feat = torch.empty_like(feat)
        G1 = torch.empty_like(H1)
        gz = torch.empty_like(rgb)
        scratch = _scratch(width, feat.device)
        with L.device_of(feat):
            L.call('dvgo_shade_bwd', ptr(g_rgb.contiguous()), ptr(rgb), ptr(masks), _i64(M), ptr(m_dev), ptr(W1.contiguous()),
                   ptr(W2.contiguous()), ptr(W3.contiguous()), _int(width), _int(d_in), _int(C), _int(1 if diffuse else 0),
                   ptr(g_feat), ptr(G1), ptr(gz), ptr(scratch), stream_of(feat))

        def wgrad():
            n_parts = max(1, min(N_PARTS, (M + 255) // 256))      # >= 8 row tiles per workgroup on small batches
            psize = width * width + width * 64 + 32 * width + 3 * width
            part = torch.empty((n_parts, psize), dtype=torch.float32, device=feat.device)
            tot = torch.empty(psize, dtype=torch.float32, device=feat.device)
            with L.device_of(feat):
                L.call('dvgo_shade_wgrad', ptr(G1), ptr(gz), ptr(masks), ptr(W3.contiguous()), ptr(H1), ptr(H2), ptr(feat), _int(C), ptr(emb),
                       _int(emb.shape[1]), ptr(ray_id), _i64(M), ptr(m_dev), _int(width), _int(1 if diffuse else 0), _int(n_parts),
                       ptr(part), ptr(tot), stream_of(feat))
            o = 0
            gW2 = tot[o:o + width * width].view(width, width); o += width * width
            gW1 = tot[o:o + width * 64].view(width, 64)[:, :d_in]; o += width * 64
            gW3 = tot[o:o + 32 * width].view(32, width)[:3]; o += 32 * width
            gb1, gb2 = tot[o:o + width], tot[o + width:o + 2 * width]
            gb3 = tot[o + 2 * width:o + 2 * width + 3] + tot[o + 2 * width + 8:o + 2 * width + 11]
            return gW1.contiguous(), gb1, gW2, gb2, gW3.contiguous(), gb3

        gf = g_feat if ctx.needs_input_grad[0] else None
        if defer_wgrad._active is not None:
            defer_wgrad._active.submit(ctx.params, wgrad, feat.device)
            return (gf, None, None, None, None, None, None, None, None, None, None, None)
        gW1, gb1, gW2, gb2, gW3, gb3 = wgrad()
        return (gf, None, None, gW1, gb1, gW2, gb2, gW3, gb3, None, None, None)


def shade(rgbnet, feat, emb, ray_id, diffuse, m_dev=None):
    """rgb [M,3] = sigmoid(rgbnet(cat([feat[:,3:] if diffuse else feat, emb[ray_id]])) + (feat[:,:3] if diffuse)),
    or None when the head is not the shape the kernel was built for.
    `m_dev`: device-side sample count when the arrays are capacity-sized (fused.py `capacity` mode): rows past it are
    neither read nor written."""
    layers = head_layers(rgbnet)
    if layers is None or not feat.is_cuda:
        return None
    l1, l2, l3 = layers
    c0 = 3 if diffuse else 0
    if l1.in_features != feat.shape[1] - c0 + emb.shape[1]:
        return None
    return _Shade.apply(feat, emb, ray_id, l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias, diffuse,
                        torch.is_grad_enabled(), m_dev)
