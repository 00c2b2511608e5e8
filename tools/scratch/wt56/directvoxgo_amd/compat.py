"""Bind-by-name shim: lets an unmodified copy of the reference's ``lib/dvgo.py`` / ``lib/dmpigo.py`` /
``lib/masked_adam.py`` pick up the HIP ops where they JIT-build their CUDA extensions.

The reference obtains its native ops at import time with
``torch.utils.cpp_extension.load(name='render_utils_cuda', sources=[...cuda...])`` (lib/dvgo.py:14-26,
lib/masked_adam.py:7-10) and imports ``torch_scatter.segment_coo`` (lib/dvgo.py:10).  ``install()``
replaces ``load`` with a dispatcher keyed on the extension *name* and registers a ``torch_scatter``
stand-in when that package is absent, so no CUDA source is ever compiled (or hipified).
Call it before importing the reference modules.
"""
import sys
import types

_NAMES = ('render_utils_cuda', 'total_variation_cuda', 'adam_upd_cuda')


def _modules():
    from . import masked_adam, ops, render_utils
    tv = types.SimpleNamespace(total_variation_add_grad=ops.total_variation_add_grad)

    def _adam(mode):
        def fn(param, grad, exp_avg, exp_avg_sq, *rest):
            if mode == 2:
                perlr, step, beta1, beta2, lr, eps = rest
            else:
                (step, beta1, beta2, lr, eps), perlr = rest, None
            masked_adam.adam_upd(param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps, mode=mode, perlr=perlr)
        return fn
    adam = types.SimpleNamespace(adam_upd=_adam(0), masked_adam_upd=_adam(1), adam_upd_with_perlr=_adam(2))
    return {'render_utils_cuda': render_utils, 'total_variation_cuda': tv, 'adam_upd_cuda': adam}


def install(patch_torch_scatter=True):
    """Returns the original ``load`` so callers can restore it."""
    import torch.utils.cpp_extension as cpp_ext
    original = cpp_ext.load
    mods = _modules()

    def load(name, *args, **kwargs):
        if name in mods:
            return mods[name]
        return original(name, *args, **kwargs)

    cpp_ext.load = load
    if patch_torch_scatter and 'torch_scatter' not in sys.modules:
        try:
            import torch_scatter  # noqa: F401
        except Exception:
            from . import ops
            ts = types.ModuleType('torch_scatter')
            ts.segment_coo = ops.segment_coo
            sys.modules['torch_scatter'] = ts
    return original
