"""ctypes binding of libdvgo_hip.so (include/dvgo_hip.h).

This is the only place the shared library is loaded.  There is no CPU fallback: if the
library is missing the import of any op module raises, and every op rejects non-CUDA tensors
with the reference's own error text (lib/cuda/render_utils.cpp:40-42).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DVGO_HIP_SO: another build of the same library (kernel A/B runs, tools/)
SO_PATH = os.environ.get('DVGO_HIP_SO') or os.path.join(_HERE, 'csrc', 'libdvgo_hip.so')
ABI_VERSION = 2

_lib = None


class _Rec2(ctypes.Structure):
    _fields_ = [('step', ctypes.c_int32), ('exp_d', ctypes.c_float), ('alpha', ctypes.c_float), ('T', ctypes.c_float)]


def lib():
    """The loaded library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f'{SO_PATH} is missing: build it with `python -m directvoxgo_amd.build` '
                '(hipcc --offload-arch=gfx950).  directvoxgo_amd has no CPU fallback.')
        _lib = ctypes.CDLL(SO_PATH)
        _lib.dvgo_abi_version.restype = ctypes.c_int
        v = _lib.dvgo_abi_version()
        if v != ABI_VERSION:
            raise RuntimeError(f'libdvgo_hip.so ABI {v} != expected {ABI_VERSION}: rebuild')
        if os.environ.get('DVGO_SHADE_VARIANT'):          # A/B runs (tools/): colour-head kernel variant bits
            _lib.dvgo_shade_variant(int(os.environ['DVGO_SHADE_VARIANT']))
    return _lib


_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_flt = ctypes.c_float


def check_input(x, name):
    """CHECK_INPUT of render_utils.cpp:40-42."""
    if not x.is_cuda:
        raise RuntimeError(f'{name} must be a CUDA tensor')
    if not x.is_contiguous():
        raise RuntimeError(f'{name} must be contiguous')


def check_f32(x, name):
    if x.dtype != torch.float32:
        # the reference dispatches float/double but the path runs in fp32
        raise RuntimeError(f'{name} must be float32, got {x.dtype}')


def ptr(t):
    return _vp(t.data_ptr()) if t is not None else _vp(0)


def stream_of(t):
    """torch's current stream on t's device as a raw hipStream_t (the C-level getter: a Stream object per launch costs
    more host time than the launch itself on small batches)."""
    return _vp(torch._C._cuda_getCurrentRawStream(t.device.index))


class _NoCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NOCTX = _NoCtx()


def device_of(t):
    """`torch.cuda.device_of(t)`, free when t already lives on the current device (the usual one-process-per-GPU case)."""
    if t.device.index == torch.cuda.current_device():
        return _NOCTX
    return torch.cuda.device_of(t)


_ERR = {-1: 'invalid argument', -2: 'size exceeds 32-bit launch range'}


# Optional per-entry-point timing with HIP events on the launching (= torch's current) stream.
# bench.py uses it to measure the average launch duration of the hot kernels inside its timed
# region; it is off (None) everywhere else.
_profile = None


def profile_start(names):
    global _profile
    _profile = {n: [] for n in names}


def profile_stop():
    """-> {name: (launches, total_ms)}; synchronises."""
    global _profile
    prof, _profile = _profile, None
    torch.cuda.synchronize()
    return {n: (len(ev), sum(a.elapsed_time(b) for a, b in ev)) for n, ev in (prof or {}).items()}


def call(name, *args):
    fn = getattr(lib(), name)
    if _profile is not None and name in _profile:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        _profile[name].append((a, b))
    else:
        rc = fn(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed: {_ERR.get(rc, "hipError %d" % rc)}')


def f3(x):
    """3 floats as a host array (model constants travel as kernel arguments in the fused path)."""
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().tolist()
    return (ctypes.c_float * 3)(*[float(v) for v in x])
