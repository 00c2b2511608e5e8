"""Checkpoint / wire format (row N5 of SURVEY.md section 8f).

Same on-disk dict as the reference (run.py:420-437): {'global_step', 'model_kwargs', 'model_state_dict',
'optimizer_state_dict'} written with torch.save; `load_model` rebuilds the model from 'model_kwargs' like
lib/utils.py:63-79, `load_checkpoint` resumes like lib/utils.py:53-60 and `MaskCache(path=...)` reads the coarse
geometry like lib/dvgo.py:586-593.

Layout.  Grids AND their Adam moments are written in the reference's contiguous [1,C,X,Y,Z] layout (the channels-last
storage of this implementation is an in-memory detail), so a file moves freely between the two code bases; on load the
optimizer state is re-laid to the parameter's own strides (`MaskedAdam._state_of`), because the update kernels walk raw
memory.

Safety.  Files are read with `torch.load(weights_only=True)`: nothing in the file is executed.  The reference stores
numpy values in 'model_kwargs' (`get_kwargs`, lib/dvgo.py:167-184: the bbox as ndarrays, `act_shift` as a numpy
scalar), which the weights-only unpickler refuses by default; exactly the numpy reconstructors needed for plain
numeric arrays / scalars are allow-listed for the duration of the load -- no other global is.
"""
import numpy as np
import torch


def _numpy_allow_list():
    import numpy
    core = getattr(numpy, '_core', None) or numpy.core
    ma = core.multiarray
    allow = [ma._reconstruct, numpy.ndarray, numpy.dtype, ma.scalar]
    # files written under numpy 1.x (the reference's environment) name the same functions through `numpy.core`
    allow += [(ma._reconstruct, 'numpy.core.multiarray._reconstruct'), (ma.scalar, 'numpy.core.multiarray.scalar')]
    for name in ('float32', 'float64', 'float16', 'int64', 'int32', 'int16', 'int8', 'uint8', 'bool'):
        allow.append(type(numpy.dtype(name)))        # numpy >= 1.25 pickles dtype instances by their DType class
    return allow


def safe_load(path, map_location='cpu'):
    """torch.load that executes nothing from the file (weights_only) yet accepts the reference's numpy kwargs."""
    with torch.serialization.safe_globals(_numpy_allow_list()):
        return torch.load(path, map_location=map_location, weights_only=True)


def _canonical(t):
    """the reference's layout for a grid-shaped tensor: contiguous [1,C,X,Y,Z]"""
    return t.contiguous() if (isinstance(t, torch.Tensor) and t.dim() == 5) else t


def _portable_state_dict(model):
    return {k: _canonical(v) for k, v in model.state_dict().items()}


def _portable_optimizer_state(optimizer):
    sd = optimizer.state_dict()
    return {'state': {i: {k: _canonical(v) for k, v in st.items()} for i, st in sd['state'].items()},
            'param_groups': sd['param_groups']}


def save_checkpoint(path, model, optimizer, global_step):
    """run.py:420-437"""
    torch.save({'global_step': global_step, 'model_kwargs': model.get_kwargs(),
                'model_state_dict': _portable_state_dict(model),
                'optimizer_state_dict': _portable_optimizer_state(optimizer) if optimizer is not None else None}, path)


def model_kwargs_of(ckpt):
    """'model_kwargs' as constructor arguments: derived quantities the constructor recomputes are dropped, numpy
    scalars become python numbers (lib/utils.py:66 passes the dict through as is)."""
    kwargs = dict(ckpt['model_kwargs'])
    kwargs.pop('act_shift', None); kwargs.pop('voxel_size_ratio', None)      # derived in __init__
    return {k: (v.item() if isinstance(v, np.generic) else v) for k, v in kwargs.items()}


def load_model(model_class, ckpt_path, **overrides):
    """lib/utils.py:63-79"""
    ckpt = safe_load(ckpt_path)
    kwargs = model_kwargs_of(ckpt)
    kwargs.update(overrides)
    model = model_class(**kwargs)
    model.load_state_dict(ckpt['model_state_dict'])
    return model


def load_checkpoint(model, optimizer, ckpt_path, no_reload_optimizer=False):
    """lib/utils.py:53-60.  The optimizer state arrives in the canonical contiguous layout, whoever wrote the file;
    `MaskedAdam` re-lays it to the parameters' strides on first use."""
    ckpt = safe_load(ckpt_path)
    model.load_state_dict(ckpt['model_state_dict'])
    if not no_reload_optimizer and optimizer is not None and ckpt.get('optimizer_state_dict') is not None:
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
    return model, optimizer, ckpt['global_step']
