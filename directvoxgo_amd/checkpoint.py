"""Checkpoint / wire format (row N5 of SURVEY.md section 8f).

Same on-disk dict as the reference (run.py:420-437): {'global_step', 'model_kwargs',
'model_state_dict', 'optimizer_state_dict'} written with torch.save; `load_model` rebuilds the
model from 'model_kwargs' like lib/utils.py:63-79.  Grids are saved in the reference's contiguous
[1,C,X,Y,Z] layout (the channels-last storage is an in-memory detail), so checkpoints move freely
between this implementation and the reference; `MaskCache(path=...)` reads them (lib/dvgo.py:586-593).
"""
import torch


def _portable_state_dict(model):
    return {k: (v.contiguous() if v.dim() == 5 else v) for k, v in model.state_dict().items()}


def save_checkpoint(path, model, optimizer, global_step):
    torch.save({'global_step': global_step, 'model_kwargs': model.get_kwargs(),
                'model_state_dict': _portable_state_dict(model),
                'optimizer_state_dict': optimizer.state_dict() if optimizer is not None else None}, path)


def load_model(model_class, ckpt_path, **overrides):
    """lib/utils.py:63-79.  `weights_only=False` because 'model_kwargs' carries numpy arrays exactly as the
    reference writes them -- only load checkpoints you trust."""
    ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=False)
    kwargs = dict(ckpt['model_kwargs'])
    kwargs.pop('act_shift', None); kwargs.pop('voxel_size_ratio', None)      # derived in __init__
    kwargs.update(overrides)
    model = model_class(**kwargs)
    model.load_state_dict(ckpt['model_state_dict'])
    return model


def load_checkpoint(model, optimizer, ckpt_path, no_reload_optimizer=False):
    """lib/utils.py:53-60"""
    ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=False)
    model.load_state_dict(ckpt['model_state_dict'])
    if not no_reload_optimizer and optimizer is not None and ckpt.get('optimizer_state_dict') is not None:
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
    return model, optimizer, ckpt['global_step']
