"""Helpers of the reference's model/utils.py that the hot path and its callers use."""
import datetime

import numpy as np
import torch

_PE_CACHE = {}


def positional_grid(w, h, device=None, dtype=torch.float32):
    """(w, h, 2) grid: channel 0 = column / h, channel 1 = row / w (model/utils.py:37-45).
    Constant per image shape, so it is built once per (shape, device) instead of on every call."""
    key = (w, h, str(device), dtype)
    if key not in _PE_CACHE:
        g = np.empty((w, h, 2), dtype=np.float64)
        g[..., 0] = (np.arange(h, dtype=np.float64) / h)[None, :]
        g[..., 1] = (np.arange(w, dtype=np.float64) / w)[:, None]
        _PE_CACHE[key] = torch.from_numpy(g).to(dtype).to(device)
    return _PE_CACHE[key]


def add_positional_encoding(x):
    """(n_samples, w, h, c) -> (n_samples, w, h, c + 2), tensors or arrays (model/utils.py:30-52)."""
    assert len(x.shape) == 4, f'array should be 4-dimensional (n_samples, w, h, c); got {x.shape}'
    n, w, h, _ = x.shape
    if isinstance(x, torch.Tensor):
        pe = positional_grid(w, h, x.device, x.dtype)
        return torch.cat((x, pe.unsqueeze(0).expand(n, w, h, 2)), dim=-1)
    pe = positional_grid(w, h).numpy().astype(x.dtype)
    return np.concatenate((x, np.broadcast_to(pe, (n, w, h, 2))), axis=-1)


def get_n_params(model):
    """Number of parameters of a torch module (model/utils.py:19-27)."""
    return sum(p.numel() for p in model.parameters())


def normalize(arr):
    """Per-variable min-max scaling over axes (0, 2, 3, 4) (model/utils.py:70-73)."""
    lo = np.min(arr, (0, 2, 3, 4))[:, None, None, None]
    hi = np.max(arr, (0, 2, 3, 4))[:, None, None, None]
    return (arr - lo) / (hi - lo)


def int_to_datetime(x):
    """Nanosecond timestamp -> datetime (model/utils.py:75-76)."""
    return datetime.datetime.fromtimestamp(x / 1e9)


def round_to_day(dt):
    return datetime.datetime(*dt.timetuple()[:3])
