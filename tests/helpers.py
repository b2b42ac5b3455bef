"""Shared helpers for the GPU parity tests (oracle = checker only)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
RTOL, ATOL = 1e-4, 1e-5          # north_star: 1e-4 rel fp32


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def dist_from_05(arr):
    return abs(abs(arr - 0.5) - 0.5)


def dev():
    return torch.device('cuda', 0)


def close(a, b, rtol=RTOL, atol=ATOL, msg=''):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


def grad_close(a, b, rtol=1e-4, rel_atol=5e-5, msg='', floor=1e-3):
    """Gradients: north_star's rtol 1e-4 with an absolute floor of 5e-5 x the tensor's largest entry (sums over ~1e3..1e5 nodes;
    round 1 used 1e-3 / 1e-4; the default backward is exact fp32, the opt-in split-bf16 data gradient passes the same bound:
    tests/test_gpu_headline.py reports its worst per-tensor error against the exact gradients).
    `floor` bounds that scale from below (gradients that are exactly zero in exact arithmetic are pure rounding noise,
    e.g. the key bias of a softmax attention: a shift of every key moves all scores of a target equally)."""
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    close(a, b, rtol=rtol, atol=rel_atol * max(floor, float(np.abs(b).max())), msg=msg)


def load_state(module, g, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}
    module.load_state_dict(sd, strict=True)


def mesh_from_golden_graph(g):
    """Build the device mesh from a golden graph case's input image (channel 0, max over samples)."""
    from model.graph_functions import _criterion
    from qtmpnn.mesh import build_mesh
    x = torch.from_numpy(g['x']).to(dev())
    n, m = x.shape[1:3]
    img0 = x[..., 0].amax(dim=0, keepdim=True)
    tf = dist_from_05 if bool(g['has_transform']) else None
    mask = g['mask'] if 'mask' in g.files else None
    hir = g['hir'] if 'hir' in g.files else None
    return build_mesh(src=_criterion(img0, n, m, 64, tf), n=n, m=m, thresh=float(g['thresh']),
                      condition=str(g['condition']), mask=mask, high_interest_region=hir)


def climatology_from_base(base):
    """(1, 365, w, h) daily normals from the (w, h) base field a fixture stores (the formula of tests/golden/make_golden.py)."""
    d = np.arange(365, dtype=np.float32)[:, None, None]
    return (base[None] * (0.5 + 0.5 * np.cos(2 * np.pi * d / 365.0)) + 0.001 * d)[None].astype(np.float32)


class TinyLoader(list):
    """In-memory stand-in for the reference's DataLoader(batch_size=1): items (x (1, T, W, H, C), y, launch_date) and
    `.dataset.image_shape` (model/mpnnlstm.py:201, 219-221)."""

    def __init__(self, items, image_shape):
        super().__init__(items)
        self.dataset = type('DS', (), {'image_shape': tuple(image_shape)})()
