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


class TinyMovingMNISTDataset(torch.utils.data.Dataset):
    """Stand-in for the reference's ModMovingMNISTDataset (data/mod_moving_mnist.py:8-38; it downloads MNIST): the same
    constructor arguments, `.x (n, T_in, W, H, 1)`, `.y (n, T_out, W, H, 1)`, `.frame_id`, `.image_shape` and item layout
    `(x, y, frame_id)`, filled by the procedural generator."""

    def __init__(self, n_samples, input_timesteps, output_timesteps, n_digits=1, gap=0, canvas_size=(32, 32),
                 digit_size=(12, 12), pixel_noise=0.05, velocity_noise=0.25, seed=0):
        from qtmpnn import synthetic
        clips = [synthetic.make_clip(7000 + 100 * seed + i, canvas=canvas_size, digit=digit_size, n_digits=n_digits,
                                     n_frames=input_timesteps + gap + output_timesteps, pixel_noise=pixel_noise,
                                     velocity_noise=velocity_noise) for i in range(n_samples)]
        self.x = np.stack([c[:input_timesteps] for c in clips]).astype(np.float32)
        self.y = np.stack([c[-output_timesteps:] for c in clips]).astype(np.float32)
        self.frame_id = np.arange(len(self.y)).astype(np.float32)
        self.image_shape = self.x.shape[2:4]

    def __len__(self):
        return len(self.y)

    def __getitem__(self, idx):
        return self.x[idx], self.y[idx], self.frame_id[idx]


class TinyIceDataset(torch.utils.data.Dataset):
    """Stand-in for the reference's IceDataset (ice_dataset.py:7-17; it needs xarray + the ERA5 / GLORYS files): `.x (n, T_in,
    lat, lon, C)`, `.y (n, T_out, lat, lon, 1)`, `.launch_dates` (int64 ns), `.image_shape`, items `(x, y, launch_date)`."""

    def __init__(self, n_samples, input_timesteps, output_timesteps, shape, channels=5, seed=0, first_day=(2010, 3, 1)):
        import datetime
        from qtmpnn import synthetic
        clips = [synthetic.make_ice_like(8000 + 100 * seed + i, shape=shape, channels=channels,
                                         n_frames=input_timesteps + output_timesteps)[0] for i in range(n_samples)]
        self.x = np.stack([c[:input_timesteps] for c in clips]).astype(np.float32)
        self.y = np.stack([c[input_timesteps:, ..., :1] for c in clips]).astype(np.float32)
        t0 = int(datetime.datetime(*first_day, 12, tzinfo=datetime.timezone.utc).timestamp()) * 10 ** 9
        self.launch_dates = np.array([t0 + 86400 * 10 ** 9 * i for i in range(n_samples)], dtype=np.int64)
        self.image_shape = self.x[0].shape[1:-1]

    def __len__(self):
        return len(self.y)

    def __getitem__(self, idx):
        return self.x[idx], self.y[idx], self.launch_dates[idx]
