"""Procedural Moving-MNIST-like clips (no dataset download is possible here).

Statistics follow the reference generator data/mod_moving_mnist.py:72-161
(random start in the inner canvas, +-1 px/frame velocity with N(0, 0.25)
velocity noise, reflection at the borders, positions truncated to uint8,
digits overlapped by `max`, N(0, pixel_noise) white noise, frames transposed
with swapaxes(1, -1)).  MNIST glyphs are replaced by procedural stroke glyphs
in [0, 1] because the reference fetches MNIST over the network
(data/mod_moving_mnist.py:47).

Pure numpy; shared by bench.py, the tests and tests/golden/make_golden.py.
"""
import numpy as np


def make_glyph(rng, size=28):
    """A digit-like glyph: a random thick polyline, values in [0, 1]."""
    n_pts = int(rng.integers(3, 6))
    pts = rng.uniform(0.18 * size, 0.82 * size, size=(n_pts, 2))
    rr, cc = np.mgrid[0:size, 0:size].astype(np.float64)
    d2 = np.full((size, size), np.inf)
    for a, b in zip(pts[:-1], pts[1:]):
        ab = b - a
        t = ((rr - a[0]) * ab[0] + (cc - a[1]) * ab[1]) / max(float(ab @ ab), 1e-9)
        t = np.clip(t, 0.0, 1.0)
        d2 = np.minimum(d2, (rr - (a[0] + t * ab[0])) ** 2 + (cc - (a[1] + t * ab[1])) ** 2)
    thick = rng.uniform(1.2, 2.2)
    g = np.clip(1.5 - np.sqrt(d2) / thick, 0.0, 1.0)
    g[g < 0.05] = 0.0
    return g.astype(np.float32)


def _trajectory(rng, n_frames, canvas, digit, velocity_noise):
    inner = np.array(canvas) - np.array(digit)
    x, y = rng.random(2) * inner
    vx, vy = rng.choice([-1, 1]), rng.choice([-1, 1])
    xs, ys = [], []
    for _ in range(n_frames):
        ny, nx = rng.normal(0, velocity_noise, 2) if velocity_noise > 0 else (0.0, 0.0)
        y += vy + ny
        x += vx + nx
        if x <= 0:
            x, vx = 0, -vx
        if x >= inner[1]:
            x, vx = inner[1], -vx
        if y <= 0:
            y, vy = 0, -vy
        if y >= inner[0]:
            y, vy = inner[0], -vy
        xs.append(x)
        ys.append(y)
    return np.array(xs, dtype=np.uint8), np.array(ys, dtype=np.uint8)


def make_clip(seed, canvas=(64, 64), digit=(28, 28), n_digits=1, n_frames=20,
              pixel_noise=0.05, velocity_noise=0.25):
    """One clip (n_frames, W, H, 1) float32."""
    rng = np.random.default_rng(seed)
    layers = []
    for _ in range(n_digits):
        g = make_glyph(rng, digit[0])
        if digit[1] != digit[0]:
            g = g[:, np.linspace(0, digit[0] - 1, digit[1]).astype(int)]
        xs, ys = _trajectory(rng, n_frames, canvas, digit, velocity_noise)
        c = np.zeros((n_frames, *canvas), dtype=np.float32)
        for i, (x, y) in enumerate(zip(xs, ys)):
            c[i, y:y + digit[1], x:x + digit[0]] = g[:canvas[0] - y, :canvas[1] - x]
        layers.append(c)
    imgs = np.max(np.stack(layers), axis=0)
    if pixel_noise > 0:
        imgs = imgs + rng.normal(0, pixel_noise, size=imgs.shape)
    imgs = np.swapaxes(imgs, 1, -1)
    return imgs[..., None].astype(np.float32)


def make_batch(config_id, clip0, batch, t_in, t_out, **kw):
    """(x (B,T_in,W,H,1), y (B,T_out,W,H,1)); seed = 1000*config + clip index."""
    xs, ys = [], []
    for i in range(batch):
        c = make_clip(1000 * config_id + clip0 + i, n_frames=t_in + t_out, **kw)
        xs.append(c[:t_in])
        ys.append(c[-t_out:])
    return np.ascontiguousarray(np.stack(xs)), np.ascontiguousarray(np.stack(ys))


def make_ice_like(seed, shape=(128, 128), channels=5, n_frames=18):
    """Smooth multi-channel fields + a land mask (ERA5/GLORYS stand-in)."""
    rng = np.random.default_rng(seed)
    n, m = shape

    def smooth(a, k):
        for ax in (-2, -1):
            ker = np.exp(-0.5 * (np.arange(-3 * k, 3 * k + 1) / k) ** 2)
            ker /= ker.sum()
            a = np.apply_along_axis(lambda v: np.convolve(np.pad(v, 3 * k, mode='edge'), ker, 'valid'), ax, a)
        return a

    base = smooth(rng.normal(size=(channels, n, m)), 6)
    drift = smooth(rng.normal(size=(channels, n, m)), 6)
    frames = []
    for t in range(n_frames):
        f = base + 0.08 * t * drift
        f = (f - f.min(axis=(1, 2), keepdims=True)) / (np.ptp(f, axis=(1, 2), keepdims=True) + 1e-9)
        f[0] = 1.0 / (1.0 + np.exp(-25.0 * (f[0] - 0.5)))
        frames.append(np.moveaxis(f, 0, -1))
    land = smooth(rng.normal(size=(n, m)), 8)
    mask = land > np.quantile(land, 0.75)
    return np.stack(frames).astype(np.float32), mask
