"""Clip-resident state transfer (csrc/remeshclip.hip) against the general kernels, bench shape: 32 clips, 64x64, 68 state
channels as [4 | 16 x 4], forward and backward, on (a) two noisy meshes, (b) noisy -> sparse (big cells): us per transfer."""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
from qtmpnn import ops, synthetic
from qtmpnn.mesh import build_mesh
dev = torch.device('cuda', 0)


def mesh(seed, noise, static=True):
    x, _ = synthetic.make_batch(seed, 0, 32, 2, 2, n_digits=2, pixel_noise=noise, canvas=(64, 64))
    return build_mesh(src=torch.from_numpy(x[:, 0, ..., 0]).to(dev), thresh=0.1, static=static)


def timeit(fn, reps=20):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(reps):
            fn()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for name, (na, nb) in {'noisy -> noisy': (0.05, 0.05), 'noisy -> sparse': (0.05, 0.0), 'sparse -> noisy': (0.0, 0.05)}.items():
    old, new = mesh(3, na), mesh(4, nb)
    widths = [4, 16, 16, 16, 16]
    parts = [torch.randn(old.N, w, device=dev) for w in widths]
    outs = [torch.empty(new.N, w, device=dev) for w in widths]
    gin = [torch.randn(new.N, w, device=dev) for w in widths]
    gout = [torch.empty(old.N, w, device=dev) for w in widths]
    res = {}
    for clip in (True, False):
        ops._CLIP_REMESH = clip
        res[clip] = (timeit(lambda: ops._remesh_raw(new, old, parts, outs, False, True)),
                     timeit(lambda: ops._remesh_raw(old, new, gin, gout, True, False)))
    print(f'{name}: N {old.n_valid} -> {new.n_valid}: forward general {res[False][0]:6.2f} us  clip {res[True][0]:6.2f} us | '
          f'backward general {res[False][1]:6.2f} us  clip {res[True][1]:6.2f} us')
