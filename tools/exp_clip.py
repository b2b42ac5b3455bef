"""Clip-resident multi-hop launch (csrc/chebclip.hip) against one k_spmm launch per hop, on the bench's input mesh
(32 clips, 64x64, noise 0.05): microseconds per recurrence, graph-replayed (the device, not the host call, is timed).

    python tools/exp_clip.py [noise]
"""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
from qtmpnn import _lib
if os.environ.get('QT_LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'micro', os.environ['QT_LIB'])
from qtmpnn import ops, synthetic
from qtmpnn.mesh import build_mesh, spmm2

dev = torch.device('cuda', 0)
noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
x, _ = synthetic.make_batch(2, 0, 32, 10, 10, n_digits=2, pixel_noise=noise, canvas=(64, 64))
img0 = torch.from_numpy(x[..., 0].max(axis=1)).to(dev)
mesh = build_mesh(src=img0, thresh=0.1, static=True)
print(f'mesh: N = {mesh.n_valid} (capacity {mesh.N}), E = {mesh.E}')


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


for K, widths in [(5, (4, 16)), (5, (16,)), (3, (4, 16)), (3, (16,)), (3, (16, 4)), (5, (4,)), (2, (16,))]:
    N = mesh.N
    Zs = [torch.randn(N, w, device=dev) for w in widths]
    TZ = [torch.empty(K - 1, N, w, device=dev) for w in widths]
    G = [torch.randn(K, N, w, device=dev) for w in widths]

    def hops_fwd():
        for k in range(1, K):
            if k == 1:
                spmm2(mesh, Zs, 1.0, None, 0.0, None, 0.0, [T[0] for T in TZ])
            else:
                spmm2(mesh, [T[k - 2] for T in TZ], 2.0, Zs if k == 2 else [T[k - 3] for T in TZ], -1.0, None, 0.0, [T[k - 1] for T in TZ])

    def hops_bwd():
        for k in range(K - 2, 0, -1):
            spmm2(mesh, [g[k + 1] for g in G], 2.0, [g[k] for g in G], 1.0, [g[k + 2] for g in G] if k + 2 < K else None, -1.0, [g[k] for g in G])
        spmm2(mesh, [g[1] for g in G], 1.0, [g[0] for g in G], 1.0, [g[2] for g in G] if K > 2 else None, -1.0, [g[0] for g in G])
    t_hf, t_hb = timeit(hops_fwd), timeit(hops_bwd)
    from qtmpnn import _lib
    per_w = {}
    for wd in (4, 2):
        per_w[wd] = (timeit(lambda: ops.clip_planes(mesh, Zs, TZ, K, width=wd)), timeit(lambda: ops.clip_clenshaw(mesh, G, K, width=wd)))
    print(f'   slice width 4: {per_w[4][0]:6.2f} / {per_w[4][1]:6.2f} us   width 2: {per_w[2][0]:6.2f} / {per_w[2][1]:6.2f} us   (automatic below)')
    t_cf, t_cb = timeit(lambda: ops.clip_planes(mesh, Zs, TZ, K)), timeit(lambda: ops.clip_clenshaw(mesh, G, K))
    nv, C = mesh.n_valid, sum(widths)
    idx = 4.0 * (nv + 1) + 8.0 * mesh.E
    print(f'K={K} widths={widths}: forward per-hop {t_hf:7.2f} us  clip {t_cf:7.2f} us ({(idx + 4.0 * nv * C * K) / t_cf / 1e3:6.0f} GB/s of its own bytes) | '
          f'backward per-hop {t_hb:7.2f} us  clip {t_cb:7.2f} us ({(idx + 4.0 * nv * C * (K + 1)) / t_cb / 1e3:6.0f} GB/s)')
