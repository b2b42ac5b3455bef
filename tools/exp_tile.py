"""Tile-resident multi-hop launches on frames of several 64 x 64 base cells (csrc/chebclip.hip, TILE = true) against one k_spmm
launch per hop: microseconds per recurrence, graph-replayed (the device, not the host call, is timed).

    python tools/exp_tile.py [cfg3|cfg4|cfg5]        (QT_LIB=<name>.so under tools/micro selects another build of the library)
cfg3: 8 clips of Moving-MNIST-like 128x128 (noise 0.05: nearly one node per pixel); cfg4: 16 ice-like clips 128x128 with land mask,
thresh 0.15 on dist_from_05; cfg5: 4 ice-like clips 256x256.
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
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
if cfg == 'cfg3':
    x, _ = synthetic.make_batch(3, 0, 8, 10, 20, n_digits=2, pixel_noise=0.05, canvas=(128, 128))
    img0, mask, thresh = torch.from_numpy(x[..., 0].max(axis=1)).to(dev), None, 0.1
    shapes = [(5, (4, 16)), (5, (16, 16)), (3, (4, 16)), (3, (16, 16)), (3, (16, 4))]
else:
    B, shape = (16, (128, 128)) if cfg == 'cfg4' else (4, (256, 256))
    clips = [synthetic.make_ice_like(1000 + k, shape=shape, channels=5, n_frames=2) for k in range(B)]
    img0 = torch.from_numpy(np.stack([abs(abs(c[0][..., 0].max(axis=0) - 0.5) - 0.5) for c in clips])).to(dev)
    mask, thresh = synthetic.make_ice_like(40, shape=shape, channels=5, n_frames=2)[1], 0.15
    shapes = [(7, (8, 32)), (7, (32,)), (3, (8, 32)), (3, (32, 4))]
mesh = build_mesh(src=img0, thresh=thresh, mask=mask, static=True)
tl = mesh.tiles
print(f'{cfg}: N = {mesh.n_valid} (capacity {mesh.N}), E = {mesh.E}, tiles per clip {tl["T"]}, '
      f'rows per tile {mesh.n_valid / (mesh.B * tl["T"]):.0f}, halo entries per tile {float(tl["cnt"].view(-1, 32)[:, 2].float().mean()):.0f}')


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


for K, widths in shapes:
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
    t_cf, t_cb = timeit(lambda: ops.clip_planes(mesh, Zs, TZ, K)), timeit(lambda: ops.clip_clenshaw(mesh, G, K))
    S, ncu = sum(widths) // 4, _lib.value('qt_num_cus')
    per = max(ncu // (mesh.B * tl['T']), 1)
    nl = -(S // -per)
    split = ' + '.join(str(mesh.B * tl['T'] * min(per, S - i * per)) for i in range(nl))
    print(f'K={K} widths={widths}: forward per-hop {t_hf:7.2f} us  tile {t_cf:7.2f} us | backward per-hop {t_hb:7.2f} us  tile {t_cb:7.2f} us'
          f'   ({mesh.B} clips x {tl["T"]} tiles x {S} slices = {mesh.B * tl["T"] * S} workgroups as {nl} launch(es) of {split} co-resident '
          f'workgroups; error word {int(tl["err"])})')
