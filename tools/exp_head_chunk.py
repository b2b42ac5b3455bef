"""Would running the 8 stacks of a layer in head chunks (projection of a chunk, then its attention) keep P in the memory-side cache?
Forward only: [qt_proj_group(G) -> qt_attn_fwd(G)] once against the same in chunks of 4 / 2 / 1 heads (cfg4 mesh; diagnostics)."""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
from qtmpnn import _lib, synthetic
from qtmpnn._lib import ptr
from qtmpnn.mesh import build_mesh
dev = torch.device('cuda', 0)
C, G, shape, B = 32, 8, (128, 128), 16
clips = [synthetic.make_ice_like(1000 + k, shape=shape, channels=5, n_frames=12)[0] for k in range(B)]
mask = synthetic.make_ice_like(40, shape=shape, channels=5, n_frames=2)[1]
x = torch.from_numpy(np.stack(clips)).to(dev)
mesh = build_mesh(src=(abs(abs(x[..., 0] - 0.5) - 0.5)).amax(dim=1), thresh=0.15, mask=mask)
N = mesh.N
xy, selfpair, eattr, rev = mesh.attn_geometry()
A = torch.randn(G, N, C, device=dev); W = torch.randn(G, C + 4, 4 * C, device=dev) * 0.1
ones = mesh.cheb_ones(1)
P = torch.empty(G, 4, N, C, device=dev); We = torch.randn(G, C, 2, device=dev)
out = torch.empty(G, N, C, device=dev); stats = torch.empty(G, N, 2, device=dev)
big = torch.empty(160 << 20, device=dev)          # 640 MB written between repetitions: every repetition starts cache-cold


def layer(hc):
    for h0 in range(0, G, hc):
        _lib.call('qt_proj_group', A.data_ptr() + 4 * h0 * N * C, C, N * C, 1, C, ptr(ones), W.data_ptr() + 4 * h0 * (C + 4) * 4 * C, None,
                  (C + 4) * 4 * C, hc, 4, C, P.data_ptr() + 4 * h0 * 4 * N * C, C, 4 * N * C, 0, N, ptr(mesh.n_dev))
        _lib.call('qt_attn_fwd', ptr(mesh.rowptr), ptr(mesh.col), ptr(xy), ptr(eattr), ptr(selfpair), P.data_ptr() + 4 * h0 * 4 * N * C, C,
                  We.data_ptr() + 4 * h0 * 2 * C, C, C, N, ptr(mesh.n_dev), 1.0, 7, None, out.data_ptr() + 4 * h0 * N * C,
                  stats.data_ptr() + 4 * h0 * 2 * N, hc, C, N * C, 4 * N * C, N * C)


def timeit(fn, reps=10):
    """Graph replay of [flush the caches with a 640 MB fill, layer] x reps minus the replay of the fills alone."""
    fn(); torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())

    def graph(body):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                big.zero_()
                body()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); b.synchronize()
        return a.elapsed_time(b) * 1e3 / reps
    return graph(fn) - graph(lambda: None)


ref = None
for hc in (8, 4, 2, 1):
    t = timeit(lambda: layer(hc))
    chk = float(out.double().sum())
    ref = chk if ref is None else ref
    print(f'chunks of {hc} heads: projection + attention forward {t:.1f} us   checksum {chk:.4f} (rel diff {abs(chk - ref) / abs(ref):.1e})')
