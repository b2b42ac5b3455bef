"""Time the edge-softmax attention kernels (TransformerConv) at the cfg4 shapes (diagnostics).

    python tools/bench_attn.py [C] [heads] [rows|planes]
Mesh: 16 ice-like 128x128 clips, land mask, transform_func, thresh 0.15 (BASELINE configs[3]).  Prints the launch times and
the compulsory bytes (every proj / g / out row once, CSR once) and gather bytes (rows re-read per edge) they move.
"""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
from qtmpnn import _lib, synthetic
from qtmpnn._lib import ptr
from qtmpnn.mesh import build_mesh
dev = torch.device('cuda', 0)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shape, B = (128, 128), 16
clips = [synthetic.make_ice_like(1000 + k, shape=shape, channels=5, n_frames=12)[0] for k in range(B)]
mask = synthetic.make_ice_like(40, shape=shape, channels=5, n_frames=2)[1]
x = torch.from_numpy(np.stack(clips)).to(dev)
tf = lambda a: abs(abs(a - 0.5) - 0.5)
src = tf(x[..., 0]).amax(dim=1)
mesh = build_mesh(src=src, thresh=0.15, mask=mask)
N, E = mesh.N, mesh.E
xy, selfpair, eattr, rev = mesh.attn_geometry()
print('N', N, 'E', E, 'self pairs', int((selfpair > 0).sum()) if selfpair is not None else 0, 'C', C)
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1
planes = (sys.argv[3] if len(sys.argv) > 3 else 'rows') == 'planes'
print('heads', G, 'layout', 'planes (G, 4, N, C)' if planes else 'rows (N, G 4C)')
proj = torch.randn(G * N * 4 * C, device=dev)
We = torch.randn(G, C, 2, device=dev)
out = torch.empty(G * N * C, device=dev); stats = torch.empty(G, N, 2, device=dev)
g = torch.randn(G * N * C, device=dev); gproj = torch.empty_like(proj); Dn = torch.empty(G, N, device=dev)
coef = torch.empty(G, rev.numel() + N, 2, device=dev)
nblk = _lib.value('qt_attn_blocks', N, C); part = torch.zeros(nblk, G * 2 * C, device=dev)
ld, ps, hs, ld_o, hs_o = (C, N * C, 4 * N * C, C, N * C) if planes else (G * 4 * C, C, 4 * C, G * C, C)
common = (ptr(mesh.rowptr), ptr(mesh.col), ptr(xy), ptr(eattr) if os.environ.get('QT_NO_EATTR') != '1' else None, ptr(selfpair), ptr(proj), ld, ptr(We), C, C, N, ptr(mesh.n_dev))


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        for _ in range(reps):
            fn()
    gr.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); gr.replay(); b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for keep in (1.0, 0.9):
    fwd = lambda: _lib.call('qt_attn_fwd', *common, keep, 7, None, ptr(out), ptr(stats), G, ld_o, ps, hs, hs_o)
    bwd = lambda: _lib.call('qt_attn_bwd', *common, keep, 7, None, ptr(g), ld_o, ptr(stats), ptr(out), ld_o, ptr(gproj), ptr(part), 0,
                              ptr(rev), ptr(coef), rev.numel(), G, 0, ps, hs, hs_o, hs_o)
    tf_, tb = timeit(fwd), timeit(bwd)
    cf = 4 * G * (N * 5 * C + N * 2 + 2 * E + N)                     # proj rows + out + stats + CSR
    gf = 4 * G * ((E + N) * 2 * C + N * 3 * C)                       # k, v per edge; q, skip, out per node
    cb = 4 * G * (N * 4 * C + N * C + N * 4 * C + 2 * 2 * E)         # proj + g + gproj, CSR twice
    print(f'keep {keep}: fwd {tf_:.1f} us ({cf / tf_ / 1e3:.0f} GB/s compulsory, {gf / tf_ / 1e3:.0f} GB/s gathered)   '
          f'bwd (target + source) {tb:.1f} us ({cb / tb / 1e3:.0f} GB/s compulsory)')
