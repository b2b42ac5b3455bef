"""Re-mesh state transfer (qt_remesh: [val4 | H0 | H1 | C0 | C1] -> 5 dense parts) at the bench shapes, forward and backward
direction, graph-replayed (diagnostics).  Meshes: an input-frame mesh -> the mesh of the next noisy frame."""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import torch
from qtmpnn import _lib
if os.environ.get('QT_LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'micro', os.environ['QT_LIB'])
from qtmpnn import synthetic, ops
from qtmpnn.mesh import build_mesh
dev = torch.device("cuda", 0)
torch.manual_seed(0)
def mesh(seed, noise, B=32):
    # (one frame per clip: the decoder's meshes come from single predicted frames; with noise 0.05 they are ~92 % 1x1 nodes)
    x, _ = synthetic.make_batch(2, seed, B, 1, 1, n_digits=2, pixel_noise=noise)
    return build_mesh(src=torch.from_numpy(x[:, 0, ..., 0]).to(dev), thresh=0.1, static=True)
def graph_time(fn, reps=20):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
for noise in (0.15, 0.05):
    old = mesh(100, noise)
    # the new mesh is decomposed from per-node values on the old one, like the decoder's re-mesh (direct row indices exist then)
    new = build_mesh(prev=(torch.rand(old.N, device=dev) * (4 * noise), old), thresh=0.1, static=True)
    widths = [4, 16, 16, 16, 16]
    parts = [torch.randn(old.N, w, device=dev) for w in widths]
    outs = [torch.empty(new.N, w, device=dev) for w in widths]
    gouts = [torch.randn(new.N, w, device=dev) for w in widths]
    gins = [torch.empty(old.N, w, device=dev) for w in widths]
    f = graph_time(lambda: ops._remesh_raw(new, old, parts, outs, False, True))
    b = graph_time(lambda: ops._remesh_raw(old, new, gouts, gins, True, False))
    nv_o, nv_n = old.n_valid, new.n_valid
    byts = (nv_o + nv_n) * 68 * 4
    print(f'noise {noise}: N {nv_o} -> {nv_n}: forward {f:.2f} us ({byts / f / 1e3:.0f} GB/s of rows), backward {b:.2f} us   '
          f'checksum {sum(float(o[:nv_n].double().sum()) for o in outs):.4f} {sum(float(o[:nv_o].double().sum()) for o in gins):.4f}')
