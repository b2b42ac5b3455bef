"""spmm timing at the bench shapes (diagnostics)."""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import torch
from qtmpnn import synthetic
from qtmpnn.mesh import build_mesh, spmm
dev = torch.device('cuda', 0)
x, _ = synthetic.make_batch(2, 0, 32, 10, 1, n_digits=2, pixel_noise=0.05)
mesh = build_mesh(src=torch.from_numpy(x[..., 0]).to(dev).amax(dim=1), thresh=0.1)
N = mesh.N
def timeit(fn, reps=100):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
for C in (4, 16, 20, 32, 68):
    Z = torch.randn(N, C, device=dev); out = torch.empty(N, C, device=dev); p_ = torch.randn(N, C, device=dev)
    t1 = timeit(lambda: spmm(mesh, Z, 2.0, p_, -1.0, None, 0.0, out, C))
    t0 = timeit(lambda: spmm(mesh, Z, 1.0, None, 0.0, None, 0.0, out, C))
    by = 4.0 * (N + 1) + 8.0 * mesh.E + 8.0 * N * C
    print(f'N {N} E {mesh.E} C {C}: with addend {t1:.2f} us ({(by + 4.0 * N * C) / t1 / 1e3:.0f} GB/s incl addend)  plain {t0:.2f} us ({by / t0 / 1e3:.0f} GB/s)')
print('--- inside a hipGraph (100 launches per replay)')
for C in (4, 16, 20, 32, 68):
    Z = torch.randn(N, C, device=dev); out = torch.empty(N, C, device=dev); p_ = torch.randn(N, C, device=dev)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        spmm(mesh, Z, 2.0, p_, -1.0, None, 0.0, out, C)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g, stream=side):
        for _ in range(100):
            spmm(mesh, Z, 2.0, p_, -1.0, None, 0.0, out, C)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); b.synchronize()
    print(f'N graph C {C}: {a.elapsed_time(b) * 10:.2f} us per launch')
print('--- two column parts in one launch (hipGraph, 100 launches per replay)')
from qtmpnn.mesh import spmm2
for Ca, Cb in ((4, 16), (16, 4), (16, 16)):
    xs = [torch.randn(N, Ca, device=dev), torch.randn(N, Cb, device=dev)]
    ps = [torch.randn(N, Ca, device=dev), torch.randn(N, Cb, device=dev)]
    outs = [torch.empty(N, Ca, device=dev), torch.empty(N, Cb, device=dev)]
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        spmm2(mesh, xs, 2.0, ps, -1.0, None, 0.0, outs)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(100):
            spmm2(mesh, xs, 2.0, ps, -1.0, None, 0.0, outs)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); b.synchronize()
    print(f'N graph parts {Ca}+{Cb}: {a.elapsed_time(b) * 10:.2f} us per launch')
