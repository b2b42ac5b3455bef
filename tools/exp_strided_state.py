"""Does the gate kernel pay for reading H / C as 64-byte column slices of a 272-byte-pitch matrix (the re-mesh transfer's
output) instead of dense (N, 16) matrices?  (diagnostics)"""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
from qtmpnn import ops, synthetic
from qtmpnn.mesh import build_mesh
dev = torch.device('cuda', 0)
x, _ = synthetic.make_batch(2, 0, 32, 10, 1, n_digits=2, pixel_noise=0.05)
mesh = build_mesh(src=torch.from_numpy(x[..., 0]).to(dev).amax(dim=1), thresh=0.1)
N, h, K = mesh.N, 16, 3
torch.manual_seed(0)
wide = torch.randn(N, 68, device=dev)
X = torch.randn(N, 4, device=dev)
W = torch.randn(K * 20 + 4, 64, device=dev) * 0.1
wc, b, ln = torch.randn(3, h, device=dev), torch.randn(4, h, device=dev), torch.randn(4, h, device=dev)


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); e.record(); e.synchronize()
    return a.elapsed_time(e) * 1e3 / reps


for name, H, C in (('strided views (pitch 272 B)', wide[:, 0:16], wide[:, 16:32]),
                   ('dense matrices', wide[:, 0:16].contiguous(), wide[:, 16:32].contiguous())):
    with torch.no_grad():
        t = timeit(lambda: ops.gate_cell(X, H, W, C, wc, b, ln, mesh, K, 1))
    print(f'{name}: forward cell (2 spmm + fused gate GEMM) {t:.1f} us')
