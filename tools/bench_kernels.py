"""Time single kernels of the path at the bench shapes (diagnostics; not part of the product)."""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
from qtmpnn import _lib, synthetic
from qtmpnn._lib import ptr
from qtmpnn.mesh import build_mesh, spmm
dev = torch.device('cuda', 0)
x, _ = synthetic.make_batch(2, 0, 32, 10, 1, n_digits=2, pixel_noise=0.05)
mesh = build_mesh(src=torch.from_numpy(x[..., 0]).to(dev).amax(dim=1), thresh=0.1)
N = mesh.N
print('N', N, 'E', mesh.E)
def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
K, C, Co = 5, 20, 64
Z = torch.randn(N, C, device=dev); TZ = torch.randn(K - 1, N, C, device=dev)
S = mesh.cheb_ones(3); W = torch.randn(K * C + 4, Co, device=dev); Y = torch.empty(N, Co, device=dev)
dense = lambda: _lib.call('qt_dense', ptr(Z), ptr(TZ), K, C, ptr(W), ptr(S), 4, ptr(W[K * C:]), 1, Co, N, None, 0, None, 0, None, ptr(Y))
print('gemm fwd  (N x 104)(104 x 64)  us', round(timeit(dense), 2))
G = torch.randn(N, Co, device=dev); Wt = torch.randn(Co, K * C, device=dev); gT = torch.empty(K, N, C, device=dev)
bwd = lambda: _lib.call('qt_dense', ptr(G), None, 1, Co, ptr(Wt), None, 0, None, K, C, N, None, 0, None, 0, None, ptr(gT))
print('gemm bwd-data (N x 64)(64 x 100) us', round(timeit(bwd), 2))
nblk = _lib.value('qt_wgrad_blocks', N); part = torch.zeros(nblk, K * C + 4, Co, device=dev)
wg = lambda: _lib.call('qt_wgrad', ptr(Z), 0, ptr(TZ), None, 0, None, K, C, 0, ptr(S), 4, ptr(G), Co, N, None, 1, ptr(part), 0)
print('wgrad us', round(timeit(wg), 2))
out = torch.empty(N, C, device=dev); p_ = torch.randn(N, C, device=dev)
print('spmm C=20 us', round(timeit(lambda: spmm(mesh, Z, 2.0, p_, -1.0, None, 0.0, out, C)), 2))
print('spmm C=20 no addend us', round(timeit(lambda: spmm(mesh, Z, 1.0, None, 0.0, None, 0.0, out, C)), 2))
cp = torch.empty(N * 164, device=dev); src = torch.randn(N * 164, device=dev)
print('copy of the same bytes (N x 164 floats) us', round(timeit(lambda: cp.copy_(src)), 2))
# interleaved operand: one (N, K*C) matrix instead of K planes (N, C)
Zi = torch.randn(N, K * C, device=dev)
densei = lambda: _lib.call('qt_dense', ptr(Zi), None, 1, K * C, ptr(W), ptr(S), 4, ptr(W[K * C:]), 1, Co, N, None, 0, None, 0, None, ptr(Y))
print('gemm fwd interleaved A (N x 100 rows contiguous) us', round(timeit(densei), 2))
gTi = torch.empty(N, K * C, device=dev)
bwdi = lambda: _lib.call('qt_dense', ptr(G), None, 1, Co, ptr(Wt), None, 0, None, 1, K * C, N, None, 0, None, 0, None, ptr(gTi))
print('gemm bwd-data interleaved out us', round(timeit(bwdi), 2))
wgi = lambda: _lib.call('qt_wgrad', ptr(Zi), 0, None, None, 0, None, 1, K * C, 0, ptr(S), 4, ptr(G), Co, N, None, 1, ptr(part), 0)
print('wgrad interleaved A us', round(timeit(wgi), 2))
