"""Head GEMM shapes: skinny VALU kernel vs the MFMA kernel (QT_GEMM_NO_SKINNY=1) (diagnostics)."""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import torch
from qtmpnn import _lib
from qtmpnn._lib import ptr
dev = torch.device('cuda', 0)
N = 120014
def timeit(fn, reps=50):
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
def shape(name, K, Ca, Cab, Co, Kb=1, Cb=None, Cbb=0, ks=4, act=0):
    Za = torch.randn(N, Ca, device=dev); TZa = torch.randn(max(K - 1, 1), N, Ca, device=dev)
    Zb = torch.randn(N, Cab, device=dev) if Cab else None; TZb = torch.randn(max(K - 1, 1), N, Cab, device=dev) if Cab else None
    S = torch.zeros(N, 4, device=dev) if ks else None
    rows = K * (Ca + Cab) + ks
    Cb = Cb if Cb is not None else Co
    W = torch.randn(rows, Kb * (Cb + Cbb), device=dev)
    out = torch.empty(Kb, N, Cb, device=dev); outb = torch.empty(Kb, N, Cbb, device=dev) if Cbb else None
    res = torch.randn(N, 4, device=dev)
    fn = lambda: _lib.call('qt_dense2', ptr(Za), 0, ptr(TZa), ptr(Zb), 0, ptr(TZb), K, Ca, Cab, ptr(W), None, ptr(S), ks,
                           ptr(W[K * (Ca + Cab):]) if ks else None, Kb, Cb, Cbb, N, None, act, ptr(res), 4, None, ptr(out), ptr(outb), 0, None, None)
    print(f'{name:34s} {timeit(fn):7.2f} us')
    return out
shape('fc1 fwd  (N x 64)(64 x 16) relu', 3, 16, 4, 16, act=1)
shape('fc2 fwd  (N x 52)(52 x 4) tanh', 3, 16, 0, 4, act=2)
shape('fc1 bwd  (N x 16)(16 x 3*20)', 1, 16, 0, None, Kb=3, Cb=16, Cbb=4, ks=0)
shape('fc2 bwd  (N x 4)(4 x 3*16)', 1, 4, 0, None, Kb=3, Cb=16, Cbb=0, ks=0)
