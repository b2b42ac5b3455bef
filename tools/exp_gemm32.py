"""The hidden-32 gate GEMMs (BASELINE configs[3] / [4]: K = 7 planes of 8 + 32 channels, 128 gate columns) in isolation:
forward qt_dense_lstm (k_gemm_fwd<4, 64, 8>: GEMM + LSTM cell) and the data gradient qt_dense2 (N x 128)(128 x 280), graph-replayed,
as TFLOP/s of the fp32 MFMA peak (157.3).    python tools/exp_gemm32.py [N]     (QT_LIB=<name>.so under tools/micro: another build)"""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import torch
from qtmpnn import _lib
if os.environ.get('QT_LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'micro', os.environ['QT_LIB'])
from qtmpnn._lib import ptr
dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 126192
h, K, Ca, Cab = 32, 7, 8, 32
Kt = K * (Ca + Cab) + 4
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)


def timeit(fn, reps=20):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        for _ in range(reps):
            fn()
    gr.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); gr.replay(); b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


X, H, TX, TH = rnd(N, Ca), rnd(N, Cab), rnd(K - 1, N, Ca), rnd(K - 1, N, Cab)
S = torch.zeros(N, 4, device=dev); S[:, 0] = 1
W = 0.1 * rnd(Kt, 4 * h)
WT = W.t().contiguous()
Cp, wc, b, ln = rnd(N, h), 0.1 * rnd(3, h), 0.1 * rnd(4, h), rnd(4, h)
Hn, Cn, gates = (torch.empty(N, w, device=dev) for w in (h, h, 4 * h))
fwd = lambda: _lib.call('qt_dense_lstm', ptr(X), Ca, ptr(TX), ptr(H), Cab, ptr(TH), K, Ca, Cab, ptr(W), ptr(WT), ptr(S), 4,
                        ptr(W[K * (Ca + Cab):]), h, N, None, ptr(Cp), h, ptr(wc), ptr(b), ptr(ln), None, ptr(Hn), ptr(Cn), ptr(gates), 0)
us = timeit(fwd)
print(f'forward gate GEMM + cell ({N} x {Kt})({Kt} x {4 * h}): {us:7.2f} us  {2.0 * N * Kt * 4 * h / us / 1e6:6.1f} TFLOP/s '
      f'({2.0 * N * Kt * 4 * h / us / 1e6 / 157.3:.2f} of the fp32 MFMA peak)')
G = rnd(N, 4 * h)
Wb = W[:K * (Ca + Cab)].contiguous()          # rows of the forward weight = the transposed operand of the data gradient
ga, gb = torch.empty(K, N, Ca, device=dev), torch.empty(K, N, Cab, device=dev)
bwd = lambda: _lib.call('qt_dense2', ptr(G), 0, None, None, 0, None, 1, 4 * h, 0, None, ptr(Wb), None, 0, None, K, Ca, Cab, N, None, 0,
                        None, 0, None, ptr(ga), ptr(gb), 0, None, None)
us = timeit(bwd)
fl = 2.0 * N * 4 * h * K * (Ca + Cab)
print(f'data gradient ({N} x {4 * h})({4 * h} x {K * (Ca + Cab)}): {us:7.2f} us  {fl / us / 1e6:6.1f} TFLOP/s ({fl / us / 1e6 / 157.3:.2f})')
