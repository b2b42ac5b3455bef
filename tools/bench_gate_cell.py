"""Time qt_dense_lstm (gate GEMM + LSTM cell) and qt_lstm_bwd_dgrad (cell backward + data gradient) at the bench shapes,
each launch replayed from a hipGraph (diagnostics; QT_GATE_CELL_TILED=1 / QT_DGRAD_TILED=1 select the one-tile-per-workgroup
kernels of round 1 -- the choice is read once per process, so A/B needs two runs).  Prints a checksum of the outputs."""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import torch
from qtmpnn import _lib
if os.environ.get('QT_LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'micro', os.environ['QT_LIB'])
from qtmpnn._lib import ptr
dev = torch.device('cuda', 0)
torch.manual_seed(0)
N, h = int(os.environ.get('QT_N', 120014)), int(os.environ.get('QT_H', 16))
CAP = 131072
nvalid = torch.tensor([N], dtype=torch.int32, device=dev)
print('CUs', torch.cuda.get_device_properties(0).multi_processor_count, 'N', N)


def graph_time(fn, reps=20):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def shape(name, K, Ca, Cab):
    X = torch.randn(CAP, Ca, device=dev)
    Hh = torch.randn(CAP, Cab, device=dev) if Cab else None
    TX = torch.randn(K - 1, CAP, Ca, device=dev)
    TH = torch.randn(K - 1, CAP, Cab, device=dev) if Cab else None
    S = torch.zeros(CAP, 4, device=dev); S[:, 0] = 1
    Kt = K * (Ca + Cab) + 4
    W = 0.1 * torch.randn(Kt, 4 * h, device=dev)
    WT = W.t().contiguous()
    Cp = torch.randn(CAP, h, device=dev)
    wc, b, ln = 0.1 * torch.randn(3, h, device=dev), 0.1 * torch.randn(4, h, device=dev), torch.randn(4, h, device=dev)
    Hn, Cn, gates = (torch.zeros(CAP, w, device=dev) for w in (h, h, 4 * h))
    fwd = lambda: _lib.call('qt_dense_lstm', ptr(X), Ca, ptr(TX), ptr(Hh), Cab, ptr(TH), K, Ca, Cab, ptr(W), ptr(WT), ptr(S), 4,
                            ptr(W[K * (Ca + Cab):]), h, CAP, ptr(nvalid), ptr(Cp), h, ptr(wc), ptr(b), ptr(ln), None, ptr(Hn), ptr(Cn), ptr(gates),
                            int(os.environ.get('QT_PLANES_SM', '0')))
    us = graph_time(fwd)
    flops = 2.0 * N * Kt * 4 * h
    byts = 4.0 * N * (Kt + h + 6 * h)
    print(f'{name}: fwd gate GEMM + cell K={Kt}: {us:7.2f} us  {flops / us / 1e6:6.1f} TFLOP/s  {byts / us / 1e3:7.0f} GB/s   '
          f'checksum {float(Hn[:N].double().sum()):.6f} {float(Cn[:N].double().sum()):.6f} {float(gates[:N].double().sum()):.6f}')
    # backward: cell backward + data gradient into the planes of the parts that want a gradient (H only for the encoder's
    # layer 0, everything for the decoder)
    live = [Cab] if (Cab and name.startswith('enc')) else ([Ca, Cab] if Cab else [Ca])
    gO, gH, gC = (torch.randn(CAP, h, device=dev) for _ in range(3))
    gG, gCp = torch.zeros(CAP, 4 * h, device=dev), torch.zeros(CAP, h, device=dev)
    nblk = max(_lib.value('qt_lstm_bwd_blocks', CAP, h), _lib.value('qt_lstm_dgrad_blocks', CAP), 1)
    part = torch.zeros(nblk, 11 * h, device=dev)
    if len(live) == 2:
        Wb = W[:K * (Ca + Cab)]
    else:
        lo = 0 if not Cab or live[0] == Ca and not name.startswith('enc') else Ca
        Wb = W[:K * (Ca + Cab)].view(K, Ca + Cab, 4 * h)[:, lo:lo + live[0]].reshape(-1, 4 * h).contiguous()
    planes = [torch.zeros(K, CAP, c, device=dev) for c in live]
    Wb = Wb.contiguous()
    whi, wlo = (torch.empty(Wb.shape, dtype=torch.bfloat16, device=dev) for _ in range(2))
    _lib.call('qt_split_bf16', ptr(Wb), Wb.numel(), ptr(whi), ptr(wlo))
    bwd = lambda: _lib.call('qt_lstm_bwd_dgrad', ptr(gO), h, ptr(gH), h, ptr(gC), h, ptr(gates), ptr(Cp), h, ptr(wc), ptr(ln), CAP,
                            ptr(nvalid), h, ptr(gG), ptr(gCp), ptr(part), 1, ptr(Wb), ptr(whi), ptr(wlo), K, live[0], live[1] if len(live) > 1 else 0,
                            ptr(planes[0]), ptr(planes[1]) if len(live) > 1 else None, int(os.environ.get('QT_PLANES_SM', '0')),
                            None, 0, None)
    part.zero_()
    us = graph_time(bwd)
    NB = K * sum(live)
    if os.environ.get('QT_FUSED', '0') == '1':
        ncu = _lib.value('qt_lstm_fused_blocks')
        slab = torch.zeros(ncu, Kt, 4 * h, device=dev)
        Zs, TZs = ([X, Hh], [TX, TH]) if Cab else ([X], [TX])
        fus = lambda: _lib.call('qt_lstm_bwd_fused', ptr(gO), h, ptr(gH), h, ptr(gC), h, ptr(gates), ptr(Cp), h, ptr(wc), ptr(ln), CAP,
                                ptr(nvalid), h, ptr(gCp), ptr(part), 1, ptr(Wb), K, live[0], live[1] if len(live) > 1 else 0,
                                ptr(planes[0]), ptr(planes[1]) if len(live) > 1 else None,
                                ptr(Zs[0]), Ca, ptr(TZs[0]), ptr(Zs[1]) if Cab else None, Cab, ptr(TZs[1]) if Cab else None,
                                K, Ca, Cab, ptr(S), 4, ptr(slab), ncu)
        usf = graph_time(fus)
        print(f'{name}: FUSED bwd cell + dgrad + wgrad: {usf:7.2f} us   slab checksum {float(slab.double().sum()):.4f}')
    byts = 4.0 * N * (3 * h + 4 * h + h + 4 * h + h + NB)
    print(f'{name}: bwd cell + dgrad NB={NB}: {us:7.2f} us  {2.0 * N * 4 * h * NB / us / 1e6:6.1f} TFLOP/s  {byts / us / 1e3:7.0f} GB/s   '
          f'checksum {float(gG[:N].double().sum()):.6f} {float(gCp[:N].double().sum()):.6f} {sum(float(p[:, :N].double().sum()) for p in planes):.6f}')


shape('enc layer0 [X4|H16] K=5', 5, 4, 16)
shape('enc layer1 [H16]    K=5', 5, 16, 0)
shape('dec layer0 [X4|H16] K=3', 3, 4, 16)
shape('dec layer1 [H16|H16] K=3', 3, 16, 16)
