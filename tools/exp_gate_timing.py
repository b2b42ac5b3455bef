"""Phase timeline of k_gate_cell_p (persistent gate GEMM + cell) from in-kernel wall_clock64 stamps of wave 0 of every
workgroup (diagnostics build tools/micro/libqt_timing.so: make -C quadtree-mpnnlstm_amd/csrc timing)."""
import ctypes, os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import torch
from qtmpnn import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'micro', os.environ.get('QT_TIMING_LIB', 'libqt_timing.so'))
from qtmpnn._lib import ptr
dev = torch.device('cuda', 0)
N, h, CAP = int(os.environ.get('QT_N', 120014)), 16, 131072
lib = _lib.load()
lib.qt_gemm_timing_buffer.argtypes = [ctypes.c_void_p]; lib.qt_gemm_timing_buffer.restype = None
nvalid = torch.tensor([N], dtype=torch.int32, device=dev)
K, Ca, Cab = 5, 4, 16
X, Hh = torch.randn(CAP, Ca, device=dev), torch.randn(CAP, Cab, device=dev)
TX, TH = torch.randn(K - 1, CAP, Ca, device=dev), torch.randn(K - 1, CAP, Cab, device=dev)
S = torch.zeros(CAP, 4, device=dev); S[:, 0] = 1
Kt = K * (Ca + Cab) + 4
W = 0.1 * torch.randn(Kt, 4 * h, device=dev); WT = W.t().contiguous()
Cp = torch.randn(CAP, h, device=dev)
wc, b, ln = 0.1 * torch.randn(3, h, device=dev), 0.1 * torch.randn(4, h, device=dev), torch.randn(4, h, device=dev)
Hn, Cn, gates = (torch.zeros(CAP, w, device=dev) for w in (h, h, 4 * h))
fn = lambda: _lib.call('qt_dense_lstm', ptr(X), Ca, ptr(TX), ptr(Hh), Cab, ptr(TH), K, Ca, Cab, ptr(W), ptr(WT), ptr(S), 4,
                       ptr(W[K * (Ca + Cab):]), h, CAP, ptr(nvalid), ptr(Cp), h, ptr(wc), ptr(b), ptr(ln), None, ptr(Hn), ptr(Cn), ptr(gates))
nb = 1024
buf = torch.zeros(nb * 2, 8, dtype=torch.int64, device=dev)
for _ in range(5): fn()
torch.cuda.synchronize()
lib.qt_gemm_timing_buffer(buf.data_ptr())
fn(); torch.cuda.synchronize()
lib.qt_gemm_timing_buffer(None)
t = buf[:nb, :5].cpu().double()
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
t = (t - t0) / 100.0
names = ['start', 'W staged', 'unit 0 mfma done', 'unit 0 epilogue done', 'wave 0 end']
print(f'N {N}: workgroups {t.shape[0]}, span {float(t[:, 4].max()):.2f} us')
for i, nm in enumerate(names):
    c = t[:, i]
    print(f'   {nm:22s} median {float(c.median()):6.2f}  min {float(c.min()):6.2f}  max {float(c.max()):6.2f} us')
