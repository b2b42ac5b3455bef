"""Phase timeline of k_gemm_fwd from in-kernel wall_clock64 stamps (diagnostics build: tools/micro/libqt_timing.so,
built with -DQT_GEMM_TIMING)."""
import ctypes, os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import torch
from qtmpnn import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'micro', 'libqt_timing.so')
from qtmpnn._lib import ptr
dev = torch.device('cuda', 0)
N = 120014
lib = _lib.load()
lib.qt_gemm_timing_buffer.argtypes = [ctypes.c_void_p]; lib.qt_gemm_timing_buffer.restype = None
nblk = (N + 127) // 128
names = ['start', 'table sync', 'W staged', 'mfma done', 'C staged', 'end']
def run(label, fn, nb=nblk):
    buf = torch.zeros(nb * 2, 8, dtype=torch.int64, device=dev)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    lib.qt_gemm_timing_buffer(buf.data_ptr())
    fn(); torch.cuda.synchronize()
    lib.qt_gemm_timing_buffer(None)
    t = buf[:nb, :6].cpu().double()
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    t = (t - t0) / 100.0          # 100 MHz counter -> us
    print(f'{label}: blocks {t.shape[0]}  kernel span {float(t[:, 5].max()):.2f} us')
    for i, nm in enumerate(names):
        c = t[:, i]
        print(f'   {nm:10s} median {float(c.median()):6.2f}  min {float(c.min()):6.2f}  max {float(c.max()):6.2f} us')
K, C, Co = 5, 20, 64
Z = torch.randn(N, C, device=dev); TZ = torch.randn(K - 1, N, C, device=dev)
S = torch.zeros(N, 4, device=dev); S[:, 0] = 1
W = torch.randn(K * C + 4, Co, device=dev); Y = torch.empty(N, Co, device=dev)
run('gate GEMM (N x 104)(104 x 64)', lambda: _lib.call('qt_dense', ptr(Z), ptr(TZ), K, C, ptr(W), ptr(S), 4, ptr(W[K * C:]), 1, Co, N, None, 0, None, 0, None, ptr(Y)))
G = torch.randn(N, Co, device=dev); Wt = torch.randn(Co, K * C, device=dev); gT = torch.empty(K, N, C, device=dev)
run('bwd-data GEMM (N x 64)(64 x 100)', lambda: _lib.call('qt_dense', ptr(G), None, 1, Co, ptr(Wt), None, 0, None, K, C, N, None, 0, None, 0, None, ptr(gT)))
