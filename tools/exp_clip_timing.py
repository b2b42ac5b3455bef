"""Phase timeline of k_cheb_clip from in-kernel wall_clock64 stamps (diagnostics build: `make -C quadtree-mpnnlstm_amd/csrc timing`,
-DQT_CLIP_TIMING): start | prologue loads issued | ELL unpacked, first plane written | barrier passed | hop 1 | hop 2 | ..."""
import ctypes, os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import torch
from qtmpnn import _lib
_lib.LIB_PATH = os.environ.get('QT_CLIP_LIB') or os.path.join(ROOT, 'tools', 'micro', 'libqt_clip_timing.so')
from qtmpnn import ops, synthetic
from qtmpnn.mesh import build_mesh
dev = torch.device('cuda', 0)
lib = _lib.load()
lib.qt_clip_timing_buffer.argtypes = [ctypes.c_void_p]; lib.qt_clip_timing_buffer.restype = None
noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
x, _ = synthetic.make_batch(2, 0, 32, 10, 10, n_digits=2, pixel_noise=noise, canvas=(64, 64))
mesh = build_mesh(src=torch.from_numpy(x[..., 0].max(axis=1)).to(dev), thresh=0.1, static=True)
print(f'mesh: N = {mesh.n_valid}, E = {mesh.E}')
for K, widths, bwd in [(5, (4, 16), False), (5, (4, 16), True), (5, (16,), False), (3, (16,), False)]:
    N = mesh.N
    Zs = [torch.randn(N, w, device=dev) for w in widths]
    TZ = [torch.empty(K - 1, N, w, device=dev) for w in widths]
    G = [torch.randn(K, N, w, device=dev) for w in widths]
    fn = (lambda: ops.clip_clenshaw(mesh, G, K)) if bwd else (lambda: ops.clip_planes(mesh, Zs, TZ, K))
    nwg = 32 * sum(widths) // 4
    buf = torch.zeros(nwg, 16, dtype=torch.int64, device=dev)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.qt_clip_timing_buffer(buf.data_ptr())
    fn(); torch.cuda.synchronize()
    lib.qt_clip_timing_buffer(None)
    t = buf.cpu().double()
    n = 4 + K - 1
    t0 = t[:, 0].min()
    t = (t[:, :n] - t0) / 100.0
    print(f'K={K} widths={widths} {"bwd" if bwd else "fwd"}: {nwg} workgroups, span {float(t[:, n - 1].max()):.2f} us')
    names = ['start', 'loads issued', 'unpacked', 'barrier'] + [f'hop {i + 1}' for i in range(K - 1)]
    for i, nm in enumerate(names):
        c = t[:, i]
        print(f'   {nm:13s} median {float(c.median()):6.2f}  min {float(c.min()):6.2f}  max {float(c.max()):6.2f} us')
