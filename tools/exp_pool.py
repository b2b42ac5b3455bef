"""remesh-transfer timing at the bench shapes (diagnostics)."""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import torch
from qtmpnn import synthetic, ops
from qtmpnn.mesh import build_mesh
dev = torch.device('cuda', 0)
def mesh(seed, noise, B=32):
    x, _ = synthetic.make_batch(2, seed, B, 3, 1, n_digits=2, pixel_noise=noise)
    return build_mesh(src=torch.from_numpy(x[..., 0]).to(dev).amax(dim=1), thresh=0.1)
def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
for noise in (0.05, 0.0):
    old, new = mesh(100, noise), mesh(200, noise)
    for C in (68, 36, 4):
        val = torch.randn(old.N, C, device=dev); out = torch.empty(new.N, C, device=dev)
        t = timeit(lambda: ops._pool_raw(new, C, out, C, 0, True, src_val=val, src_mesh=old))
        print(f'noise {noise} N {old.N}->{new.N} C {C}: {t:.2f} us  ({(old.N + new.N) * C * 4 / t / 1e3:.0f} GB/s rows)')
