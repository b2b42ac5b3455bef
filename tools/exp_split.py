import sys, os, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
import bench
from qtmpnn import synthetic
dev = torch.device('cuda', 0)
nfp = bench.make_predictor(dev, capturable=True)
nfp.model.train(); nfp.model.static_shapes = True
mask = np.zeros((64, 64), dtype=bool)
x, y = synthetic.make_batch(2, 0, 32, 10, 10, n_digits=2, pixel_noise=0.05)
x, y = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
c = torch.zeros(32, 10, 64, 64, 1, device=dev)
def timeit(fn, n=10):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def fwd():
    with torch.no_grad():
        return nfp.forward_loss(x, y, c, mask)
def enc_only():
    with torch.no_grad():
        nfp.model.process_inputs(x, mask=mask)
def fwdbwd():
    for p in nfp.model.parameters(): p.grad = None
    nfp.forward_loss(x, y, c, mask).backward()
print('encoder fwd (10 steps + first mesh) ms', round(timeit(enc_only), 3))
print('forward (no grad) ms', round(timeit(fwd), 3))
print('forward+backward ms', round(timeit(fwdbwd), 3))
