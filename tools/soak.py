"""Soak: many hipGraph replays of the training step on cycling batches -- finite, decreasing loss and flat memory (diagnostics)."""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
import bench
from qtmpnn import synthetic
dev = torch.device('cuda', 0)
nfp = bench.make_predictor(dev, capturable=True)
nfp.model.train()
mask = np.zeros((64, 64), dtype=bool)
pool = []
for i in range(8):
    x, y = synthetic.make_batch(2, i * 32, 32, 10, 10, n_digits=2, pixel_noise=0.05)
    pool.append((torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), torch.zeros(32, 10, 64, 64, 1, device=dev)))
step = nfp.make_graphed_step(*pool[0], mask=mask, warmup=2)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
losses = []
m0 = torch.cuda.memory_allocated()
t0 = time.time()
for i in range(n):
    losses.append(step(*pool[i % 8]).clone())
torch.cuda.synchronize()
dt = time.time() - t0
L = torch.stack(losses).cpu().numpy()
print('steps', n, 'ms/step', round(dt / n * 1e3, 3), 'finite', bool(np.isfinite(L).all()), 'loss first10', L[:10].mean().round(4), 'last10', L[-10:].mean().round(4),
      'mem delta MB', round((torch.cuda.memory_allocated() - m0) / 1e6, 2), 'max alloc GB', round(torch.cuda.max_memory_allocated() / 1e9, 2))
