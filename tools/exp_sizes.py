"""Node-size histogram of the rollout meshes of the bench model after a few training steps (diagnostics)."""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
import bench
from qtmpnn import synthetic
dev = torch.device('cuda', 0)
nfp = bench.make_predictor(dev, capturable=False)
nfp.model.train()
noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
x, y = synthetic.make_batch(2, 0, 32, 10, 10, n_digits=2, pixel_noise=noise)
b = (torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), torch.zeros(32, 10, 64, 64, 1, device=dev))
mask = np.zeros((64, 64), dtype=bool)
for it in range(8):
    nfp.train_step(*b, mask)
outs, meshes = nfp.model(b[0], b[1], b[2], teacher_forcing_ratio=0, mask=mask)
for t, ms in enumerate(meshes):
    sz = ms.cell[:ms.N, 2].long()
    h = torch.bincount(sz, minlength=65)
    print('step', t, 'N', ms.N, {s: int(h[s]) for s in (1, 2, 4, 8, 16, 32, 64) if int(h[s])})
