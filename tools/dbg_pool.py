import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import torch, numpy as np
from qtmpnn import synthetic, ops
from qtmpnn.mesh import build_mesh
dev = torch.device('cuda', 0)
def mesh(seed, noise, B=1):
    x, _ = synthetic.make_batch(2, seed, B, 3, 1, n_digits=2, pixel_noise=noise)
    return build_mesh(src=torch.from_numpy(x[..., 0]).to(dev).amax(dim=1), thresh=0.1)
for C in (8, 4, 68):
    old, new = mesh(11, 0.0), mesh(12, 0.03)
    val = torch.randn(old.N, C, device=dev)
    out = ops.remesh_transfer(val, old, new)
    img = val[old.labels.view(-1).long()]            # (P, C)
    ref = torch.zeros(new.N, C, device=dev).index_add_(0, new.labels.view(-1).long(), img) / new.npix.view(-1, 1)
    bad = ((out - ref).abs() > 1e-5)
    rows = bad.any(1).nonzero().view(-1)
    lvl = torch.zeros(new.N, dtype=torch.long, device=dev); lvl[new.labels.view(-1).long()] = new.level.view(-1).long()
    print('C', C, 'N', new.N, 'bad rows', rows.numel(), 'levels of bad rows', torch.bincount(lvl[rows], minlength=7).tolist(),
          'bad per col', bad.sum(0).tolist()[:8], 'level hist', torch.bincount(lvl, minlength=7).tolist())
