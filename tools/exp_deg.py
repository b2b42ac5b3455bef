import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import torch
from qtmpnn import synthetic
from qtmpnn.mesh import build_mesh
dev = torch.device('cuda', 0)
for noise in (0.05, 0.0):
    x, _ = synthetic.make_batch(2, 0, 32, 10, 1, n_digits=2, pixel_noise=noise)
    mesh = build_mesh(src=torch.from_numpy(x[..., 0]).to(dev).amax(dim=1), thresh=0.1)
    deg = (mesh.rowptr[1:] - mesh.rowptr[:-1]).long()
    print('noise', noise, 'N', mesh.N, 'deg hist', torch.bincount(torch.clamp(deg, max=40)).tolist(), 'max', int(deg.max()))
    print(' sizes', torch.bincount(mesh.cell[:, 2].long()).nonzero().view(-1).tolist(), torch.bincount(mesh.cell[:, 2].long())[torch.bincount(mesh.cell[:, 2].long()).nonzero().view(-1)].tolist())
