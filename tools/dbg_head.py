import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import torch
from qtmpnn import ops, _lib
from qtmpnn.mesh import Mesh
dev = torch.device('cuda', 0)
for h in (8, 16):
    O = torch.randn(100, h, device=dev); ln = torch.randn(2, h, device=dev); cc = torch.randn(100, 1, device=dev)
    try:
        z = ops.head_input(O, ln, cc, h + 4, Mesh())
        print(h, 'ok', z[0].shape, z[1].shape)
    except Exception as e:
        print(h, 'ERR', e)
print(_lib._SIGNATURES['qt_head_fwd'])
