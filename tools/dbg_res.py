import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from qtmpnn import ops
orig = ops._ChebPoly.forward
def fwd(ctx, Z, W, res, drop, mesh, K, Ks, act, acc):
    if res is not None:
        print('res', tuple(res.shape), res.stride(), res.is_contiguous(), type(res.grad_fn).__name__ if res.grad_fn else None)
    return orig(ctx, Z, W, res, drop, mesh, K, Ks, act, acc)
ops._ChebPoly.forward = staticmethod(fwd)
import test_gpu_rollout as T
from helpers import golden
g = golden('rollout_mnist64_noise_h8.npz')
model, outs, meshes, loss = T._run(g)
