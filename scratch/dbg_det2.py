import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
from model.mpnnlstm import NextFramePredictorS2S
from qtmpnn import synthetic
dev = torch.device('cuda', 0)
x, y = synthetic.make_batch(1, 0, 3, 4, 4, n_digits=1, pixel_noise=0.05)
t = lambda a: torch.from_numpy(a).to(dev)
mask = np.zeros((64, 64), dtype=bool)
concat = torch.zeros(3, 4, 64, 64, 1, device=dev)
def fresh(cap):
    torch.manual_seed(3)
    nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=4, output_timesteps=4, device=dev,
                                model_kwargs=dict(hidden_size=16, dropout=0.0, n_layers=2))
    nfp.initiate_training(lr=1e-3, lr_decay=0.95, capturable=cap)
    return nfp
def pdiff(a, b):
    worst, wk = 0, None
    for (k, p), (_, q) in zip(a.model.named_parameters(), b.model.named_parameters()):
        d = (p - q).abs().max().item()
        if d > worst: worst, wk = d, k
    return worst, wk
def gdiff(a, b):
    worst, wk = 0, None
    for (k, p), (_, q) in zip(a.model.named_parameters(), b.model.named_parameters()):
        if p.grad is None: continue
        d = (p.grad - q.grad).abs().max().item()
        if d > worst: worst, wk = d, k
    return worst, wk
for cap in (False, True):
    A, B = fresh(cap), fresh(cap)
    for step in range(3):
        la, lb = A.train_step(t(x), t(y), concat, mask), B.train_step(t(x), t(y), concat, mask)
        print('capturable' if cap else 'plain', 'step', step, 'loss', float(la), float(lb), 'grad diff', gdiff(A, B), 'param diff', pdiff(A, B))
