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
def fresh():
    torch.manual_seed(3)
    return NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=4, output_timesteps=4, device=dev,
                                 model_kwargs=dict(hidden_size=16, dropout=0.0, n_layers=2))
def grads(n):
    for p in n.model.parameters(): p.grad = None
    l = n.forward_loss(t(x), t(y), concat, mask); l.backward()
    return {k: (None if p.grad is None else p.grad.clone()) for k, p in n.model.named_parameters()}
A, B = fresh(), fresh()
print('init equal', all(torch.equal(p, q) for p, q in zip(A.model.parameters(), B.model.parameters())))
gA, gB, gA2 = grads(A), grads(B), grads(A)
for name, (u, v) in (('A vs B', (gA, gB)), ('A vs A again', (gA, gA2))):
    bad = []
    for k in u:
        if (u[k] is None) != (v[k] is None): bad.append((k, 'None mismatch')); continue
        if u[k] is None: continue
        if not torch.equal(u[k], v[k]): bad.append((k, (u[k] - v[k]).abs().max().item(), u[k].abs().max().item()))
    print(name, 'differing', len(bad)); [print('   ', b) for b in bad[:12]]
