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
for static in (False, True):
    torch.manual_seed(3)
    nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=4, output_timesteps=4, device=dev,
                                model_kwargs=dict(hidden_size=16, dropout=0.0, n_layers=2))
    nfp.model.static_shapes = static
    res = []
    for it in range(4):
        for p in nfp.model.parameters(): p.grad = None
        # poison the allocator's free memory so that uninitialised reads differ between iterations
        junk = torch.full((64 * 1024 * 1024,), float(it + 1) * 1e3, device=dev); del junk
        l = nfp.forward_loss(t(x), t(y), concat, mask); l.backward()
        res.append((l.detach().clone(), {k: p.grad.clone() for k, p in nfp.model.named_parameters() if p.grad is not None}))
    for it in range(1, 4):
        bad = [k for k in res[0][1] if not torch.equal(res[0][1][k], res[it][1][k])]
        print('static' if static else 'dynamic', 'iter', it, 'loss equal', torch.equal(res[0][0], res[it][0]), 'params differing', len(bad), bad[:6])
        for k in bad[:3]:
            d = (res[0][1][k] - res[it][1][k]).abs().max().item()
            print('   ', k, 'max abs diff', d, 'scale', res[0][1][k].abs().max().item())
