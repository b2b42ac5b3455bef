import faulthandler; faulthandler.enable()
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
from model.mpnnlstm import NextFramePredictorS2S
from qtmpnn import synthetic
dev = torch.device('cuda', 0)
x, y = synthetic.make_batch(1, 0, 3, 4, 4, n_digits=1, pixel_noise=0.05)
x2, y2 = synthetic.make_batch(1, 50, 3, 4, 4, n_digits=1, pixel_noise=0.05)
t = lambda a: torch.from_numpy(a).to(dev)
mask = np.zeros((64, 64), dtype=bool)
concat = torch.zeros(3, 4, 64, 64, 1, device=dev)
def fresh(capturable, static=False):
    torch.manual_seed(3)
    nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=4, output_timesteps=4, device=dev,
                                model_kwargs=dict(hidden_size=16, dropout=0.0, n_layers=2))
    nfp.initiate_training(lr=float(os.environ.get("LR", "0.001")), lr_decay=0.95, capturable=capturable)
    nfp.model.static_shapes = static
    return nfp
# ---- stage A: graphed fwd+bwd grads vs eager grads (same weights, no optimizer)
def grads_eager(n, a, b):
    for p in n.model.parameters(): p.grad = None
    l = n.forward_loss(a, b, concat, mask); l.backward()
    return float(l), {k: p.grad.clone() for k, p in n.model.named_parameters() if p.grad is not None}
n = fresh(True, True)
le, ge = grads_eager(n, t(x2), t(y2))
sx, sy = t(x).clone(), t(y).clone()
def fb():
    for p in n.model.parameters(): p.grad = None
    l = n.forward_loss(sx, sy, concat, mask); l.backward(); return l.detach()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fb()
torch.cuda.current_stream().wait_stream(side)
gr = torch.cuda.CUDAGraph()
for p in n.model.parameters(): p.grad = None
with torch.cuda.graph(gr, stream=side):
    sl = fb()
sx.copy_(t(x2)); sy.copy_(t(y2)); gr.replay(); torch.cuda.synchronize()
worst = 0
for k, p in n.model.named_parameters():
    if p.grad is None: continue
    d = (p.grad - ge[k]).abs().max().item(); s = ge[k].abs().max().item()
    worst = max(worst, d / (s + 1e-12))
    if d > 1e-6 * (s + 1e-9): print('GRAD MISMATCH', k, d, s)
print('stage A loss', float(sl), le, 'worst rel grad diff', worst)
gr.replay(); torch.cuda.synchronize()
worst = 0
for k, p in n.model.named_parameters():
    if p.grad is None: continue
    d = (p.grad - ge[k]).abs().max().item(); s = ge[k].abs().max().item()
    worst = max(worst, d / (s + 1e-12))
print('stage A second replay worst rel grad diff', worst)
