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
def fresh():
    torch.manual_seed(3)
    nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=4, output_timesteps=4, device=dev,
                                model_kwargs=dict(hidden_size=16, dropout=0.0, n_layers=2))
    nfp.initiate_training(lr=1e-3, lr_decay=0.95, capturable=True)
    nfp.model.static_shapes = True
    return nfp
def pdiff(a, b, tag):
    worst = 0; wk = None
    for (k, p), (_, q) in zip(a.model.named_parameters(), b.model.named_parameters()):
        d = (p - q).abs().max().item()
        if d > worst: worst, wk = d, k
    print(tag, 'max param diff', worst, wk)
A, A2, B = fresh(), fresh(), fresh()
pdiff(A, B, 'init')
for n in (A, A2):
    for _ in range(2): n.train_step(t(x), t(y), concat, mask)
pdiff(A, A2, 'two eager instances after 2 steps')
step = B.make_graphed_step(t(x), t(y), concat, mask, warmup=2)
pdiff(A, B, 'eager vs graphed-warmup after 2 steps')
la = float(A.train_step(t(x2), t(y2), concat, mask)); lb = float(step(t(x2), t(y2), concat))
print('loss step 3', la, lb)
pdiff(A, B, 'after step 3 (eager vs graph replay)')
la = float(A.train_step(t(x), t(y), concat, mask)); lb = float(step(t(x), t(y), concat))
print('loss step 4', la, lb)
pdiff(A, B, 'after step 4')
st = A.optimizer.state[next(iter(A.model.parameters()))]; sb = B.optimizer.state[next(iter(B.model.parameters()))]
print('adam step counters', st['step'].item(), sb['step'].item())
