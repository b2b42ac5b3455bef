"""Soak of the tile-resident path: many hipGraph replays of the configs[2] training step (8 clips of 128x128, in=10/out=20,
hidden 16, stacks of two ChebConvs: the encoder's recurrences run as tile-resident launches) on cycling batches -- finite,
decreasing loss, flat memory, and the error word of the tile launches stays 0.    python tools/soak_cfg3.py [steps]"""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
from model.mpnnlstm import NextFramePredictorS2S
from qtmpnn import synthetic
from qtmpnn.mesh import tile_error_word
dev = torch.device('cuda', 0)
torch.manual_seed(1)
B, t_in, t_out, shape = 8, 10, 20, (128, 128)
nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=t_in, output_timesteps=t_out, device=dev,
                            model_kwargs=dict(hidden_size=16, dropout=0.1, n_layers=2))
nfp.initiate_training(lr=0.01, lr_decay=0.95, capturable=True)
nfp.model.train()
mask = np.zeros(shape, dtype=bool)
pool = []
for i in range(4):
    x, y = synthetic.make_batch(3, i * B, B, t_in, t_out, n_digits=2, pixel_noise=0.05, canvas=shape)
    pool.append((torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), torch.zeros(B, t_out, *shape, 1, device=dev)))
step = nfp.make_graphed_step(*pool[0], mask=mask, warmup=2)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
losses = []
m0 = torch.cuda.memory_allocated()
t0 = time.time()
for i in range(n):
    losses.append(step(*pool[i % 4]).clone())
torch.cuda.synchronize()
dt = time.time() - t0
L = torch.stack(losses).cpu().numpy()
print('steps', n, 'ms/step', round(dt / n * 1e3, 3), 'finite', bool(np.isfinite(L).all()), 'loss first10', L[:10].mean().round(4), 'last10', L[-10:].mean().round(4),
      'mem delta MB', round((torch.cuda.memory_allocated() - m0) / 1e6, 2), 'max alloc GB', round(torch.cuda.max_memory_allocated() / 1e9, 2),
      'tile error word', tile_error_word())
