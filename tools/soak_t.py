"""Soak of the TransformerConv configuration (cfg4t shapes): many hipGraph replays on cycling batches -- finite, decreasing loss,
flat memory (diagnostics)."""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
from model.mpnnlstm import NextFramePredictorS2S
from qtmpnn import synthetic
dev = torch.device('cuda', 0)
B, t_in, t_out, shape = 16, 12, 6, (128, 128)
torch.manual_seed(1)
nfp = NextFramePredictorS2S(thresh=0.15, input_features=5, input_timesteps=t_in, output_timesteps=t_out, device=dev,
                            transform_func=lambda a: abs(abs(a - 0.5) - 0.5),
                            model_kwargs=dict(hidden_size=32, dropout=0.1, n_layers=1, n_conv_layers=3, convolution_type='TransformerConv',
                                              transform_func=lambda a: abs(abs(a - 0.5) - 0.5)))
nfp.initiate_training(lr=0.003, lr_decay=0.95, capturable=True)
nfp.model.train()
mask = synthetic.make_ice_like(40, shape=shape, channels=5, n_frames=2)[1]
pool = []
for i in range(4):
    clips = [synthetic.make_ice_like(1000 * i + k, shape=shape, channels=5, n_frames=t_in + t_out)[0] for k in range(B)]
    x = np.stack([c[:t_in] for c in clips]); y = np.stack([c[t_in:, ..., :1] for c in clips])
    pool.append((torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), torch.zeros(B, t_out, *shape, 1, device=dev)))
step = nfp.make_graphed_step(*pool[0], mask=mask, warmup=2)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
losses, m0, t0 = [], torch.cuda.memory_allocated(), time.time()
for i in range(n):
    losses.append(step(*pool[i % 4]).clone())
torch.cuda.synchronize()
dt = time.time() - t0
L = torch.stack(losses).cpu().numpy()
print('steps', n, 'ms/step', round(dt / n * 1e3, 2), 'finite', bool(np.isfinite(L).all()), 'loss first10', L[:10].mean().round(4), 'last10',
      L[-10:].mean().round(4), 'mem delta MB', round((torch.cuda.memory_allocated() - m0) / 1e6, 2), 'max alloc GB',
      round(torch.cuda.max_memory_allocated() / 1e9, 2))
