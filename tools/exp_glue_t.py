"""Which host lines launch the torch glue kernels of one TransformerConv training step (cfg4t shapes, 4 clips; diagnostics)."""
import os, sys, collections
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
from model.mpnnlstm import NextFramePredictorS2S
from qtmpnn import synthetic
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda', 0)
B, t_in, t_out, shape = int(os.environ.get('QT_B', 4)), 12, 6, (128, 128)
torch.manual_seed(1)
nfp = NextFramePredictorS2S(thresh=0.15, input_features=5, input_timesteps=t_in, output_timesteps=t_out, device=dev,
                            transform_func=lambda a: abs(abs(a - 0.5) - 0.5),
                            model_kwargs=dict(hidden_size=32, dropout=0.1, n_layers=1, n_conv_layers=3, convolution_type=os.environ.get('QT_CONV', 'TransformerConv'),
                                              transform_func=lambda a: abs(abs(a - 0.5) - 0.5)))
nfp.initiate_training(lr=0.01, lr_decay=0.95, capturable=True)
nfp.model.train(); nfp.model.static_shapes = True
mask = synthetic.make_ice_like(40, shape=shape, channels=5, n_frames=2)[1]
clips = [synthetic.make_ice_like(1000 + k, shape=shape, channels=5, n_frames=t_in + t_out)[0] for k in range(B)]
x = np.stack([c[:t_in] for c in clips]); y = np.stack([c[t_in:, ..., :1] for c in clips])
b = (torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), torch.zeros(B, t_out, *shape, 1, device=dev))
for _ in range(2): nfp.train_step(*b, mask)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    nfp.train_step(*b, mask)
    torch.cuda.synchronize()
ev = prof.events()
# kernels by launching aten op + innermost repo frame
cnt = collections.Counter(); tim = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith('aten::') and e.kernels:
        # only leaf ops (kernels attached)
        chain, q = [], e.cpu_parent
        while q is not None:
            chain.append(q.name.replace('aten::', '').replace('autograd::engine::evaluate_function: ', 'bw:'))
            q = q.cpu_parent
        shapes = str([tuple(s) for s in (e.input_shapes or []) if s][:3])
        frame = ' < '.join(chain[:4]) + ' ' + shapes
        key = (e.name, frame.replace(ROOT, '')[:110])
        cnt[key] += len(e.kernels); tim[key] += sum(k.duration for k in e.kernels)
tot = sum(cnt.values())
print('glue kernels', tot, 'time us', sum(tim.values()))
for k, c in sorted(cnt.items(), key=lambda kv: -tim[kv[0]])[:60]:
    print(f'{c:5d} {tim[k]:8.0f} us  {k[0]:28s} {k[1]}')
