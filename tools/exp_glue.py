"""Which host lines launch the torch glue kernels of one training step (diagnostics)."""
import os, sys, collections
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
import bench
from qtmpnn import synthetic
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda', 0)
nfp = bench.make_predictor(dev, capturable=True)
nfp.model.train(); nfp.model.static_shapes = True
x, y = synthetic.make_batch(2, 0, 32, 10, 10, n_digits=2, pixel_noise=0.05)
b = (torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), torch.zeros(32, 10, 64, 64, 1, device=dev))
mask = np.zeros((64, 64), dtype=bool)
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
