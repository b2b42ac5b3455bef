"""List the aten ops of one eager training step (diagnostics: where do the glue kernels come from?)."""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
import bench
from qtmpnn import synthetic
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda', 0)
nfp = bench.make_predictor(dev, capturable=True)
nfp.model.train(); nfp.model.static_shapes = True
mask = np.zeros((64, 64), dtype=bool)
x, y = synthetic.make_batch(2, 0, 32, 10, 10, n_digits=2, pixel_noise=0.05)
x, y = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
c = torch.zeros(32, 10, 64, 64, 1, device=dev)
for _ in range(2): nfp.train_step(x, y, c, mask)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    nfp.train_step(x, y, c, mask)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=6):
    if e.device_time_total > 0 and e.key.startswith('aten::'):
        stack = [s for s in e.stack if 'quadtree-mpnnlstm_amd' in s or 'torch/optim' in s or 'clip_grad' in s]
        rows.append((e.count, e.device_time_total, e.key, stack[0].split('quadtree-mpnnlstm_amd/')[-1][:70] if stack else '(autograd / other)'))
rows.sort(key=lambda r: -r[0])
for cnt, t, key, where in rows[:45]:
    print(f'{cnt:5d} {t/1e3:8.2f} ms  {key:32s} {where}')
