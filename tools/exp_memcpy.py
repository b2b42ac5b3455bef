"""Where do the small device-to-device copies of one training step come from (diagnostics)?"""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    nfp.train_step(*b, mask)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        for k in e.kernels:
            if 'emcpy' in k.name or 'copyBuffer' in k.name or 'emset' in k.name or 'fillBuffer' in k.name:
                chain, q = [e.name], e.cpu_parent
                while q is not None and len(chain) < 5:
                    chain.append(q.name); q = q.cpu_parent
                shapes = str([tuple(s) for s in (e.input_shapes or []) if s][:2])
                cnt[(k.name[:28], ' < '.join(chain)[:110] + ' ' + shapes)] += 1
for k, c in cnt.most_common(25):
    print(c, k)
