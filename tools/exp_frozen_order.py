"""Frozen-model step time (lr = 0) of several predictors created one after another in ONE process: is a later predictor faster
than the first one whatever its arithmetic (allocator / cache state), or is the split-bf16 data gradient what makes it faster?"""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
import bench
from qtmpnn import ops, synthetic
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
mask = np.zeros((64, 64), dtype=bool)
pool = []
for i in range(4):
    x, y = synthetic.make_batch(2, i * 32, 32, 10, 10, n_digits=2, pixel_noise=0.05, canvas=(64, 64))
    pool.append((torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), torch.zeros(32, 10, 64, 64, 1, device=dev)))


def frozen(split, n=20):
    prev = ops.set_dgrad_split_bf16(split)
    try:
        p = bench.make_predictor(dev, capturable=True)
        p.model.train()
        step = p.make_graphed_step(*pool[0], mask=mask, warmup=2)
        g = p.optimizer.param_groups[0]
        g['lr'].fill_(0.0)
        for i in range(3):
            step(*pool[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            l = step(*pool[i % 4])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        nodes = [int(ms.n_dev.item()) for ms in getattr(p.model, '_last_meshes', [])]
        return dt, float(l)
    finally:
        ops.set_dgrad_split_bf16(prev)


for name, split in [('exact #1', False), ('exact #2', False), ('split #3', True), ('exact #4', False), ('split #5', True)]:
    dt, l = frozen(split)
    print(f'{name}: {dt:.3f} ms per frozen step, loss {l:.5f}', flush=True)
