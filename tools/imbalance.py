"""Load imbalance of the data-parallel step from data-dependent meshes, measured on ONE GPU: the 8 shards an 8-rank weak-scaling
run of bench.py would see (rank r owns clips [r*B, (r+1)*B) of every global batch: the same seeds as bench.py) are timed one
after the other with the same weights.  Two series: the model as initialised with the learning rate at 0 (meshes of the decoder
follow the untrained model's output) and 30 real training steps per shard.  max / mean of the per-shard step times bounds the
weak-scaling efficiency from above (every step ends with an all-reduce, so the slowest rank sets the pace); the all-reduce itself
(135 KiB, one call) and host launch jitter are NOT in this number.    python tools/imbalance.py [world]"""
import json, os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
import bench
from qtmpnn import synthetic
dev = torch.device('cuda', 0)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B, n_pool = 32, 4
mask = np.zeros(bench.CANVAS, dtype=bool)


def shard_pool(rank):
    pool = []
    for i in range(n_pool):
        x, y = synthetic.make_batch(2, (i * world + rank) * B, B, bench.T_IN, bench.T_OUT, n_digits=bench.N_DIGITS,
                                    pixel_noise=bench.NOISE, canvas=bench.CANVAS)
        pool.append((torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), torch.zeros(B, bench.T_OUT, *bench.CANVAS, 1, device=dev)))
    return pool


def timed(step, pool, n):
    for i in range(3):
        step(*pool[i % n_pool])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(*pool[i % n_pool])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {'frozen': [], 'trained': []}
for rank in range(world):
    pool = shard_pool(rank)
    for mode in ('frozen', 'trained'):
        os.environ['QT_BENCH_LR'] = '0' if mode == 'frozen' else str(bench.LR)
        nfp = bench.make_predictor(dev, capturable=True)
        nfp.model.train()
        step = nfp.make_graphed_step(*pool[0], mask=mask, warmup=2)
        res[mode].append(round(timed(step, pool, 30), 3))
        del step, nfp
    print(f'shard {rank}: frozen {res["frozen"][-1]} ms, trained {res["trained"][-1]} ms', file=sys.stderr, flush=True)
out = {'world': world, 'clips_per_rank': B, 'ms_per_step': res}
for mode, v in res.items():
    out[mode + '_max_over_mean'] = round(max(v) / (sum(v) / len(v)), 4)
    out[mode + '_efficiency_bound'] = round((sum(v) / len(v)) / max(v), 4)
print(json.dumps(out))
