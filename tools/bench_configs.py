"""Throughput of the other BASELINE.json configurations on one GPU (informational; bench.py stays on configs[1]).

    python tools/bench_configs.py [cfg3|cfg4] [steps]
cfg3: Moving-MNIST-like 128x128, 2 digits, in=10/out=20, 8 clips per GPU (the per-GPU share of the 8-GPU config).
cfg4: ice-like 128x128 patches, 5 channels, in=12/out=6, 16 clips, land mask, transform_func, hidden 32, 1 layer, 3 conv layers.
cfg4t: cfg4 with convolution_type='TransformerConv' (what ice_exp.py hard-codes).
cfg4tp: cfg4t on the pixelwise mesh (thresh=-inf: what ice_exp.py:145 really runs -- no quadtree, one node per unmasked pixel).
cfg5: ice-like 256x256, 5 channels, in=12/out=12, 4 clips per GPU (the per-GPU share of BASELINE configs[4]: 32 clips over 8 GPUs),
      quadtree rebuilt at every step (ice_exp_nwt.py:46,80,89-96 with a finite threshold); cfg5t: the same with TransformerConv.
cfg2n0 / cfg3n0: the Moving-MNIST configs with noise-free inputs (SURVEY 8(d)'s second series: sparse input meshes).
"""
import json, os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
from model.mpnnlstm import NextFramePredictorS2S
from qtmpnn import synthetic
dev = torch.device('cuda', 0)
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.manual_seed(1)
if cfg in ('cfg3', 'cfg3n0', 'cfg2n0'):
    B, t_in, t_out, shape = (8, 10, 20, (128, 128)) if cfg != 'cfg2n0' else (32, 10, 10, (64, 64))
    kw, thresh, tf, feat = dict(hidden_size=16, dropout=0.1, n_layers=2), 0.1, None, 1
    mask = np.zeros(shape, dtype=bool)
    noise = 0.0 if cfg.endswith('n0') else 0.05
    def batch(i):
        x, y = synthetic.make_batch(2 if cfg == 'cfg2n0' else 3, i * B, B, t_in, t_out, n_digits=2, pixel_noise=noise, canvas=shape)
        return x, y
else:
    B, t_in, t_out, shape = (16, 12, 6, (128, 128)) if not cfg.startswith('cfg5') else (4, 12, 12, (256, 256))
    kw, thresh, feat = dict(hidden_size=32, dropout=0.1, n_layers=1, n_conv_layers=3), 0.15, 5
    if cfg in ('cfg4t', 'cfg4tp', 'cfg5t'):
        kw['convolution_type'] = 'TransformerConv'
    if cfg == 'cfg4tp':
        thresh = -np.inf
    tf = lambda a: abs(abs(a - 0.5) - 0.5)
    kw['transform_func'] = tf
    mask = synthetic.make_ice_like(40, shape=shape, channels=5, n_frames=2)[1]
    def batch(i):
        clips = [synthetic.make_ice_like(1000 * i + k, shape=shape, channels=5, n_frames=t_in + t_out)[0] for k in range(B)]
        return np.stack([c[:t_in] for c in clips]), np.stack([c[t_in:, ..., :1] for c in clips])
nfp = NextFramePredictorS2S(thresh=thresh, input_features=feat, input_timesteps=t_in, output_timesteps=t_out, device=dev,
                            transform_func=tf, model_kwargs=kw)
nfp.initiate_training(lr=0.01, lr_decay=0.95, capturable=True)
nfp.model.train()
pool = []
for i in range(2):
    x, y = batch(i)
    pool.append((torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), torch.zeros(B, t_out, *shape, 1, device=dev)))
step = nfp.make_graphed_step(*pool[0], mask=mask, warmup=2)
for i in range(3):
    l = step(*pool[i % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    l = step(*pool[i % 2])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
rec = {}
if os.environ.get('QT_CFG_ROOFLINE', '1') == '1':
    # the message-aggregate roofline object of bench.py for THIS configuration (per-launch algorithmic bytes / measured launch
    # time, HIP events around graph replays of every distinct launch of one forward + backward on the model as initialised)
    sys.path.insert(0, ROOT)
    import bench
    del step
    nfp._graph = None
    torch.manual_seed(1)
    fresh = NextFramePredictorS2S(thresh=thresh, input_features=feat, input_timesteps=t_in, output_timesteps=t_out, device=dev,
                                  transform_func=tf, model_kwargs=kw)
    fresh.initiate_training(lr=0.01, lr_decay=0.95, capturable=False)
    fresh.model.train()
    # HBM-side bytes per launch from the committed PMC passes of THIS configuration, when there are any
    tf_name = f'r05_pmc_traffic_{cfg}.json'
    if kw.get('convolution_type', 'ChebConv') == 'ChebConv':
        rec['roofline'] = bench.spmm_roofline(fresh, pool[0], mask, traffic_files=(tf_name,))
        pmc = os.path.join(ROOT, 'profiles', tf_name)
        if os.path.exists(pmc):       # per-kernel records: the per-hop kernel's own figure beside the launch-weighted one
            ks = json.load(open(pmc))['kernels']
            rec['roofline']['traffic_by_kernel'] = {k: v['traffic_bytes_per_launch'] for k, v in ks.items()
                                                    if k.startswith(('k_spmm<', 'k_cheb_clip'))}
            # `traffic`: launch-weighted over ALL message-aggregate launches of the step (per-hop k_spmm on rows of >= 4 channels and
            # the fused launches), not over the fused ones alone -- on the big frames the per-hop kernel dominates
            agg = [v for k, v in ks.items() if k.startswith('k_cheb_clip') or (k.startswith('k_spmm<') and not k.startswith('k_spmm<1,'))]
            if agg:
                rec['roofline']['traffic'] = int(sum(v['traffic_bytes_per_launch'] * v['dispatches'] for v in agg) / sum(v['dispatches'] for v in agg))
                rec['roofline']['traffic_source'] = 'profiles/' + tf_name
    else:
        rec['roofline'] = bench.attn_roofline(fresh, pool[0], mask)
        pmc = os.path.join(ROOT, 'profiles', tf_name)
        if os.path.exists(pmc):
            ks = json.load(open(pmc))['kernels']
            rec['roofline']['traffic_by_kernel'] = {k: v['traffic_bytes_per_launch'] for k, v in ks.items() if k.startswith('k_attn')}
            rec['roofline']['traffic_source'] = 'profiles/' + tf_name
    rec['rollout_sizes'] = bench.rollout_sizes(fresh, pool[0], mask)
from qtmpnn.mesh import tile_error_word
assert tile_error_word() == 0, 'a tile-resident launch reported an error (persistent error word)'
print(json.dumps({'config': cfg, 'frames_per_s': round(B * (t_in + t_out) * steps / dt, 1), 'ms_per_step': round(dt / steps * 1e3, 2),
                  'clips': B, 'shape': shape, 't_in': t_in, 't_out': t_out, 'loss': round(float(l), 5), 'launch': 'hipGraph replay', **rec}))
