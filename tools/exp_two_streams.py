"""Would two independent half-batch chains on two streams inside one hipGraph beat one full-batch chain?  (diagnostics)
Two separate predictors (no shared weights -- only the GPU-side concurrency is being measured) step 16 clips each on their
own stream inside one captured graph, against one predictor stepping 32 clips."""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import numpy as np, torch
from model.mpnnlstm import NextFramePredictorS2S
from qtmpnn import synthetic
dev = torch.device('cuda', 0)
mask = np.zeros((64, 64), dtype=bool)


def predictor():
    torch.manual_seed(1)
    nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=10, output_timesteps=10, device=dev,
                                model_kwargs=dict(hidden_size=16, dropout=0.1, n_layers=2))
    nfp.initiate_training(lr=0.0, lr_decay=0.95, capturable=True)
    nfp.model.static_shapes = True
    nfp.model.train()
    return nfp


def data(lo, n):
    x, y = synthetic.make_batch(2, lo, n, 10, 10, n_digits=2, pixel_noise=0.05)
    return torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), torch.zeros(n, 10, 64, 64, 1, device=dev)


def fwd_bwd(nfp, b):
    nfp.zero_grad()
    loss = nfp.forward_loss(*b, mask)
    loss.backward()
    nfp._clip_and_step(nfp._grads_ready(), 10.0)
    return loss.detach()


def timeit(g, reps=20):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


full, ha, hb = predictor(), predictor(), predictor()
bf, ba, bb = data(0, 32), data(0, 16), data(16, 16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
s1.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s1):
    for _ in range(2):
        fwd_bwd(full, bf); fwd_bwd(ha, ba); fwd_bwd(hb, bb)
torch.cuda.current_stream().wait_stream(s1)
torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=s1):
    fwd_bwd(full, bf)
print(f'one chain, 32 clips: {timeit(g1):.3f} ms per step')
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=s1):
    fwd_bwd(ha, ba)
    fwd_bwd(hb, bb)
print(f'two chains of 16 clips, one after the other on one stream: {timeit(g2):.3f} ms')
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3, stream=s1):
    s2.wait_stream(s1)
    fwd_bwd(ha, ba)
    with torch.cuda.stream(s2):
        fwd_bwd(hb, bb)
    s1.wait_stream(s2)
print(f'two chains of 16 clips on two streams inside one graph: {timeit(g3):.3f} ms')

for nch in (4, 8):
    per = 32 // nch
    ps = [predictor() for _ in range(nch)]
    bs = [data(i * per, per) for i in range(nch)]
    ss = [torch.cuda.Stream() for _ in range(nch)]
    s1.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1):
        for _ in range(2):
            for p_, b_ in zip(ps, bs):
                fwd_bwd(p_, b_)
    torch.cuda.current_stream().wait_stream(s1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s1):
        for st in ss:
            st.wait_stream(s1)
        for p_, b_, st in zip(ps, bs, ss):
            with torch.cuda.stream(st):
                fwd_bwd(p_, b_)
        for st in ss:
            s1.wait_stream(st)
    print(f'{nch} chains of {per} clips on {nch} streams inside one graph: {timeit(g):.3f} ms')
