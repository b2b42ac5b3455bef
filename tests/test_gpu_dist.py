"""Data-parallel equivalence on the GPU: two ranks training on half the clips each must reproduce the single-process step
on all clips: same loss (mean of the rank losses) and the same weights after the update.  With two or more GPUs visible the
ranks take one GPU each and exchange gradients over RCCL (backend "nccl"); on a one-GPU box both ranks share cuda:0 and
the transport is gloo (RCCL needs one GPU per rank).  bench.py's multi-rank path (probes on rank 0 only while the other
ranks wait) is run the same way."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

T, B = 3, 4


def _backend():
    return 'nccl' if torch.cuda.device_count() >= 2 else 'gloo'


def _setup(rank=0):
    from model.mpnnlstm import NextFramePredictorS2S
    torch.manual_seed(11)
    dev = torch.device('cuda', rank if torch.cuda.device_count() >= 2 else 0)
    torch.cuda.set_device(dev)
    nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=T, output_timesteps=T, device=dev,
                                model_kwargs=dict(hidden_size=8, dropout=0.0, n_layers=1))
    nfp.initiate_training(lr=1e-3, lr_decay=0.95, capturable=True)
    return nfp, dev


def _data(dev, lo, hi):
    from qtmpnn import synthetic
    x, y = synthetic.make_batch(3, 0, B, T, T, n_digits=1, pixel_noise=0.02)
    t = lambda a: torch.from_numpy(a[lo:hi]).to(dev)
    return t(x), t(y), torch.zeros(hi - lo, T, 64, 64, 1, device=dev)


def _rank(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from qtmpnn.dist import init_from_env, shard_range
    init_from_env(_backend())
    nfp, dev = _setup(rank)
    lo, hi = shard_range(B, rank, world)
    x, y, c = _data(dev, lo, hi)
    mask = np.zeros((64, 64), dtype=bool)
    step = nfp.make_graphed_step(x, y, c, mask, warmup=1)          # 1 eager DP step, then graph1 + all-reduce + graph2
    losses = [float(step(x, y, c)) for _ in range(2)]
    out.put((rank, losses, {k: v.detach().cpu().numpy() for k, v in nfp.model.state_dict().items()}))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_equal_one_process():
    nfp, dev = _setup()
    x, y, c = _data(dev, 0, B)
    mask = np.zeros((64, 64), dtype=bool)
    ref_losses = [float(nfp.train_step(x, y, c, mask)) for _ in range(3)]
    ref = {k: v.detach().cpu() for k, v in nfp.model.state_dict().items()}

    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ['PYTHONPATH'] = os.pathsep.join([os.path.join(root, 'quadtree-mpnnlstm_amd'), root, os.path.join(root, 'tests'),
                                                os.environ.get('PYTHONPATH', '')])
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((out.get(timeout=240) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # steps 2 and 3 of the reference correspond to the two graphed steps; global loss = mean over ranks
    for i in range(2):
        glob = 0.5 * (res[0][1][i] + res[1][1][i])
        assert abs(glob - ref_losses[i + 1]) <= 1e-4 * abs(ref_losses[i + 1]), (i, glob, ref_losses, res[0][1], res[1][1])
    for k in ref:
        np.testing.assert_allclose(res[0][2][k], res[1][2][k], rtol=0, atol=0, err_msg=f'ranks diverged: {k}')
        np.testing.assert_allclose(res[0][2][k], ref[k].numpy(), rtol=2e-3, atol=2e-4, err_msg=k)



def _rccl_world_of_one(port, out):
    """Child: backend "nccl" (= RCCL) with ONE rank on cuda:0; the multi-rank step structure forced (graph1, all-reduce, graph2)
    against the single-graph step from the same seed."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    assert dist.get_backend() == 'nccl'
    t = torch.arange(8, dtype=torch.float32, device='cuda:0')
    dist.all_reduce(t)                                   # the communicator (and RCCL's kernels) exist before any capture
    torch.cuda.synchronize()
    assert t.tolist() == list(range(8))
    mask = np.zeros((64, 64), dtype=bool)
    res = {}
    for name, force in (('forced', True), ('single', False)):
        nfp, dev = _setup()
        x, y, c = _data(dev, 0, B)
        step = nfp.make_graphed_step(x, y, c, mask, warmup=1, force_multi=force)
        losses = [float(step(x, y, c)) for _ in range(3)]
        torch.cuda.synchronize()
        res[name] = (losses, {k: v.detach().cpu().numpy() for k, v in nfp.model.state_dict().items()})
    rccl = [m.split()[-1] for m in open('/proc/self/maps') if 'librccl' in m]
    out.put((res, sorted(set(rccl))))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_runs_the_multi_rank_step_structure_in_a_world_of_one():
    """No second GPU is ever visible to these tests, so RCCL itself is exercised with world_size = 1: under backend "nccl" the
    forced multi-rank step -- graph1.replay() (forward + loss + backward), dist.all_reduce(flat) on the replay stream,
    graph2.replay() (average + clip + fused Adam) -- must give bit-identical losses and weights to the single-graph step over
    three steps.  It proves that librccl loads, that the collective is ordered correctly between two hipGraph replays on the
    side stream, and that capture_error_mode='thread_local' survives the process group's watchdog thread."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ['PYTHONPATH'] = os.pathsep.join([os.path.join(root, 'quadtree-mpnnlstm_amd'), root, os.path.join(root, 'tests'),
                                                os.environ.get('PYTHONPATH', '')])
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    p = ctx.Process(target=_rccl_world_of_one, args=(port, out))
    p.start()
    res, rccl = out.get(timeout=300)
    p.join(60)
    assert p.exitcode == 0
    assert rccl, 'librccl is not mapped into the process: the collective did not go through RCCL'
    (la, wa), (lb, wb) = res['forced'], res['single']
    assert la == lb, (la, lb)
    for k in wa:
        assert np.array_equal(wa[k], wb[k]), k


def test_bench_two_ranks_prints_one_line_with_probes():
    """bench.py --gpus 2 (spawned through torchrun as the driver does): rank 0 runs the roofline probes alone after the
    timed region, so they must be collective-free -- the run has to end with ONE JSON line carrying `roofline`."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_PORT=str(port), QT_DIST_BACKEND=_backend())
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '4',
           '--frozen-steps', '2']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['config']['global_batch'] == 8 and rec['value'] > 0
    assert rec['roofline'].get('frac', 0) > 0, rec['roofline']
    assert rec['frozen_ms_per_step'] > 0
