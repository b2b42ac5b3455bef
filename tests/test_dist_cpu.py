"""world_size-2 gloo tests (CPU) of the data-parallel glue: shard ranges, parameter broadcast and the single flat
gradient all-reduce that is the path's only exchange."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from qtmpnn.dist import HostBarrier, allreduce_gradients, broadcast_parameters, init_from_env, shard_range
    r, w, _ = init_from_env('gloo')
    assert (r, w) == (rank, world)
    side = HostBarrier(timeout_s=60)                   # gloo side group (public API): what bench.py's ranks wait in on the host
    torch.manual_seed(100 + rank)                      # different initial weights per rank
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    broadcast_parameters(model)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(g, gathered[0]) for g in gathered), 'broadcast did not equalise the weights'
    # rank-dependent gradients; one parameter has no gradient on rank 1 (counts as zero)
    params = list(model.parameters())
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    if rank == 1:
        params[-1].grad = None
    allreduce_gradients(params)
    for i, p in enumerate(params):
        want = (1 * (i + 1) + (2 * (i + 1) if i < len(params) - 1 else 0.0)) / 2.0
        assert torch.allclose(p.grad, torch.full_like(p, want)), (i, p.grad.flatten()[0].item(), want)
    lo, hi = shard_range(64, rank, world)
    assert (lo, hi) == (32 * rank, 32 * rank + 32)
    # rank 1 waits on the host while rank 0 'works' for a while: both leave the barrier only after rank 0 arrives
    import time
    t0 = time.perf_counter()
    if rank == 0:
        time.sleep(0.5)
    side.wait()
    assert time.perf_counter() - t0 >= 0.45
    side.wait()
    dist.barrier()
    dist.destroy_process_group()
    out.put(rank)


def test_flat_allreduce_broadcast_and_sharding_world2():
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'quadtree-mpnnlstm_amd')
    os.environ['PYTHONPATH'] = pkg + os.pathsep + os.environ.get('PYTHONPATH', '')
    if pkg not in sys.path:
        sys.path.insert(0, pkg)
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, f'rank exited with {p.exitcode}'
    assert sorted(out.get(timeout=5) for _ in range(2)) == [0, 1]
