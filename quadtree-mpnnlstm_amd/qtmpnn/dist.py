"""Data-parallel glue: clips shard across ranks, one flat gradient all-reduce per step.

The reference has no distributed code.  Clips are independent, the model is
tens of thousands of parameters, so the only exchange is ONE all-reduce of a single flat fp32
buffer (135 KiB for the Moving-MNIST model) -- latency-bound on xGMI, hence one call, never one
per parameter tensor.  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* when launched by torchrun.
    Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if os.environ['MASTER_ADDR'] in ('127.0.0.1', 'localhost'):
            # one node: gloo (the CPU tests' backend and the host-side barrier's side group) binds to the loopback interface instead
            # of looking the host name up (container host names may not resolve)
            os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class HostBarrier:
    """A barrier the ranks wait in on the HOST: a gloo side group over the same rendezvous (public API only --
    `dist.new_group(backend='gloo')` + `dist.barrier(group=...)`; round 3 polled the private default store).  A rank that waits
    here for another rank's seconds of single-rank work sits in no RCCL collective: nothing for the watchdog to time, no
    GPU spin.  Create it right after init_process_group, on every rank (new_group is itself collective)."""

    def __init__(self, timeout_s=1800):
        import datetime
        self.group = None
        if dist.is_initialized() and dist.get_world_size() > 1:
            self.group = dist.new_group(backend='gloo', timeout=datetime.timedelta(seconds=timeout_s))

    def wait(self):
        if self.group is not None:
            dist.barrier(group=self.group)


def _via_host(t, group):
    """True when a CUDA tensor's collective runs over gloo (the one-GPU rehearsal transport of the tests; RCCL is the product's)."""
    return t.is_cuda and dist.get_backend(group) == 'gloo'


def all_reduce_sum(t, group=None):
    """Sum `t` over the group in place, ordered on torch's current stream like any other launch.  Over RCCL that is what
    dist.all_reduce does.  Over gloo a CUDA tensor is staged through host memory HERE, with plain synchronous copies: gloo's own
    CUDA staging (pinned buffers on side streams) once handed two ranks that share one GPU a stale gradient (a 1.7 % loss jump in
    one of ~15 runs of tests/test_gpu_dist.py; the RCCL path is bit-exact against the single-graph step every time)."""
    if _via_host(t, group):
        h = t.detach().cpu()                # (synchronises with the producing stream)
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        torch.cuda.current_stream().synchronize()
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items clips for `rank` (equal sizes; n_items % world == 0)."""
    assert n_items % world == 0, f'{n_items} clips do not split evenly over {world} ranks'
    per = n_items // world
    return rank * per, (rank + 1) * per


def allreduce_gradients(params, group=None, force=False):
    """Average gradients over the group with a single flat all-reduce (missing grads count as zero).  force: issue the
    collective in a group of one rank too (it changes nothing there; used to exercise the transport)."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    all_reduce_sum(flat, group)
    flat.div_(world)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s weights (one flat broadcast)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    params = list(module.parameters())
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    if _via_host(flat, group):
        h = flat.cpu()
        dist.broadcast(h, src=src, group=group)
        flat.copy_(h)
    else:
        dist.broadcast(flat, src=src, group=group)
    off = 0
    with torch.no_grad():
        for p in params:
            p.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
