"""Practical ceiling for a launch of k_spmm's size: a plain device copy that moves the same bytes (diagnostics).
Graph-replayed like bench.py's roofline probe, so the numbers are comparable."""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import torch
dev = torch.device('cuda', 0)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); e.record(); e.synchronize()
    return a.elapsed_time(e) * 1e3 / reps


for mb_read, mb_write in ((15, 15), (21, 10), (8, 4), (4, 2)):
    n_r, n_w = mb_read * 250_000, mb_write * 250_000
    src = torch.randn(n_r, device=dev)
    dst = torch.empty(n_w, device=dev)
    if n_r == n_w:
        fn = lambda: dst.copy_(src)
    else:
        a, b = src[:n_w], src[n_w:2 * n_w] if 2 * n_w <= n_r else src[:n_w]
        fn = lambda: torch.add(a, b, out=dst)           # reads 2 x n_w floats, writes n_w
        mb_read = 2 * mb_write
    t = timeit(fn)
    print(f'read {mb_read} MB + write {mb_write} MB: {t:.2f} us  ({(mb_read + mb_write) * 1e6 / t / 1e3:.0f} GB/s)')
