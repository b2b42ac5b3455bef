"""Practical ceiling for launches of the step's sizes: plain streaming kernels (torch copy / add / sum / fill) that move the same
bytes in ONE launch, graph-replayed back to back like bench.py's roofline probe -- what the 8 TB/s figure turns into for a launch
that lasts tens of microseconds (ramp-up, drain, read / write mix).   python tools/exp_ceiling.py"""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import torch
dev = torch.device('cuda', 0)
big = torch.empty(96 * 1024 * 1024, device=dev)           # 384 MB written between probes: nothing stays in the 256 MB cache


def graph_time(body, reps=10):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(reps):
            body()
    g.replay(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); e.record(); e.synchronize()
    return a.elapsed_time(e) * 1e3 / reps


_EVICT = [None]


def timeit(fn, reps=10, cold=False):
    """back to back: `reps` launches in one graph (operands of <= 256 MB stay in the Infinity Cache between them);
    cold: every launch preceded, inside the graph, by a 384 MB fill (the cache then holds that fill's dirty lines, as it holds the
    previous kernels' outputs inside a training step), minus the time of the fills alone."""
    fn(); torch.cuda.synchronize()
    if not cold:
        return graph_time(fn, reps)
    if _EVICT[0] is None:
        _EVICT[0] = graph_time(lambda: big.fill_(1.0), reps)

    def both():
        big.fill_(1.0)
        fn()
    return graph_time(both, reps) - _EVICT[0]


def probe(label, mb_read, mb_write):
    n_w = int(mb_write * 250_000) if mb_write else 0
    n_r = int(mb_read * 250_000)
    if n_w == 0:                                    # read only: a sum
        src = torch.randn(n_r, device=dev); out = torch.empty((), device=dev)
        fn = lambda: torch.sum(src, dim=0, out=out)
    elif mb_read == 0:                              # write only
        dst = torch.empty(n_w, device=dev)
        fn = lambda: dst.fill_(1.0)
    else:
        k = max(1, round(n_r / n_w))                # reads k operands of the output's size, writes one
        srcs = [torch.randn(n_w, device=dev) for _ in range(k)]
        dst = torch.empty(n_w, device=dev)
        if k == 1:
            fn = lambda: dst.copy_(srcs[0])
        elif k == 2:
            fn = lambda: torch.add(srcs[0], srcs[1], out=dst)
        else:
            st = torch.stack(srcs)
            fn = lambda: torch.sum(st, dim=0, out=dst)
        mb_read = k * mb_write
    tw, tc = timeit(fn), timeit(fn, cold=True)
    tot = mb_read + mb_write
    print(f'{label:58s} read {mb_read:6.0f} MB + write {mb_write:5.0f} MB: back to back {tw:7.2f} us ({tot / tw:5.2f} TB/s)   after a 384 MB fill {tc:7.2f} us ({tot / tc:5.2f} TB/s)')


probe('(calibration: a 4 KB copy -- the fixed cost of a measurement)', 0.004, 0.004)
probe('k_cheb_clip forward (its own bytes: 48 MB)', 30, 15)
probe('k_remesh_clip (68 MB)', 34, 34)
probe('k_gate_cell_p (~105 MB)', 52, 52)
probe('k_dgrad_cell (132 MB by PMC)', 66, 66)
probe('k_gemm_wgrad_group (750 MB read)', 750, 0)
probe('k_spmm on 256 x 256 frames (81 MB by PMC)', 54, 27)
probe('forward projection, cfg4t (103 MB read, 410 MB written)', 103, 410)
probe('k_proj_bwd, cfg4t (513 MB read, 103 MB written)', 515, 103)
probe('k_attn_fwd operands (0.71 GB)', 568, 142)
probe('read-heavy: 8 operands summed into one (94 MB each)', 752, 94)
probe('read-heavy: 16 operands summed into one (47 MB each)', 752, 47)
probe('a fill of 410 MB', 0, 410)
probe('a copy of 2 GB', 1000, 1000)
