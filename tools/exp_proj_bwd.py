"""One-pass projection backward (csrc/projbwd.hip) against the two launches it replaces, graph-replayed on the cfg4t shape
(8 groups, N = 1.24e5 rows, hidden 32): microseconds per layer-use.   python tools/exp_proj_bwd.py [N]"""
import ctypes, os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
import torch
from qtmpnn import _lib
if os.environ.get('QT_LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'micro', os.environ['QT_LIB'])
from qtmpnn._lib import ptr
dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 124000
G, cin, C = 8, 32, 32
co = 4 * C
A = torch.randn(G, N, cin, device=dev)
gP = torch.randn(G, 4, N, C, device=dev)
W = torch.randn(G, cin + 4, co, device=dev)
nxt = torch.empty(G, 4, N, cin, device=dev)
nb = _lib.value('qt_proj_bwd_blocks', G)
part = torch.zeros(nb, G, cin + 4, co, device=dev)
ones = torch.zeros(N, 4, device=dev); ones[:, 0] = 1
Ns = (ctypes.c_int * 1)(N)
nbw = _lib.value('qt_wgrad_group_blocks', 1, Ns)
partw = torch.empty(nbw, G, cin + 4, co, device=dev)
vp = ctypes.c_void_p * 1
big = torch.empty(80 * 1024 * 1024, device=dev)          # 320 MB written between repetitions: nothing stays in the 256 MB cache


def fused():
    _lib.call('qt_proj_bwd', ptr(gP), 4 * N * C, N * C, ptr(A), N * cin, cin, ptr(W), (cin + 4) * co, co, nxt.data_ptr() + 4 * 3 * N * cin,
              4 * N * cin, cin, ptr(part), N, None, G, cin, C, 1, 1)


def dgrad():
    _lib.call('qt_proj_group', ptr(gP), C, 4 * N * C, 4, C, None, None, ptr(W), (cin + 4) * co, G, 1, cin, nxt.data_ptr() + 4 * 3 * N * cin,
              cin, 4 * N * cin, 1, N, None)


def wgrad():
    _lib.call('qt_wgrad_groups', 1, vp(A.data_ptr()), (ctypes.c_int * 1)(cin), vp(ones.data_ptr()), vp(gP.data_ptr()), Ns, vp(None), cin, 4, co,
              C, C, G, cin, co, 1, ptr(partw))


def timeit(fn, reps=10, flush=True):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    ts = []
    for _ in range(reps):
        if flush:
            big.fill_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


by = 4.0 * (G * 4 * N * C + 2 * G * N * cin)
for flush in (True, False):
    tf, td, tw = timeit(fused, flush=flush), timeit(dgrad, flush=flush), timeit(wgrad, flush=flush)
    print(f'N = {N}, operands {"cold" if flush else "warm"}: one pass {tf:7.1f} us ({by / tf / 1e6:.2f} TB/s of its operands once, '
          f'{2 * 2.0 * G * N * 128 * 32 / tf / 1e6:.1f} TFLOP/s)  |  data gradient {td:7.1f} + weight gradient {tw:7.1f} = {td + tw:7.1f} us')

# a cell's first layer: four heads share one input and one (36, 512) weight matrix; the four partial data gradients are added afterwards
H4 = 4
A1 = torch.randn(N, cin, device=dev)
W1 = torch.randn(1, cin + 4, H4 * co, device=dev)
gP1 = gP[:H4]
partial = torch.empty(H4, N, cin, device=dev)
nb1 = _lib.value('qt_proj_bwd_blocks', H4)
part1 = torch.zeros(nb1, 1, cin + 4, H4 * co, device=dev)
out1 = torch.empty(N, cin, device=dev)
partw1 = torch.empty(_lib.value('qt_wgrad_group_blocks', 1, Ns), 1, cin + 4, H4 * co, device=dev)


def shared():
    _lib.call('qt_proj_bwd', ptr(gP1), 4 * N * C, N * C, ptr(A1), 0, cin, ptr(W1), co, H4 * co, ptr(partial), N * cin, cin, ptr(part1), N, None,
              H4, cin, C, 1, 1)


def shared_sum():
    shared()
    torch.sum(partial, dim=0, out=out1)


def dgrad1():
    _lib.call('qt_proj_group', ptr(gP1), C, 4 * H4 * N * C, 4 * H4, C, None, None, ptr(W1), (cin + 4) * H4 * co, 1, 1, cin, ptr(out1), cin, 0, 1, N, None)


def wgrad1():
    _lib.call('qt_wgrad_groups', 1, vp(A1.data_ptr()), (ctypes.c_int * 1)(cin), vp(ones.data_ptr()), vp(gP1.data_ptr()), Ns, vp(None), cin, 4, H4 * co,
              C, C, 1, 0, H4 * co, 1, ptr(partw1))


for flush in (True, False):
    ts, tss, td, tw = (timeit(f, flush=flush) for f in (shared, shared_sum, dgrad1, wgrad1))
    print(f'first layer (4 heads, one input), operands {"cold" if flush else "warm"}: one pass {ts:7.1f} us, with the sum of the partials {tss:7.1f} us  |  '
          f'data gradient {td:7.1f} + weight gradient {tw:7.1f} = {td + tw:7.1f} us')

if os.environ.get('QT_LIB', '').startswith('libqt_pbtiming'):
    import ctypes as C_
    lib = _lib.load()
    dbg = torch.zeros(G * nb * 4 * 6, dtype=torch.int64, device=dev)
    lib.qt_proj_bwd_timing_buffer.argtypes = [C_.c_void_p]
    lib.qt_proj_bwd_timing_buffer(dbg.data_ptr())
    fused(); torch.cuda.synchronize()
    d = dbg.view(G * nb, 4, 6).double().cpu() / 100.0          # wall_clock64: 100 MHz -> microseconds
    names = ['wait at barrier 1', 'stash (incl. wait for operands)', 'barrier 2', 'issue prefetch', 'MFMA chain', 'row stores']
    for role, sl in (('data-gradient waves', slice(0, 2)), ('weight-gradient waves', slice(2, 4))):
        m = d[:, sl].mean(dim=(0, 1))
        print(role + ': ' + ', '.join(f'{n} {v:.1f} us' for n, v in zip(names, m.tolist())) + f'  | sum {float(m.sum()):.1f} us')
