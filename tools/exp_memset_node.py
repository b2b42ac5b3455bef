"""ONE controlled look at a hipMemsetAsync node inside a captured hipGraph (VERDICT r1 item 10 / HISTORY.md section E: in
round 1 a 4-byte hipMemsetAsync on a counter, issued by a C-ABI entry during the capture of the training step, aborted at
replay and was replaced by a fill kernel; no log was kept).  Each variant runs in its own subprocess, once, under a timeout:

    a: memset of 4 bytes of a tensor allocated BEFORE the capture (ordinary caching-allocator block)
    b: memset of 4 bytes of a tensor allocated INSIDE the capture (graph-private pool), as the round-1 code did
    c: as b, with further allocations after it inside the capture and two graphs sharing the side stream's pool

Prints what each variant's replays leave in the buffer."""
import ctypes, os, subprocess, sys

CODE = r'''
import ctypes, sys, torch
variant = sys.argv[1]
hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
dev = torch.device('cuda', 0)
pre = torch.full((1024,), 7, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
    buf = pre if variant == 'a' else torch.full((1024,), 7, dtype=torch.int32, device=dev)
    if variant == 'c':
        junk = [torch.empty(1 << 20, device=dev) for _ in range(4)]
    buf.add_(1)
    rc = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, 4, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    buf.add_(1)
print('capture rc', rc, flush=True)
for i in range(3):
    g.replay()
    torch.cuda.synchronize()
    print('replay', i, buf[:3].tolist(), flush=True)
print('ok', flush=True)
'''

for v in sys.argv[1:] or ['a', 'b', 'c']:
    try:
        out = subprocess.run([sys.executable, '-c', CODE, v], capture_output=True, text=True, timeout=120)
        print(f'--- variant {v}: exit {out.returncode}\n{out.stdout.strip()}\n{out.stderr.strip()[-600:]}', flush=True)
    except subprocess.TimeoutExpired:
        print(f'--- variant {v}: timed out', flush=True)
