"""Child process of tests/test_gpu_ops.py::test_tile_capacity_overflow_is_reported_not_waited_for.  Started with QT_LIB_PATH
pointing at libqtmpnn_hip_smallcaps.so (QT_TILE_HALO_CAP = 16 instead of 256): on a noisy 128 x 128 frame every tile has more
boundary rows than that, so qt_edges_norm_tiles runs its overflow branch -- sentinel address -1 for the rows without a record, bit 1
in the caller's error word -- and the tile-resident launches neither wait for those rows nor fault: they return at once with
(documented) garbage and the same bit."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
from qtmpnn import _lib, ops, synthetic                      # noqa: E402
from qtmpnn import mesh as M                                 # noqa: E402

assert _lib.LIB_PATH.endswith('libqtmpnn_hip_smallcaps.so') and _lib.value('qt_tile_cap', 2) == 16, _lib.LIB_PATH
dev = torch.device('cuda', 0)
B = 2
img = np.stack([synthetic.make_clip(40 + i, canvas=(128, 128), n_digits=2, n_frames=1, pixel_noise=0.05)[0, ..., 0] for i in range(B)])
mesh = M.build_mesh(src=torch.from_numpy(img).to(dev), thresh=0.1)
tl = mesh.tiles
assert tl is not None and tl['brec'].shape[1] == 16
cnt = tl['cnt'].cpu().numpy().reshape(-1, 32)
assert cnt[:, 5].all(), cnt[:, :6]                            # every tile overflowed
assert int(tl['err']) == 2                                    # reported by the mesh build already
baddr = tl['baddr'].cpu().numpy()
off = mesh.cell_off.cpu().numpy()
rp, col = mesh.rowptr.cpu().numpy(), mesh.col.cpu().numpy()
nsent = 0
for ts in range(cnt.shape[0]):                                # every boundary row: a valid address inside its tile's slots, or -1
    t0, t1 = int(off[ts]), int(off[ts + 1])
    for r in range(t0, t1):
        if any(not (t0 <= c < t1) for c in col[rp[r]:rp[r + 1]]):
            assert baddr[r] == -1 or ts * 16 <= baddr[r] < ts * 16 + 16, (ts, r, baddr[r])
            nsent += baddr[r] == -1
assert nsent > 0
M.tile_error_word(reset=True)
K, N = 5, mesh.N
Z = torch.randn(N, 16, device=dev)
out = torch.empty(K - 1, N, 16, device=dev)
t0 = time.time()
ops.clip_planes(mesh, [Z], [out], K)
G = torch.randn(K, N, 16, device=dev)
ops.clip_clenshaw(mesh, [G], K)
torch.cuda.synchronize()
dt = time.time() - t0
assert dt < 5.0, f'{dt:.2f} s: the launches waited for rows that were never going to be published'
assert M.tile_error_word() == 2, M.tile_error_word()          # capacity only: nobody timed out
try:
    M.check_tile_errors(always=True)
    raise SystemExit('check_tile_errors did not raise')
except RuntimeError as e:
    assert 'capacity' in str(e), e
print(f'tile overflow ok: {nsent} boundary rows without a record, launches took {dt * 1e3:.1f} ms')
