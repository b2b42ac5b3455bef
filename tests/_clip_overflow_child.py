"""Child process of tests/test_gpu_ops.py::test_clip_resident_pool_overflow_walks_the_csr.  Started with QT_LIB_PATH pointing at
libqtmpnn_hip_smallcaps.so (the library built with QT_TAIL_CAP = 48): every sparse mesh overflows the per-clip tail pool, so the
rows whose run did not fit carry info base 0xffff and k_cheb_clip walks the CSR arrays for them (gather_tail_csr).  Forward
planes and the Clenshaw backward (row-major and slice-major gradient planes) must equal the per-hop launches bit for bit."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
from qtmpnn import _lib, ops, synthetic                      # noqa: E402
from qtmpnn.mesh import build_mesh, spmm2                    # noqa: E402

assert _lib.LIB_PATH.endswith('libqtmpnn_hip_smallcaps.so') and _lib.value('qt_tail_cap') == 48, _lib.LIB_PATH
dev = torch.device('cuda', 0)
B = 3
img = np.stack([synthetic.make_clip(11 + i, n_frames=1, pixel_noise=0.0)[0, ..., 0] for i in range(B)])
mesh = build_mesh(src=torch.from_numpy(img).to(dev), thresh=0.1)
tinfo = mesh.tail_info.cpu().numpy().view(np.uint32)
tcnt = mesh.tail_cnt.cpu().numpy()
over = int(((tinfo & 0xffff) == 0xffff).sum())
fit = int(((tinfo != 0) & ((tinfo & 0xffff) != 0xffff)).sum())
assert over > 0 and fit > 0 and all(int(tcnt[32 * c]) > 48 for c in range(B)), (over, fit, tcnt[::32])
for K, widths in ((5, (4, 16)), (3, (16,)), (4, (8, 4))):
    for width in (4, 2):
        torch.manual_seed(K)
        N = mesh.N
        Zs = [torch.randn(N, w, device=dev) for w in widths]
        fused = [torch.empty(K - 1, N, w, device=dev) for w in widths]
        assert ops._clip_resident(mesh, list(widths), K)
        ops.clip_planes(mesh, Zs, fused, K, width=width)
        prev, ops._CLIP_CHEB = ops._CLIP_CHEB, False
        try:
            ref, sm = ops._cheb_planes(Zs, mesh, K)
        finally:
            ops._CLIP_CHEB = prev
        for a, r in zip(fused, ref):
            assert torch.equal(ops.planes_rowmajor(a, 1), r), (K, widths, width, 'forward')
        G = [torch.randn(K, N, w, device=dev) for w in widths]
        Gr = [g.clone() for g in G]
        for k in range(K - 2, 0, -1):
            spmm2(mesh, [g[k + 1] for g in Gr], 2.0, [g[k] for g in Gr], 1.0, [g[k + 2] for g in Gr] if k + 2 < K else None, -1.0,
                  [g[k] for g in Gr])
        spmm2(mesh, [g[1] for g in Gr], 1.0, [g[0] for g in Gr], 1.0, [g[2] for g in Gr] if K > 2 else None, -1.0, [g[0] for g in Gr])
        Gf = [g.clone() for g in G]
        ops.clip_clenshaw(mesh, Gf, K, width=width)
        Gs = []
        for g0, w in zip(G, widths):
            t = g0.clone()
            t[1:] = g0[1:].view(K - 1, N, w // 4, 4).permute(0, 2, 1, 3).reshape(K - 1, N, w)
            Gs.append(t)
        ops.clip_clenshaw(mesh, Gs, K, sm=1, width=width)
        for a, s, r in zip(Gf, Gs, Gr):
            assert torch.equal(a[0], r[0]) and torch.equal(s[0], r[0]), (K, widths, width, 'backward')
print(f'overflow ok: {over} rows walk the CSR, {fit} rows run from the pool')
