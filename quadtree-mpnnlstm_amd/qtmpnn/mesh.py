"""Device-resident quadtree mesh of a batch of B clips (block-diagonal graph).

Replaces the reference's per-clip host state: quadtree labels, the dense (N, P)
`mapping`, `n_pixels_per_node`, `edge_index`, `edge_attr` (model/graph_functions.py:590-681,
model/seq2seq.py:297-304).  Everything is built by the HIP kernels of libqtmpnn_hip.so; the
only host read-back is the node count N (one sync per mesh).
"""
import math
import os

import numpy as np
import torch

from . import _lib
from ._lib import ptr

CONDITIONS = ('max_larger_than', 'max_smaller_than', 'min_larger_than', 'min_smaller_than')


_MASKS = {}
_TILE_ERR = {}           # device -> the persistent error word of the tile-resident launches (tile_error_word / check_tile_errors)
_TILE_USED = set()       # devices that issued a tile-resident launch since their word was last checked
_TILE_CHEB = os.environ.get('QT_NO_TILE_CHEB') != '1'      # (A/B switch: 1 = frames of several base cells stay on one k_spmm launch per hop)
_ONES1 = {}


def _mask_key(a):
    """Cache key of a mask by CONTENT (host arrays / tensors: shape + hash of the bytes, so an array changed in place is a new
    mask and an equal array built anew is the same one) or by identity (CUDA tensors: never read back)."""
    if a is None:
        return None
    if torch.is_tensor(a):
        if a.is_cuda:
            return ('cuda', a.data_ptr(), tuple(a.shape), a._version)
        a = a.numpy()
    arr = np.ascontiguousarray(np.asarray(a) != 0)
    return (arr.shape, hash(arr.tobytes()))


def host_mask(a):
    """A mask as a host bool array (None stays None): numpy arrays, lists, CPU and CUDA tensors alike (a CUDA tensor is read back --
    callers on a hot path keep the result)."""
    if a is None:
        return None
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    return np.asarray(a) != 0


def _as_u8(a, device, shape):
    """Device uint8 copy of a host mask, cached per (content, device): the upload happens once, outside any hipGraph capture
    (a warm-up step always precedes capture).  A mask whose content has not been seen before cannot be uploaded while a
    capture is running -- that is reported as what it is instead of hipErrorStreamCaptureUnsupported."""
    if a is None:
        return None
    if torch.is_tensor(a) and a.is_cuda and a.dtype == torch.uint8:
        return a.contiguous()
    key = (_mask_key(a), str(device))
    hit = _MASKS.get(key)
    if hit is not None:
        return hit
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError('a mask / high_interest_region with new content was passed inside a hipGraph capture: its upload is a '
                           'host-to-device copy, which a capture cannot hold.  Run one eager (warm-up) step with this mask first '
                           '(make_graphed_step does), or pass it as a CUDA uint8 tensor')
    t = torch.as_tensor(np.asarray(a)) if not torch.is_tensor(a) else a
    assert tuple(t.shape) == tuple(shape), f'mask shape {tuple(t.shape)} != image shape {tuple(shape)}'
    d = (t != 0).to(device=device, dtype=torch.uint8).contiguous()
    if len(_MASKS) > 64:
        _MASKS.clear()
    _MASKS[key] = d
    return d


def tile_err_word(device):
    """The device's persistent error word of the tile-resident launches (include/qtmpnn.h, qt_cheb_tile_fwd): one int32,
    allocated once OUTSIDE any hipGraph memory pool and never zeroed by a mesh build, so it outlives meshes and graph replays."""
    key = str(device)
    w = _TILE_ERR.get(key)
    if w is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('the first mesh with tile structures was built inside a hipGraph capture: run one eager (warm-up) '
                               'step first (make_graphed_step does)')
        w = _TILE_ERR[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return w


class Mesh:
    """labels (B,n,m) i32 | level (B,n,m) u8 | cell (N,4) i32 | npix (N) | posfeat (N,3)
    | rowptr (N+1) / col / nrm CSR of L^ = -D^-1/2 W D^-1/2 | node_off (B+1)."""

    def __init__(self):
        self._ones = {}
        self._E = None
        self.n_dev = None        # device int32[1] with the valid node count when N is a capacity (static mode)
        self.ell = None          # (N, 8) int32: [col x4 | nrm bits x4] of the first four edges of every row
        self.tail_cnt = self.tail_pool = self.tail_info = self.tail_rec = None     # the edges beyond the fourth, per clip (qt_edges_norm)
        self.tiles = None        # frames of several base cells: the per-TILE arrays of qt_edges_norm_tiles (dict) for qt_cheb_tile_*
        self.cell_off = None     # (B * tiles + 1) first node of every 64 x 64 tile in label order (None: no such contiguity)
        self.pixelwise = False   # every unmasked pixel is a node (thresh = -inf); unflatten then NaN-fills the mask
        self.recipe = None       # arguments that rebuild a data-independent mesh for another batch size
        self.loss_mask = None    # (n, m) u8 when the labels do not encode the mask (homogeneous preset mesh): the loss
        self.npix_valid = None   # masks pixels explicitly and counts a node's unmasked pixels
        self.built_from = None   # weak reference to the mesh this one was decomposed from (build_mesh(prev=...)), with
        self.fwd_src = None      # fwd_src (N) / bwd_src (N_old) int32: per single-pixel node the other mesh's node under
        self.bwd_src = None      # its pixel (-1: larger node) -- the state transfer's direct row index (qt_remesh)

    # -- sizes ------------------------------------------------------------------
    @property
    def P(self):
        return self.n * self.m

    @property
    def E(self):
        """Directed non-self edges (lazy: needs a host read)."""
        if self._E is None or self.n_dev is not None:
            self._E = int(self.rowptr[-1].item()) if self.N > 0 else 0
        return self._E

    @property
    def n_valid(self):
        """Valid node count (a host read in static mode)."""
        return self.N if self.n_dev is None else int(self.n_dev.item())

    def __len__(self):
        return self.N

    def for_batch(self, B):
        """The same data-independent mesh (pixelwise / preset static) replicated block-diagonally for B clips."""
        if B == self.B:
            return self
        assert self.recipe is not None, 'only pixelwise and preset static meshes can be re-batched'
        key = ('batch', B)
        if key not in self._ones:
            self._ones[key] = self.recipe(B)
        return self._ones[key]

    # -- per-node data of the attention convolutions --------------------------------
    def attn_geometry(self):
        """(xy (N, 2) centroids in edge-attribute units (graph_functions.py:657), selfpair (N) or None, eattr (E, 2), rev (E)): the
        per-edge [angle, dist] of the stored edges and the position of each edge's transpose (computed once per mesh on the device) and which nodes carry a self pair
        (multi-pixel cells of quadtree meshes, get_adj :329-333; pixelwise meshes have none, get_adj_pixelwise)."""
        if 'geom' not in self._ones:
            # (python scalars travel as kernel arguments: no host-to-device copy, so this also runs inside a graph capture)
            xy = torch.stack([self.posfeat[:, 0] * (self.m * self.resolution), self.posfeat[:, 1] * (self.n * self.resolution)], dim=1)
            ecap = max(int(self.col.numel()), 1)
            eattr = torch.empty(ecap, 2, device=xy.device)
            rev = torch.empty(ecap, dtype=torch.int32, device=xy.device)
            _lib.call('qt_attn_edge_attrs', ptr(self.rowptr), ptr(self.col), ptr(xy), self.N, ptr(self.n_dev), ptr(eattr), ptr(rev))
            self._ones['geom'] = (xy, None if self.pixelwise else (self.npix > 1).float(), eattr, rev)
        return self._ones['geom']

    # -- T_k(L^) 1 for the bias terms of stacked ChebConvs ------------------------
    def cheb_ones(self, ks):
        """(N, 4*ceil(ks/4)) matrix [1, L^1, T_2(L^)1, ... | 0-pad] (fp32; padded so rows are float4 operands)."""
        if ks == 1:                      # [1 0 0 0] does not depend on the mesh: one constant per (rows, device)
            key = (self.N, str(self.labels.device))
            if key not in _ONES1:
                if len(_ONES1) > 8:
                    _ONES1.clear()
                c = torch.zeros(self.N, 4, device=self.labels.device)
                c[:, 0] = 1.0
                _ONES1[key] = c
            return _ONES1[key]
        if ks not in self._ones:
            cols = [torch.ones(self.N, device=self.labels.device)]
            for k in range(1, ks):
                nxt = torch.empty_like(cols[0])
                if k == 1:
                    spmm(self, cols[0], 1.0, None, 0.0, None, 0.0, nxt, 1)
                else:
                    spmm(self, cols[-1], 2.0, cols[-2], -1.0, None, 0.0, nxt, 1)
                cols.append(nxt)
            cols += [torch.zeros_like(cols[0])] * ((-ks) % 4)
            self._ones[ks] = torch.stack(cols, dim=1).contiguous()
        return self._ones[ks]

    # -- reference-compatible views (not on the hot path) ---------------------------
    def edge_index(self, self_loops=True):
        """Sorted (2, E) int64 edge list; with self pairs for multi-pixel cells like get_adj (:329-333)."""
        rp = self.rowptr.long()
        src = torch.repeat_interleave(torch.arange(self.N, device=rp.device), rp[1:] - rp[:-1])
        dst = self.col[: src.numel()].long()
        if self_loops:
            multi = torch.nonzero(self.npix > 1).flatten()
            src, dst = torch.cat([src, multi]), torch.cat([dst, multi])
        order = torch.argsort(src * max(self.N, 1) + dst)
        return torch.stack([src[order], dst[order]])

    def edge_attrs(self, use_edge_attrs=False, resolution=0.25):
        """dist (E,) or [angle, dist] (E, 2) for edge_index(self_loops=True) (graph_functions.py:358-370)."""
        ei = self.edge_index(True)
        xx = self.posfeat[:, 0] * self.m * resolution
        yy = self.posfeat[:, 1] * self.n * resolution
        dx, dy = xx[ei[0]] - xx[ei[1]], yy[ei[0]] - yy[ei[1]]
        d = torch.sqrt(dy ** 2 + dx ** 2)
        if not use_edge_attrs:
            return d
        return torch.stack([torch.atan2(dx, dy) % (2 * math.pi) / (2 * math.pi), d]).T

    def to_dense(self):
        """The reference's dense (N, P) mapping (single clip only; debugging / notebooks)."""
        assert self.B == 1, 'dense mapping is a single-clip notion'
        lab = self.labels.reshape(-1).long()
        mp = torch.zeros(self.N, lab.numel(), device=lab.device)
        ok = lab >= 0
        mp[lab[ok], torch.nonzero(ok).flatten()] = 1.0
        return mp

    def reshape(self, *shape):
        """`mapping.reshape(-1, *image_shape)` of the reference's notebooks (notebooks/create_mesh.ipynb): one image-shaped
        0/1 plane per node, from the dense form."""
        return self.to_dense().reshape(*shape)


def spmm(mesh, x, alpha, p, beta, q, gamma, out, C):
    _lib.call('qt_spmm', ptr(mesh.rowptr), ptr(mesh.col), ptr(mesh.nrm), mesh.N, ptr(mesh.n_dev), C, ptr(x), alpha,
              ptr(p), beta, ptr(q), gamma, ptr(out))


def spmm2(mesh, xs, alpha, ps, beta, qs, gamma, outs):
    """The same product for rows kept as one or two matrices side by side: xs / ps / qs / outs are lists of 1 or 2 operands
    (ps / qs may be None); x / p / q operands may be column views of wider matrices (row-strided), outs are dense."""
    def op(ts, i):
        if ts is None or i >= len(ts) or ts[i] is None:
            return None, 0
        t = ts[i]
        return ptr(t), (t.stride(0) if t.shape[0] > 1 else t.shape[1])
    args = []
    for i in range(2):
        if i < len(xs):
            (x, ldx), (p, ldp), (q, ldq) = op(xs, i), op(ps, i), op(qs, i)
            args += [xs[i].shape[1], x, ldx, p, ldp, q, ldq, ptr(outs[i])]
        else:
            args += [0, None, 0, None, 0, None, 0, None]
    ell = ptr(mesh.ell) if os.environ.get('QT_SPMM_NO_ELL') != '1' else None
    _lib.call('qt_spmm2', ptr(mesh.rowptr), ptr(mesh.col), ptr(mesh.nrm), mesh.N, ptr(mesh.n_dev), *args, alpha, beta, gamma, ell)


def build_mesh(src=None, prev=None, B=1, n=None, m=None, thresh=0.05, condition='max_larger_than', mask=None,
               high_interest_region=None, max_size=64, resolution=0.25, size_norm=None, device=None, static=False, tiles=True):
    """Quadtree-decompose B criterion images and emit the block-diagonal mesh.

    src  : (B, rows, cols) fp32 criterion image (edge-padded on the fly), or
    prev : (nodeval (N_old,) fp32 -- any element stride, e.g. column 0 of a wider matrix --, old Mesh): the
           un-flattened previous output, never materialised.
    static: size every buffer for the worst case N = B*n*m and keep the node count on the device
            (mesh.n_dev): no host sync, fixed shapes -> the whole step can be captured in a hipGraph.
    tiles : frames of several 64 x 64 base cells: also build the per-tile structures of the tile-resident recurrences
            (qt_edges_norm_tiles; ~10 us more per build).  Seq2Seq asks for them only on the meshes whose recurrences can
            take that path (K >= 4: the encoder's stacks of two or more ChebConvs).
    """
    assert condition in CONDITIONS, f'unknown condition {condition}'
    assert max_size & (max_size - 1) == 0, f'max_size / max_grid_size = {max_size}: must be a power of two'
    if src is not None:
        _lib.require_cuda(src, 'criterion image')
        src = src.contiguous().float()
        device = src.device
        B = src.shape[0]
        n = n if n is not None else src.shape[1]
        m = m if m is not None else src.shape[2]
    else:
        nodeval, old = prev
        _lib.require_cuda(nodeval, 'node values')
        nodeval = nodeval.detach().float()
        if nodeval.dim() != 1 or (nodeval.numel() > 1 and nodeval.stride(0) < 1):
            nodeval = nodeval.reshape(-1).contiguous()
        device, B, n, m = nodeval.device, old.B, old.n, old.m
    nbi, nbj = -(n // -max_size), -(m // -max_size)
    if nbi > nbj:
        raise IndexError('padded rows exceed padded columns: the reference split window is empty here '
                         '(model/graph_functions.py:222-229)')
    nbase = nbi * nbj
    mk = _as_u8(mask, device, (n, m))
    hr = _as_u8(high_interest_region, device, (n, m))
    i32 = dict(dtype=torch.int32, device=device)
    local_id = torch.empty(B, n, m, **i32)
    level = torch.empty(B, n, m, dtype=torch.uint8, device=device)
    # 64 x 64 base cells are decomposed by four workgroups each (one per quadrant): leaf counts per quadrant, in DFS order
    quads = int(max_size == 64 and os.environ.get('QT_NO_STAGE1_QUADS') != '1')
    ncnt = B * nbase * (4 if quads else 1)
    cnt = torch.empty(ncnt, **i32)
    offs = torch.empty(ncnt + 1, **i32)
    tmp = torch.empty(ncnt // 1024 + 8, **i32)
    direct = src is None and os.environ.get('QT_NO_DIRECT_SRC') != '1'
    # bwd_src (per OLD node: the new node under its single pixel) is completed by stage 3, which writes only the entries of old
    # nodes whose head pixel carries their label; stage 1 fills it with -1 first, so an entry nobody writes reads "no direct row"
    bwd_src = torch.empty(max(old.N, 1), **i32) if direct else None
    fill_len = bwd_src.numel() if direct else 0
    if src is not None:
        _lib.call('qt_quadtree_stage1', ptr(src), src.shape[1], src.shape[2], None, 0, None, B, n, m, max_size,
                  float(thresh), CONDITIONS.index(condition), ptr(mk), ptr(hr), ptr(local_id), ptr(level), ptr(cnt), quads,
                  None, 0)
    else:
        _lib.call('qt_quadtree_stage1', None, 0, 0, ptr(nodeval), nodeval.stride(0) if nodeval.numel() > 1 else 1,
                  ptr(old.labels), B, n, m, max_size,
                  float(thresh), CONDITIONS.index(condition), ptr(mk), ptr(hr), ptr(local_id), ptr(level), ptr(cnt), quads,
                  ptr(bwd_src), fill_len)
    fused_scan = static and ncnt <= 1024 and os.environ.get('QT_NO_FUSED_SCAN') != '1'   # stage 3 scans the counts itself
    if not fused_scan:
        _lib.call('qt_scan_i32', ptr(cnt), ptr(offs), ncnt, ptr(tmp))
    N = B * n * m if static else int(offs[-1].item())     # dynamic mode: the one host sync of a mesh build

    ms = Mesh()
    ms.B, ms.n, ms.m, ms.N, ms.max_size, ms.resolution = B, n, m, N, max_size, resolution
    ms.mask = mk
    ms.labels = torch.empty(B, n, m, **i32)
    ms.level = level
    ms.cell = torch.empty(max(N, 1), 4, **i32)
    ms.node_off = torch.empty(B + 1, **i32)
    # static mode: rows beyond the valid count stay uninitialised -- every consumer is row local (pinned by
    # tests/test_gpu_ops.py::test_static_mode_ignores_capacity_rows)
    ms.posfeat = torch.empty(N, 3, device=device)
    ms.npix = torch.empty(N, device=device)
    size_norm = size_norm if size_norm is not None else (max_size / 2) ** 2
    old_lab = old_lvl = None
    cell_off = torch.empty(B * nbase + 1, **i32)
    if direct:
        import weakref
        ms.built_from = weakref.ref(old)
        ms.fwd_src = torch.empty(max(N, 1), **i32)         # (stage 3 writes every valid node's entry)
        ms.bwd_src = bwd_src
        old_lab, old_lvl = old.labels, old.level
    _lib.call('qt_quadtree_stage3', ptr(local_id), ptr(cnt if fused_scan else offs), B, n, m, max_size, ptr(ms.labels), ptr(level),
              ptr(ms.cell), ptr(ms.node_off), float(size_norm), ptr(ms.posfeat), ptr(ms.npix), int(fused_scan), quads,
              ptr(old_lab), ptr(old_lvl), ptr(ms.fwd_src), ptr(ms.bwd_src), ptr(cell_off))
    # first node of every 64 x 64 tile (base cell) in label order: the tile-resident transfer stages a tile's source rows by range
    ms.cell_off = cell_off if max_size == 64 else (ms.node_off if n <= 64 and m <= 64 else None)
    nd = None
    if static:
        ms.n_dev = ms.node_off[B:]                # view of the last entry = N
        nd = ptr(ms.n_dev)
    _finish_mesh(ms, device, None, resolution, nd, tiles)
    return ms


def _finish_mesh(ms, device, size_norm, resolution, nd, want_tiles=True):
    """CSR adjacency (and, when size_norm is given, the node features) of a mesh whose labels / level / cell / node_off
    are in place: count -> fill (+ degree) -> normalise, three launches."""
    N, n, m, B = ms.N, ms.n, ms.m, ms.B
    i32 = dict(dtype=torch.int32, device=device)
    if size_norm is not None:
        ms.posfeat = torch.empty(N, 3, device=device)
        ms.npix = torch.empty(N, device=device)
    ms.rowptr = torch.zeros(N + 1, **i32) if N == 0 else torch.empty(N + 1, **i32)   # k_edges_fill writes every entry
    ms.dis = torch.empty(N, device=device)
    if N == 0:
        ms.col = torch.empty(0, **i32)
        ms.w = ms.nrm = torch.empty(0, device=device)
        return
    if size_norm is not None:
        _lib.call('qt_node_features', ptr(ms.cell), N, nd, n, m, float(size_norm), ptr(ms.posfeat), ptr(ms.npix))
    nblk = _lib.value('qt_edges_blocks', N)
    cnt4 = torch.empty(nblk * 1024, **i32)
    sums = torch.empty(nblk + 1, **i32)
    # per-clip pool of the edges beyond a row's fourth (qt_edges_norm fills it; the clip-resident recurrence kernel reads it):
    # only for meshes whose clips fit that kernel's LDS planes (frames up to 64 x 64) -- bigger frames never take it, and
    # their rows' atomic adds on a few per-clip counters (B = 1: one cache line) would only slow qt_edges_norm down
    if n * m <= _lib.value('qt_cheb_clip_rows'):
        tcap = _lib.value('qt_tail_cap')
        ms.tail_cnt = torch.empty(B * 32, **i32)             # (QT_TAIL_CNT_STRIDE ints apart: one cache line per clip's counter)
        ms.tail_pool = torch.empty(B, tcap, 2, **i32)
        ms.tail_info = torch.empty(N, **i32)
        ms.tail_rec = torch.empty(B, 4096, 8, **i32)         # (B, QT_TAIL_REC_CAP, 8): one record per row with more than four edges
    # frames of several 64 x 64 base cells whose tiles are label ranges (quadtree meshes with max_size 64): per-TILE records,
    # pools, boundary records and halo lists for the tile-resident recurrences (csrc/chebclip.hip, TILE = true), plus their sync
    # words; counters and sync words are zeroed by qt_edges_count's launch
    tiles = None
    if (want_tiles and _TILE_CHEB and n * m > _lib.value('qt_cheb_clip_rows') and ms.cell_off is not None and ms.cell_off is not ms.node_off
            and getattr(ms, 'max_size', 0) == 64):
        nbj = -(m // -64)
        T = (-(n // -64)) * nbj
        if T <= 64:
            BT = B * T
            nsync, nx = _lib.value('qt_cheb_tile_sync_words', B), _lib.value('qt_cheb_tile_xbuf_words', B, T)
            cap = lambda w: _lib.value('qt_tile_cap', w)
            # [tile counters | sync words (padded to 16 bytes) | exchange buffer]: one buffer, zeroed by qt_edges_count's launch
            ns4 = (nsync + 3) // 4 * 4
            zbuf = torch.empty(BT * 32 + ns4 + nx, **i32)
            tiles = dict(T=T, nbj=nbj, cnt=zbuf[:BT * 32], sync=zbuf[BT * 32:BT * 32 + nsync], xbuf=zbuf[BT * 32 + ns4:],
                         pool=torch.empty(BT, cap(0), 2, **i32), rec=torch.empty(BT, cap(1), 8, **i32), brec=torch.empty(BT, cap(2), 8, **i32),
                         bpool=torch.empty(BT, cap(3), 2, **i32), halo=torch.empty(BT, cap(2), **i32), baddr=torch.empty(max(N, 1), **i32),
                         zbuf=zbuf)
    ms.tiles = tiles
    if tiles is not None:
        tiles['err'] = tile_err_word(device)
    _lib.call('qt_edges_count', ptr(ms.labels), ptr(ms.cell), N, nd, n, m, ptr(cnt4), ptr(sums), ptr(ms.tail_cnt), B,
              ptr(tiles['zbuf']) if tiles else None, tiles['zbuf'].numel() if tiles else 0)
    emax = 4 * B * n * m                          # every directed edge owns >= 1 of the 4*P pixel adjacencies
    ms.col = torch.empty(emax, **i32)
    ms.w = torch.empty(emax, device=device)
    ms.nrm = torch.empty(emax, device=device)
    _lib.call('qt_edges_fill', ptr(ms.labels), ptr(ms.cell), ptr(cnt4), ptr(sums), N, nd, n, m, float(resolution),
              ptr(ms.rowptr), ptr(ms.col), ptr(ms.w), ptr(ms.dis))
    ms.ell = torch.empty(N, 8, **i32)             # first four edges per row as two 16-byte vectors (k_spmm's fast path)
    if tiles:
        _lib.call('qt_edges_norm_tiles', ptr(ms.rowptr), ptr(ms.col), ptr(ms.w), ptr(ms.dis), N, nd, ptr(ms.nrm), ptr(ms.ell),
                  ptr(ms.cell), ptr(ms.cell_off), tiles['T'], tiles['nbj'], ptr(tiles['cnt']), ptr(tiles['pool']), ptr(tiles['rec']),
                  ptr(tiles['brec']), ptr(tiles['bpool']), ptr(tiles['halo']), ptr(tiles['baddr']), ptr(tiles['err']))
    else:
        _lib.call('qt_edges_norm', ptr(ms.rowptr), ptr(ms.col), ptr(ms.w), ptr(ms.dis), N, nd, ptr(ms.nrm), ptr(ms.ell), ptr(ms.cell),
                  ptr(ms.node_off), ptr(ms.tail_cnt), ptr(ms.tail_pool), ptr(ms.tail_info), ptr(ms.tail_rec))


def tile_error_word(reset=False):
    """OR of the persistent error words of all devices (one 4-byte device read each): bit 0 = a tile-resident launch gave up
    waiting for a neighbour tile, bit 1 = a tile capacity of the mesh build was exceeded.  The words are never reset by the
    library: they report every launch since the process started (or since reset=True was last passed)."""
    err = 0
    for w in _TILE_ERR.values():
        err |= int(w.item())
        if reset:
            w.zero_()
    _TILE_USED.clear()
    return err


def check_tile_errors(always=False):
    """Raise if a tile-resident launch since the last check produced wrong planes (host side of the error word: the trainer
    calls this after every eager training step that issued such a launch, once per epoch, every 64 graph replays and after
    predict()).  Without `always` the device is read only when such a launch was issued since the last check."""
    if not (_TILE_USED or (always and _TILE_ERR)):
        return
    err = tile_error_word(reset=True)
    if err:
        why = []
        if err & 1:
            why.append('a tile waited in vain for a neighbour tile of its clip (the tiles of a launch were not all resident: is '
                       'another process using this GPU?  one process per GPU, or QT_NO_TILE_CHEB=1)')
        if err & 2:
            why.append('a tile capacity of the mesh build was exceeded (QT_TILE_HALO_CAP / QT_TILE_BPOOL_CAP: not a quadtree mesh?)')
        raise RuntimeError('tile-resident Chebyshev launch failed, results since the last check are invalid: ' + '; '.join(why))


def build_homogeneous_mesh(n, m, max_size, mask, B=1, device=None, resolution=0.25):
    """Uniform preset mesh of max_size x max_size cells (clipped at the image border) with the cells that lie entirely under
    `mask` removed and the rest renumbered in the same order (create_static_homogeneous_graph,
    model/graph_functions.py:707-737).  A partly masked cell keeps ALL its pixels -- the reference builds it without the
    mask -- so its labels do not encode the mask: `loss_mask` / `npix_valid` carry it for the masked loss."""
    base = build_mesh(src=torch.zeros(B, n, m, device=device), thresh=float('inf'), mask=None, max_size=max_size,
                      resolution=resolution)
    mk = _as_u8(mask, device, (n, m))
    keep_px = torch.ones(n, m, device=device) if mk is None else (mk == 0).float()
    lab = base.labels.view(B, -1).long()
    cnt = torch.zeros(base.N, device=device).index_add_(0, lab.view(-1), keep_px.view(1, -1).expand(B, -1).reshape(-1))
    kept = cnt > 0
    new_id = torch.cumsum(kept.int(), 0, dtype=torch.int32) - 1
    ms = Mesh()
    ms.B, ms.n, ms.m, ms.max_size, ms.resolution, ms.mask = B, n, m, max_size, resolution, mk
    ms.N = int(kept.sum().item())
    ms.labels = torch.where(kept[lab], new_id[lab], torch.full_like(new_id[lab], -1)).view(B, n, m).contiguous()
    ms.level = base.level
    ms.cell = base.cell[kept].contiguous() if ms.N else base.cell[:1].clone()
    per_clip = torch.zeros(B, dtype=torch.int32, device=device).index_add_(0, base.cell[kept][:, 3].long(),
                                                                           torch.ones(ms.N, dtype=torch.int32, device=device))
    ms.node_off = torch.cat([torch.zeros(1, dtype=torch.int32, device=device), torch.cumsum(per_clip, 0, dtype=torch.int32)])
    ms.cell_off = ms.node_off if n <= 64 and m <= 64 else None
    ms.loss_mask = mk
    ms.npix_valid = cnt[kept].contiguous()
    ms.recipe = lambda b: build_homogeneous_mesh(n, m, max_size, mask, b, device, resolution)
    _finish_mesh(ms, device, (max_size / 2) ** 2, resolution, None)
    return ms


_PIXEL_MESHES = {}


def build_pixel_mesh(B, n, m, mask=None, device=None, resolution=0.25):
    """Every unmasked pixel is a node, raster order, 4-neighbour edges (image_to_graph_pixelwise / get_adj_pixelwise,
    model/graph_functions.py:471-539).  Data independent: built once per (mask, B, shape) and cached.
    The reference passes edge_weight=None (unit weights) here; the CSR carries the uniform pixel distance instead,
    which gives the same L^ because the symmetric normalisation is scale invariant."""
    key = (_mask_key(mask), B, n, m, str(device), float(resolution))
    hit = _PIXEL_MESHES.get(key)
    if hit is not None:
        return hit
    mk = _as_u8(mask, device, (n, m))
    valid = torch.ones(n, m, dtype=torch.bool, device=device) if mk is None else mk == 0
    flat = valid.reshape(-1)
    idx = torch.cumsum(flat.int(), 0, dtype=torch.int32) - 1
    nv = int(flat.sum().item())
    i32 = dict(dtype=torch.int32, device=device)
    offs = torch.arange(B, **i32).view(B, 1) * nv
    ms = Mesh()
    ms.B, ms.n, ms.m, ms.N, ms.max_size, ms.resolution, ms.mask, ms.pixelwise = B, n, m, B * nv, 1, resolution, mk, True
    ms.labels = torch.where(flat, idx, torch.full_like(idx, -1)).view(1, -1).repeat(B, 1)
    ms.labels = torch.where(ms.labels >= 0, ms.labels + offs, ms.labels).view(B, n, m).contiguous()
    ms.level = torch.zeros(B, n, m, dtype=torch.uint8, device=device)
    rc = torch.nonzero(valid).to(torch.int32)                                    # raster order
    cell = torch.cat([rc, torch.ones(nv, 1, **i32), torch.zeros(nv, 1, **i32)], dim=1)
    ms.cell = torch.cat([cell + torch.tensor([0, 0, 0, b], **i32) for b in range(B)]).contiguous()
    ms.node_off = (torch.arange(B + 1, **i32) * nv).contiguous()
    ms.cell_off = ms.node_off if n <= 64 and m <= 64 else None
    ms.recipe = lambda b: build_pixel_mesh(b, n, m, mask, device, resolution)
    _finish_mesh(ms, device, 1.0 / (resolution ** 2), resolution, None)          # size feature = resolution^2 (:521)
    if len(_PIXEL_MESHES) > 16:
        _PIXEL_MESHES.clear()
    _PIXEL_MESHES[key] = ms
    return ms
