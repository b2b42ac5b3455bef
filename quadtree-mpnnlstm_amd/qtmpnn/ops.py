"""torch.autograd.Function wrappers pairing the forward / backward HIP kernels.

Every op runs on the GPU through libqtmpnn_hip.so; there is no CPU fallback.
"""
import os

import torch
from torch.autograd import Function

from . import _lib
from . import mesh as _mesh
from ._lib import ptr
from .mesh import spmm, spmm2

ACT_NONE, ACT_RELU, ACT_TANH_RES, ACT_RELU_BWD = 0, 1, 2, 3
_HEAD_DGRAD = os.environ.get('QT_NO_HEAD_DGRAD') != '1'    # (A/B switch: 1 = the head's two backward products as two launches)
_HEAD_FUSE = os.environ.get('QT_NO_HEAD_FUSE') != '1'      # (A/B switch: 1 = the decoder head's two products as two launches each way)

# Arithmetic of the backward pass's gate-GEMM data gradient (gG W^T).  False (default): exact fp32 products on the fp32 MFMA,
# like every other product of the path and like the reference (model/model.py:394-463 is fp32 throughout).  True: the opt-in
# 2-term split-bf16 product (hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate; ~16 significand bits per
# operand) -- gradients only, measured beside the exact path by bench.py (`split_bf16_dgrad`) and kept alive by
# tests/test_gpu_headline.py.  Read when a pass packs its weights, so it can be switched between passes (not inside a
# captured step: a hipGraph replays what it captured).
DGRAD_SPLIT_BF16 = os.environ.get('QT_DGRAD_SPLIT_BF16') == '1'


def set_dgrad_split_bf16(on):
    global DGRAD_SPLIT_BF16
    prev, DGRAD_SPLIT_BF16 = DGRAD_SPLIT_BF16, bool(on)
    return prev


def _c(t):
    return t if t is None or t.is_contiguous() else t.contiguous()


def _row_stride(t):
    """Row stride of a 2-D operand in floats (0 for None); a 1-row view may carry any stride(0)."""
    if t is None:
        return 0
    return t.stride(0) if t.shape[0] > 1 else t.shape[1]


def _rows(t):
    """(tensor, row stride) for a 2-D fp32 operand whose rows are dense and 16-byte aligned -- column slices of a
    wider matrix (views produced by split / cat backward) are passed to the kernels as they are, without a copy."""
    if t is None:
        return None, 0
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.stride(0) >= t.shape[1] and t.data_ptr() % 16 == 0:
        return t, t.stride(0)
    t = t.contiguous()
    return t, t.shape[1]


class _Unalias(Function):
    """Identity whose backward hands out a private copy of the gradient.

    `a + b` gives both operands the SAME gradient tensor; when both are (views of) leaf parameters autograd may
    let two .grad fields alias one buffer, and the in-place clip_grad_norm_ then scales that buffer twice."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.clone()


def unalias(x):
    return _Unalias.apply(x)


class GradAcc:
    """Sums the parameter-gradient partials of every use of one packed weight inside ONE forward pass.

    A weight is used once per rollout step; instead of one reduction + one autograd add per use, every use adds its
    per-block partial sums into a shared slab buffer (fixed order: each block owns its slab) and only the FIRST
    forward use -- the last one to run in the backward pass, because the recurrent state chains the uses --
    reduces the slabs and hands the gradient to autograd.
    """
    __slots__ = ('uses', 'done', 'part', 'wt', 'pending', 'wslab', 'pslab')

    def __init__(self):
        self.uses, self.done, self.part, self.wt, self.pending, self.wslab = 0, 0, None, {}, [], None
        self.pslab = {}        # segment -> (blocks, G, cin + 4, co) slabs of the one-pass projection backward (qt_proj_bwd)

    def enter(self):
        self.uses += 1
        return self.uses - 1

    def slab(self, like, nblk, width):
        if self.part is None:
            self.part = like.new_zeros(nblk, width)
        return self.part

    def weight_slab(self, like, rows, cols):
        """(qt_lstm_fused_blocks(), rows, cols) zeros: the fused cell backward (qt_lstm_bwd_fused) adds every use's partial weight
        gradient of workgroup b into slab b; the pass's last backward sums the slabs."""
        if self.wslab is None:
            self.wslab = like.new_zeros(_lib.value('qt_lstm_fused_blocks'), rows, cols)
        return self.wslab

    def leave(self, use_idx):
        """True when this use must emit the summed gradient."""
        self.done += 1
        if use_idx != 0:
            return False
        if self.done != self.uses:
            raise RuntimeError('GradAcc: the first use of a packed weight ran its backward before a later use; '
                               'the uses are not chained by the recurrent state')
        return True


# ------------------------------------------------------------------------------ parameter packing
class PackPlan:
    """Gathers the many small parameter tensors of a module (238 in the reference model) into its packed weight matrices
    with a handful of launches, forward and backward.

    `layout(T, fill)` receives stand-ins for the parameters -- index tensors of the parameters' shapes -- and returns
    {name: tensor | (tensor, tensor)} built ONLY with data-movement ops (stack / cat / permute / reshape / transpose / pad
    with `fill`); a tuple means the sum of its two members.  The plan records, for every output element, which parameter
    elements it is made of; the forward is then `flat[src0] + flat[src1]` on the concatenated parameters and the backward
    the transposed gather, whose result is handed to autograd as per-parameter views (no stack / unbind / slice
    backward / AccumulateGrad copies: ~200 tiny kernels per training step with per-tensor torch ops)."""

    def __init__(self, params, layout, flat=None):
        self.params = list(params)
        # flat: qtmpnn.flat.FlatParams whose parameter list is exactly `params` -- the gather then reads the shared buffer
        # (no concatenation) and its backward's gradient vector is the model's flat gradient
        self.flat = flat
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        n_flat = offs[-1]
        T = [torch.arange(o, o + n, dtype=torch.int64).view(p.shape) for o, n, p in zip(offs, sizes, self.params)]
        outs = layout(T, -1)
        self.names, self.shapes, self.out_offs = [], [], [0]
        src = []
        for name, v in outs.items():
            a, b = v if isinstance(v, tuple) else (v, None)
            self.names.append(name)
            self.shapes.append(tuple(a.shape))
            sa = a.reshape(-1)
            sb = b.reshape(-1) if b is not None else torch.full_like(sa, -1)
            src.append(torch.stack([sa, sb], dim=1))
            self.out_offs.append(self.out_offs[-1] + sa.numel())
        src = torch.cat(src)                                  # (n_out, 2), -1 = zero
        n_out = src.shape[0]
        inv = torch.full((n_flat, 2), -1, dtype=torch.int64)
        fill_count = torch.zeros(n_flat, dtype=torch.int64)
        pos = torch.arange(n_out, dtype=torch.int64)
        for c in range(2):
            col = src[:, c]
            ok = col >= 0
            for j, i in zip(col[ok].tolist(), pos[ok].tolist()):
                k = int(fill_count[j])
                assert k < 2, 'a parameter element may feed at most two packed elements'
                inv[j, k] = i
                fill_count[j] = k + 1
        self.two_src = bool((src[:, 1] >= 0).any())
        self.two_inv = bool((inv[:, 1] >= 0).any())
        self.src = torch.where(src >= 0, src, torch.full_like(src, n_flat)).t().contiguous().to(dev)       # zero slot
        self.inv = torch.where(inv >= 0, inv, torch.full_like(inv, n_out)).t().contiguous().to(dev)
        self.offs, self.n_flat, self.n_out, self.device = offs, n_flat, n_out, dev
        self.zero1 = torch.zeros(1, device=dev)

    def same_params(self, params):
        return len(params) == len(self.params) and all(a is b for a, b in zip(params, self.params))

    def __call__(self):
        outs = _PackGather.apply(self, *self.params)
        return dict(zip(self.names, outs))


class _PackGather(Function):
    @staticmethod
    def forward(ctx, plan, *params):
        if plan.flat is not None and plan.flat.intact():
            flat = plan.flat.buffer                 # parameters are views of it, element n_flat is the zero slot
        else:
            flat = torch.cat([p.reshape(-1) for p in params] + [plan.zero1])
        out = flat[plan.src[0]]
        if plan.two_src:
            out = out + flat[plan.src[1]]
        ctx.plan = plan
        ctx.set_materialize_grads(False)
        return tuple(out[a:b].view(shp) for a, b, shp in zip(plan.out_offs[:-1], plan.out_offs[1:], plan.shapes))

    @staticmethod
    def backward(ctx, *gs):
        plan = ctx.plan
        parts = []
        for g, a, b in zip(gs, plan.out_offs[:-1], plan.out_offs[1:]):
            parts.append(g.reshape(-1) if g is not None else plan.zero1.expand(b - a))
        gcat = torch.cat(parts + [plan.zero1])
        gflat = gcat[plan.inv[0]]
        if plan.two_inv:
            gflat = gflat + gcat[plan.inv[1]]
        if plan.flat is not None:
            plan.flat.last_grad = gflat
        return (None, *[gflat[a:b].view(p.shape) for a, b, p in zip(plan.offs[:-1], plan.offs[1:], plan.params)])


# ------------------------------------------------------------------------------ column concatenation
class _ConcatCols(Function):
    """[t_0 | t_1 | ...] for 2-D fp32 node matrices (row-strided views welcome); backward = column views, no copy."""

    @staticmethod
    def forward(ctx, mesh, *ts):
        import ctypes
        ops_ = [_rows(t.float()) for t in ts]
        N = ts[0].shape[0]
        widths = [t.shape[1] for t in ts]
        out = ts[0].new_empty(N, sum(widths))
        n = len(ts)
        srcs = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _ in ops_])
        w = (ctypes.c_int * n)(*widths)
        lds = (ctypes.c_int * n)(*[ld for _, ld in ops_])
        _lib.call('qt_concat', srcs, w, lds, n, N, ptr(mesh.n_dev) if mesh is not None else None, ptr(out))
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, g):
        outs, o = [], 0
        for w in ctx.widths:
            outs.append(g[:, o:o + w])
            o += w
        return (None, *outs)


def concat_cols(tensors, mesh=None):
    """Column concatenation on the custom kernel when every width is a multiple of 4 (else torch.cat)."""
    if len(tensors) <= 8 and all(t.dim() == 2 and t.shape[1] % 4 == 0 and t.is_cuda for t in tensors):
        return _ConcatCols.apply(mesh, *tensors)
    return torch.cat(tensors, dim=1)


# ------------------------------------------------------------------------------ ChebConv stacks
# A node-feature operand Z is a list of one or two matrices side by side, e.g. [X (N, 4), H (N, 16)] for Z = [X | H] of a
# recurrent cell: the parts are never concatenated, and each may be a column view of a wider matrix (row-strided).
def _zparts(Za, Zb):
    parts = [_rows(Za.float())]
    if Zb is not None:
        parts.append(_rows(Zb.float()))
    return [t for t, _ in parts]


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else t.shape[1]


def _plane_args(Zs, TZs):
    """(a0, lda0, a_rest, a0b, lda0b, a_restb) of the C ABI."""
    a = [ptr(Zs[0]), _ld(Zs[0]), ptr(TZs[0])]
    if len(Zs) > 1:
        return a + [ptr(Zs[1]), _ld(Zs[1]), ptr(TZs[1])]
    return a + [None, 0, None]


_CLIP_CHEB = os.environ.get('QT_NO_CLIP_CHEB') != '1'      # (A/B switch: 1 = one k_spmm launch per hop everywhere)
# Recurrences of at least this many planes take the clip-resident launch.  Its fixed cost (~4 us: one memory phase that fills the
# registers and LDS) is paid once per launch and a hop costs ~4 us against ~7.4 us for a k_spmm launch at the bench shape
# (tools/exp_clip.py): K = 5 (stacks of two ChebConvs) 21.4 vs 29.6 us forward, 21.3 vs 31.4 us backward; K = 3 13.4 vs 14.6 and
# 14.3 vs 15.4; K = 2 (one hop) 8.4 vs 6.3: a single hop stays on k_spmm.
_CLIP_MIN_K = int(os.environ.get('QT_CLIP_MIN_K', '3'))
_CLIP_ROWS = []


def _CLIP_ROWS_OF():
    if not _CLIP_ROWS:
        _CLIP_ROWS.append(_lib.value('qt_cheb_clip_rows'))
    return _CLIP_ROWS[0]


def _clip_resident(mesh, widths, K):
    """True when the K - 1 hops of a recurrence run as ONE clip-resident launch (csrc/chebclip.hip): every clip of the block-
    diagonal mesh fits the kernel's LDS planes (n x m <= 4096 nodes), rows are float4 slices, the ELL side array exists -- or
    as tile-resident launches on a frame of several base cells (_tile_resident).  Either way the planes leave slice-major."""
    _CLIP_ROWS_OF()
    if _tile_resident(mesh, widths, K):
        return True
    return (_CLIP_CHEB and K >= max(_CLIP_MIN_K, 2) and mesh.N > 0 and mesh.ell is not None and mesh.tail_rec is not None and mesh.n * mesh.m <= _CLIP_ROWS[0]
            and len(widths) <= 2 and all(w % 4 == 0 for w in widths) and getattr(mesh, 'node_off', None) is not None)


_TILE_MIN_K = int(os.environ.get('QT_TILE_MIN_K', '4'))
_NUM_CUS = []


def _tile_resident(mesh, widths, K):
    """True when the K - 1 hops of a recurrence on a frame of SEVERAL 64 x 64 base cells run as ONE tile-resident launch
    (csrc/chebclip.hip, TILE = true): one workgroup per (clip, tile, 4-channel slice), the rows on tile borders exchanged between
    the tiles of a clip after every hop.  Taken where it was measured faster than one k_spmm launch per hop (tools/exp_tile.py):
    at least three hops (a hop costs ~4.3 us here -- a cross-CU hand-off -- against 7 - 8 us per k_spmm launch, after a ~5 us
    prologue) and all workgroups resident in one round (8 clips x 4 tiles x 5 .. 8 slices; 640 workgroups -- hidden 32 on 16
    clips -- would take three rounds and lose)."""
    tl = getattr(mesh, 'tiles', None)
    if not (_CLIP_CHEB and tl is not None and K >= max(_TILE_MIN_K, 2) and K <= 16 and mesh.N > 0 and mesh.ell is not None
            and len(widths) <= 2 and all(w % 4 == 0 for w in widths) and sum(widths) <= 4 * _lib.value('qt_tile_cap', 4)):
        return False
    if not _NUM_CUS:
        _NUM_CUS.append(_lib.value('qt_num_cus'))
    if _shared_device():
        return False
    return mesh.B * tl['T'] * (sum(widths) // 4) <= _NUM_CUS[0]


def _shared_device():
    """True when several ranks of this job run on ONE GPU (more local ranks than devices -- bench.py's QT_DIST_BACKEND=gloo
    rehearsal mode, the two-rank test on a one-GPU box -- or QT_SHARED_GPU=1): the tile-resident launches need all tiles of a
    launch co-resident, which two processes issuing them side by side cannot promise, so those runs stay on one k_spmm launch
    per hop.  (QT_SHARED_GPU=0 overrides the rank count.)"""
    flag = os.environ.get('QT_SHARED_GPU')
    if flag is not None:
        return flag == '1'
    try:
        return int(os.environ.get('LOCAL_WORLD_SIZE', '1')) > max(torch.cuda.device_count(), 1)
    except ValueError:
        return False


def _tile_args(mesh):
    tl = mesh.tiles
    return (ptr(mesh.rowptr), ptr(mesh.col), ptr(mesh.nrm), ptr(mesh.ell), ptr(mesh.cell_off), ptr(tl['cnt']), ptr(tl['pool']),
            ptr(tl['rec']), ptr(tl['brec']), ptr(tl['bpool']), ptr(tl['halo']), ptr(tl['baddr']), ptr(tl['xbuf']), ptr(tl['sync']),
            ptr(tl['err']), mesh.B, tl['T'], tl['nbj'])


def clip_planes(mesh, Zs, TZs, K, width=0):
    """TZs[i] <- T_1 .. T_{K-1} of the recurrence on the parts Zs, all hops in ONE launch (qt_cheb_clip_fwd).  The planes are
    written SLICE-major: TZs[i] (allocated (K - 1, N, C_i)) then holds (K - 1, C_i / 4, N, 4).  width: channels per workgroup
    (0 = the library's choice; 2 / 4 pin it: diagnostics and parity tests)."""
    two = len(Zs) > 1
    if getattr(mesh, 'tiles', None) is not None and mesh.n * mesh.m > _CLIP_ROWS_OF():
        _mesh._TILE_USED.add(str(Zs[0].device))
        _lib.call('qt_cheb_tile_fwd', *_tile_args(mesh), Zs[0].shape[0], K, Zs[0].shape[1], ptr(Zs[0]), _ld(Zs[0]), ptr(TZs[0]),
                  Zs[1].shape[1] if two else 0, ptr(Zs[1]) if two else None, _ld(Zs[1]) if two else 0, ptr(TZs[1]) if two else None)
        return
    _lib.call('qt_cheb_clip_fwd', ptr(mesh.rowptr), ptr(mesh.col), ptr(mesh.nrm), ptr(mesh.ell), ptr(mesh.node_off),
              ptr(mesh.tail_cnt), ptr(mesh.tail_pool), ptr(mesh.tail_rec), mesh.B, Zs[0].shape[0], K, Zs[0].shape[1], ptr(Zs[0]), _ld(Zs[0]), ptr(TZs[0]),
              Zs[1].shape[1] if two else 0, ptr(Zs[1]) if two else None, _ld(Zs[1]) if two else 0, ptr(TZs[1]) if two else None, int(width))


def clip_clenshaw(mesh, Gs, K, sm=0, width=0):
    """Gs[i] (K, N, C_i) gradient planes: plane 0 <- A_0 + L^ b_1 - b_2 (Clenshaw), all hops in ONE launch (qt_cheb_clip_bwd).
    sm: planes 1 .. K-1 are stored slice-major (written so by the data-gradient kernels on request)."""
    two = len(Gs) > 1
    if getattr(mesh, 'tiles', None) is not None and mesh.n * mesh.m > _CLIP_ROWS_OF():
        _mesh._TILE_USED.add(str(Gs[0].device))
        _lib.call('qt_cheb_tile_bwd', *_tile_args(mesh), Gs[0].shape[1], K, Gs[0].shape[2], ptr(Gs[0]),
                  Gs[1].shape[2] if two else 0, ptr(Gs[1]) if two else None, int(sm))
        return
    _lib.call('qt_cheb_clip_bwd', ptr(mesh.rowptr), ptr(mesh.col), ptr(mesh.nrm), ptr(mesh.ell), ptr(mesh.node_off),
              ptr(mesh.tail_cnt), ptr(mesh.tail_pool), ptr(mesh.tail_rec), mesh.B, Gs[0].shape[1], K, Gs[0].shape[2], ptr(Gs[0]),
              Gs[1].shape[2] if two else 0, ptr(Gs[1]) if two else None, int(sm), int(width))


def _cheb_planes(Zs, mesh, K):
    """T_1 .. T_{K-1} of the Chebyshev recurrence on Z (T_0 = Z itself): per part (max(K-1, 1), N, C)."""
    N = Zs[0].shape[0]
    TZs = [Z.new_empty(max(K - 1, 1), N, Z.shape[1]) for Z in Zs]
    if _clip_resident(mesh, [Z.shape[1] for Z in Zs], K):
        clip_planes(mesh, Zs, TZs, K)          # (these planes are stored slice-major: see planes_rowmajor)
        return TZs, 1
    for k in range(1, K):
        if k == 1:
            spmm2(mesh, Zs, 1.0, None, 0.0, None, 0.0, [T[0] for T in TZs])
        else:
            spmm2(mesh, [T[k - 2] for T in TZs], 2.0, Zs if k == 2 else [T[k - 3] for T in TZs], -1.0, None, 0.0,
                  [T[k - 1] for T in TZs])
    return TZs, 0


def planes_rowmajor(T, sm):
    """(K - 1, N, C) row-major view / copy of a plane tensor of _cheb_planes (sm: stored as (K - 1, C / 4, N, 4))."""
    if not sm:
        return T
    Km, N, C = T.shape
    return T.view(Km, C // 4, N, 4).permute(0, 2, 1, 3).reshape(Km, N, C)


def _w_t(W, acc):
    """W^T for the forward GEMM's straight-copy staging; one transpose per pass when the weight is shared (acc), none
    for a single use (the kernel then transposes while staging)."""
    if acc is None or W.shape[1] <= 16:
        return None
    Wt = acc.wt.get('T')
    if Wt is None:
        Wt = acc.wt['T'] = W.t().contiguous()
    return Wt


def _dgrad_weight(W, K, Cs, live, acc):
    """(Wb, skinny): the rows of W that multiply the column parts `live` of [T_0 .. T_{K-1}] -- the data gradient is
    G @ Wb^T, and the GEMM stages (Wb^T)^T = Wb's own rows: no transposed copy at all (except for <= 16 output columns,
    which the skinny VALU kernel reads as a plain (Co, K Cl) matrix: then Wb is that transpose)."""
    C, Co = sum(Cs), W.shape[1]
    Cl = [Cs[i] for i in live]
    skinny = K * sum(Cl) <= 16
    key = f'wb{live[0]}{len(live)}{int(skinny)}'
    Wb = acc.wt.get(key) if acc is not None else None
    if Wb is None:
        Wb = W[:K * C]
        if len(live) < len(Cs):
            lo = sum(Cs[:live[0]])
            Wb = Wb.view(K, C, Co)[:, lo:lo + Cs[live[0]]].reshape(-1, Co)     # shared by every use in this pass
        if skinny:
            Wb = Wb.t().contiguous()
        elif acc is not None and Wb.is_cuda and DGRAD_SPLIT_BF16:
            # opt-in: two bf16 terms of the same rows for the split-bf16 data gradient of qt_lstm_bwd_dgrad (once per pass)
            Wb = Wb.contiguous()
            hi, lo = (torch.empty(Wb.shape, dtype=torch.bfloat16, device=Wb.device) for _ in range(2))
            _lib.call('qt_split_bf16', ptr(Wb), Wb.numel(), ptr(hi), ptr(lo))
            Wb._qt_split = (hi, lo)
        if acc is not None:
            acc.wt[key] = Wb
    return Wb, skinny


def _cheb_backward(Zs, TZs, W, G, mesh, K, Ks, acc, use_idx, need_gZ, need_gW, gTs_pre=None, w_fused=False, sm=0):
    """(gZ parts, gW) of Y = [T_0 .. T_{K-1} | S] W from G = dL/dY (N, Co).  w_fused: this use's weight-gradient partials are
    already in acc.wslab (qt_lstm_bwd_fused; G is then None)."""
    N = Zs[0].shape[0]
    Cs = [Z.shape[1] for Z in Zs]
    C = sum(Cs)
    Co = W.shape[1]
    gZs = None
    need = list(need_gZ) if isinstance(need_gZ, (list, tuple)) else [bool(need_gZ)] * len(Zs)
    need_gZ = any(need)
    if need_gZ and N > 0:
        # only the column parts whose input wants a gradient are propagated (the encoder's X is data: the K-1 Clenshaw
        # launches of its cells then carry H's 16 channels alone, and the data-gradient GEMM is narrower)
        live = [i for i, f in enumerate(need) if f]
        Cl = [Cs[i] for i in live]
        clip = _clip_resident(mesh, Cl, K)
        gsm = 0                                # the gradient planes 1.. leave the GEMM slice-major when the fused Clenshaw reads them
        if gTs_pre is not None:
            gTs, gsm = gTs_pre                 # already computed together with G (qt_lstm_bwd_dgrad)
        else:
            Wb, skinny = _dgrad_weight(W, K, Cs, live, acc)
            gTs = [Zs[0].new_empty(K, N, c) for c in Cl]
            split = Wb.__dict__.get('_qt_split') if hasattr(Wb, '__dict__') else None
            if split is not None and Co >= 64 and Co % 16 == 0 and K * sum(Cl) >= 64:
                # wide gate matrices (hidden 32: 128 columns in, K C >= 64 out): the fp32-MFMA rate bounds the exact product (narrow
                # products are memory-bound and stay exact); split-bf16, gradients only
                _lib.call('qt_dense_sb', ptr(G), 0, Co, ptr(split[0]), ptr(split[1]), K, Cl[0], Cl[1] if len(Cl) > 1 else 0, N,
                          ptr(mesh.n_dev), ptr(gTs[0]), ptr(gTs[1]) if len(Cl) > 1 else None)
            else:
                gsm = int(clip)
                _lib.call('qt_dense2', ptr(G), 0, None, None, 0, None, 1, Co, 0, ptr(Wb) if skinny else None,
                          None if skinny else ptr(Wb), None, 0, None, K, Cl[0],
                          Cl[1] if len(Cl) > 1 else 0, N, ptr(mesh.n_dev), ACT_NONE, None, 0, None, ptr(gTs[0]),
                          ptr(gTs[1]) if len(Cl) > 1 else None, 2 * gsm, None, None)
        # Clenshaw: b_k = A_k + 2 L^ b_{k+1} - b_{k+2}, in place;  gZ = A_0 + L^ b_1 - b_2
        if clip:                                # all hops in one launch; only plane 0 (= gZ) is rewritten
            clip_clenshaw(mesh, gTs, K, gsm)
            K_hops = 0
        else:
            K_hops = K
        for k in range(K_hops - 2, 0, -1):
            spmm2(mesh, [g[k + 1] for g in gTs], 2.0, [g[k] for g in gTs], 1.0,
                  [g[k + 2] for g in gTs] if k + 2 < K else None, -1.0, [g[k] for g in gTs])
        if K_hops > 1:
            spmm2(mesh, [g[1] for g in gTs], 1.0, [g[0] for g in gTs], 1.0, [g[2] for g in gTs] if K > 2 else None, -1.0,
                  [g[0] for g in gTs])
        gZs = [None] * len(Zs)
        for i, g in zip(live, gTs):
            gZs[i] = g[0]
    elif need_gZ:
        gZs = [torch.zeros_like(Z) if f else None for Z, f in zip(Zs, need)]
    gW = None
    if need_gW:
        ksp = (Ks + 3) // 4 * 4
        S = mesh.cheb_ones(Ks) if Ks else None
        if acc is None:
            gW = torch.empty_like(W) if N > 0 else torch.zeros_like(W)
            if N > 0:
                nblk = _lib.value('qt_wgrad_blocks', N)
                part = Zs[0].new_empty(nblk, W.shape[0], Co)
                _lib.call('qt_wgrad', *_plane_args(Zs, TZs), K, Cs[0], Cs[1] if len(Cs) > 1 else 0, ptr(S), ksp, ptr(G), Co, N,
                          ptr(mesh.n_dev), 0, ptr(part), sm)
                _lib.call('qt_colsum', ptr(part), nblk, W.numel(), ptr(gW))
        else:
            # deferred: the weight gradient is off the critical path, so all uses of this pass (<= 16 per launch) are
            # reduced together by the last backward -- one long launch instead of one short one per rollout step
            if not w_fused:
                acc.pending.append((Zs, TZs, S, G, N, mesh.n_dev, sm))
            if acc.leave(use_idx):
                gW = _wgrad_group(acc.pending, W, K, Cs, ksp, Co) if acc.pending else None
                acc.pending = []
                if acc.wslab is not None:          # + the partials the fused launches added, slab by slab
                    gf = torch.empty_like(W)
                    _lib.call('qt_colsum', ptr(acc.wslab), acc.wslab.shape[0], W.numel(), ptr(gf))
                    gW = gf if gW is None else gW + gf
                    acc.wslab = None
    return gZs, gW


class _ChebPoly(Function):
    """Y = act( sum_k T_k(L^) Z M_k + S Bm ),  W = [M_0; ...; M_{K-1}; Bm]  ((K*C + Ks), Co),  Z = [Za | Zb].

    One call evaluates what the reference spreads over many ChebConv modules
    (model/model.py:59-97, :394-424): the Chebyshev recurrence runs once on Z = [X | H] for
    all four gates, stacked ChebConv layers are pre-composed into one degree-2L polynomial
    (compose_chebconvs below), and their inner biases become the S = T_k(L^)1 columns.
    """

    @staticmethod
    def forward(ctx, Za, Zb, W, res, drop, mesh, K, Ks, act, acc, W2=None, acc2=None):
        """W2 (Co + 4, 4), Co = 16: a second product in the same launch, U = [Y | 1 0 0 0] @ W2 (an ordinary K = 1 cheb_poly
        with one bias row on the 16 output channels: the decoder head's fc_out1 -> coefficient columns of fc_out2); returns
        (Y, U).  Its backward is fused the other way round: gU @ W2^T and Y's ReLU gradient in one launch."""
        _lib.require_cuda(Za, 'node features')
        Zs, W = _zparts(Za, Zb), _c(W.float())
        N = Zs[0].shape[0]
        Cs = [Z.shape[1] for Z in Zs]
        Co = W.shape[1]
        ksp = (Ks + 3) // 4 * 4
        assert W.shape[0] == K * sum(Cs) + ksp, f'weight rows {W.shape[0]} != {K}*{sum(Cs)}+{ksp}'
        TZs, sm = _cheb_planes(Zs, mesh, K)
        S = mesh.cheb_ones(Ks) if Ks else None
        Y = Zs[0].new_empty(N, Co)
        drop = _c(drop)
        U = None
        if W2 is not None:
            W2 = _c(W2.float())
            assert Co == 16 and W2.shape == (Co + 4, 4) and act in (ACT_NONE, ACT_RELU), (Co, W2.shape, act)
            U = Y.new_empty(N, 4)
        _lib.call('qt_dense2', *_plane_args(Zs, TZs), K, Cs[0], Cs[1] if len(Cs) > 1 else 0, ptr(W),
                  None if W2 is not None else ptr(_w_t(W, acc)), ptr(S), ksp,
                  ptr(W[K * sum(Cs):]) if Ks else None, 1, Co, 0, N, ptr(mesh.n_dev), act, ptr(res), _row_stride(res), ptr(drop),
                  ptr(Y), None, sm, ptr(W2), ptr(U))
        ctx.mesh, ctx.K, ctx.Ks, ctx.act, ctx.acc, ctx.nz, ctx.sm = mesh, K, Ks, act, acc, len(Zs), sm
        ctx.use_idx = acc.enter() if acc is not None else 0
        ctx.acc2 = acc2
        ctx.use_idx2 = acc2.enter() if (W2 is not None and acc2 is not None) else 0
        ctx.save_for_backward(*Zs, *TZs, W, Y if (act != ACT_NONE or W2 is not None) else None, res, drop, W2)
        if W2 is not None:
            ctx.set_materialize_grads(False)
            return Y, U
        return Y

    @staticmethod
    def backward(ctx, gY, gU=None):
        nz = ctx.nz
        saved = ctx.saved_tensors
        Zs, TZs = list(saved[:nz]), list(saved[nz:2 * nz])
        W, Y, res, drop, W2 = saved[2 * nz:]
        mesh, K, Ks, act = ctx.mesh, ctx.K, ctx.Ks, ctx.act
        N = Zs[0].shape[0]
        Co = W.shape[1]
        gW2 = None
        if W2 is not None:
            # the second product's backward: its weight gradient is deferred like any other (Z = Y, one plane), its data gradient
            # gU @ W2[:Co]^T lands in this op's own incoming gradient -- with Y's ReLU gradient applied by the same launch
            if gU is None:
                gU = Y.new_zeros(N, 4)
            gU = _c(gU.float())
            _, gW2 = _cheb_backward([Y], [Y.new_empty(1, 0, Co)], W2, gU, mesh, 1, 1, ctx.acc2, ctx.use_idx2, False,
                                    ctx.needs_input_grad[10])
            Wb2, _ = _dgrad_weight(W2, 1, [Co], [0], ctx.acc2)               # (4, Co): W2[:Co]^T
            G = Y.new_empty(N, Co)
            fuse = act == ACT_RELU and gY is None
            need = list(ctx.needs_input_grad[:nz])
            if _HEAD_DGRAD and fuse and N > 0 and all(need) and Co == 16:
                # both backward products of the head in ONE launch, one lane per node row (qt_head_dgrad): G = relu'(Y) (gU Wb2)
                # stays in registers between them (and is stored for the deferred weight gradient)
                Cs = [Z.shape[1] for Z in Zs]
                Wb1, sk1 = _dgrad_weight(W, K, Cs, list(range(nz)), ctx.acc)
                assert not sk1 and Wb1.is_contiguous()
                gsm = int(_clip_resident(mesh, Cs, K))
                gTs = [Y.new_empty(K, N, c) for c in Cs]
                _lib.call('qt_head_dgrad', ptr(gU), ptr(Wb2), ptr(Y), ptr(Wb1), K, Cs[0], Cs[1] if nz > 1 else 0, N, ptr(mesh.n_dev),
                          ptr(G), ptr(gTs[0]), ptr(gTs[1]) if nz > 1 else None, gsm)
                gZs, gW = _cheb_backward(Zs, TZs, W, G, mesh, K, Ks, ctx.acc, ctx.use_idx, need, ctx.needs_input_grad[2],
                                         gTs_pre=(gTs, gsm), sm=ctx.sm)
                return gZs[0], (gZs[1] if nz > 1 else None), gW, None, None, None, None, None, None, None, gW2, None
            if N > 0:
                _lib.call('qt_dense2', ptr(gU), 0, None, None, 0, None, 1, 4, 0, ptr(Wb2), None, None, 0, None, 1, Co, 0, N,
                          ptr(mesh.n_dev), ACT_RELU_BWD if fuse else ACT_NONE, ptr(Y) if fuse else None, Co if fuse else 0, None,
                          ptr(G), None, 0, None, None)
            if fuse:
                act = ACT_NONE                      # (done)
            elif gY is not None:
                G = G + gY.float()
            gY = G
        G = _c(gY.float())
        gres = None
        if act != ACT_NONE:
            gin = G
            G = torch.empty_like(gin)
            if act == ACT_TANH_RES and ctx.needs_input_grad[3]:
                assert res.is_contiguous() and res.shape[1] <= Co
                gres = torch.empty_like(res)
            if N > 0:
                _lib.call('qt_act_bwd', ptr(gin), ptr(Y), ptr(res), _row_stride(res), ptr(drop), act, N,
                          ptr(mesh.n_dev), Co, ptr(G), ptr(gres), None)
        gZs, gW = _cheb_backward(Zs, TZs, W, G, mesh, K, Ks, ctx.acc, ctx.use_idx,
                                 list(ctx.needs_input_grad[:nz]), ctx.needs_input_grad[2], sm=ctx.sm)
        gZa = gZs[0] if gZs is not None else None
        gZb = gZs[1] if gZs is not None and nz > 1 else None
        return gZa, gZb, gW, gres, None, None, None, None, None, None, gW2, None


def _wgrad_group(uses, W, K, Cs, ksp, Co):
    import ctypes
    uses = [u for u in uses if u[4] > 0]
    if not uses:
        return torch.zeros_like(W)
    sm = uses[0][6]
    assert all(u[6] == sm for u in uses), 'the uses of one weight must share the layout of their planes'
    two = len(Cs) > 1
    chunks = [uses[i:i + 16] for i in range(0, len(uses), 16)]
    counts = []
    for ch in chunks:
        Ns = (ctypes.c_int * len(ch))(*[u[4] for u in ch])
        counts.append((Ns, _lib.value('qt_wgrad_group_blocks', len(ch), Ns)))
    part = W.new_empty(sum(c for _, c in counts), W.shape[0], Co)
    off = 0
    for ch, (Ns, nb) in zip(chunks, counts):
        vp, ip = ctypes.c_void_p * len(ch), ctypes.c_int * len(ch)
        pa = lambda f: vp(*[(f(u).data_ptr() if f(u) is not None else None) for u in ch])
        _lib.call('qt_wgrad_group', len(ch),
                  pa(lambda u: u[0][0]), ip(*[_ld(u[0][0]) for u in ch]), pa(lambda u: u[1][0]),
                  pa(lambda u: u[0][1]) if two else None, ip(*[_ld(u[0][1]) for u in ch]) if two else None,
                  pa(lambda u: u[1][1]) if two else None,
                  pa(lambda u: u[2]), pa(lambda u: u[3]), Ns, pa(lambda u: u[5]), K, Cs[0], Cs[1] if two else 0, ksp, Co,
                  ptr(part[off:]), sm)
        off += nb
    gW = torch.empty_like(W)
    _lib.call('qt_colsum', ptr(part), part.shape[0], W.numel(), ptr(gW))
    return gW


def _spmm1(mesh, x, ldx, alpha, out, ldo, p=None, ldp=0, beta=0.0, q=None, ldq=0, gamma=0.0, pad4=0, act=ACT_NONE, res=None,
           ldr=0, drop=None):
    """Raw qt_spmm1 on data pointers (ints): single strided columns of (N, 4) matrices."""
    _lib.call('qt_spmm1', ptr(mesh.rowptr), ptr(mesh.col), ptr(mesh.nrm), ptr(mesh.ell), mesh.N, ptr(mesh.n_dev), x, ldx, alpha,
              p, ldp, beta, q, ldq, gamma, out, ldo, pad4, act, ptr(res), ldr, ptr(drop))


class _ScalarCheb3(Function):
    """y = tanh(drop (u_0 + L^ u_1 + T_2(L^) u_2)) + res[:, 0] from U = [u_0 u_1 u_2 0] (N, 4): the propagation half of a
    K = 3 ChebConv with ONE output channel whose coefficient columns were applied first (U = z [w_0 w_1 w_2] + [b 0 0 0],
    an ordinary cheb_poly with K = 1).  Clenshaw on single columns: b_1 = u_1 + 2 L^ u_2, y_pre = u_0 + L^ b_1 - u_2 -- two
    launches that move 4 bytes per row and neighbour where propagating z first moves its 64-byte rows (the reference order,
    model/seq2seq.py:174-186 through PyG's ChebConv; the same sum, associated differently).  Returns (N, 4): column 0 = y."""

    @staticmethod
    def forward(ctx, U, res, drop, mesh, alias=False):
        assert U.dim() == 2 and U.shape[1] == 4 and U.is_contiguous() and U.dtype == torch.float32
        N = U.shape[0]
        res, ldr = _rows(res)
        drop = _c(drop)
        b1 = U.new_empty(N)
        Y = U.new_empty(N, 4)
        u = U.data_ptr()
        _spmm1(mesh, u + 8, 4, 2.0, b1.data_ptr(), 1, p=u + 4, ldp=4, beta=1.0)
        _spmm1(mesh, b1.data_ptr(), 1, 1.0, Y.data_ptr(), 4, p=u, ldp=4, beta=1.0, q=u + 8, ldq=4, gamma=-1.0, pad4=1,
               act=ACT_TANH_RES, res=res, ldr=ldr, drop=drop)
        ctx.save_for_backward(Y, res, drop)
        ctx.mesh, ctx.alias = mesh, alias
        if alias:
            # Y once more for its second consumer (the same storage, but NOT a view of Y: the consumer looks at `_base`): both
            # gradients then arrive here and qt_act_bwd adds them on load, instead of an elementwise launch of autograd's
            ctx.set_materialize_grads(False)
            return Y, Y.new_empty(0).set_(Y.untyped_storage(), Y.storage_offset(), Y.shape, Y.stride())
        return Y

    @staticmethod
    def backward(ctx, gY, gY2=None):
        Y, res, drop = ctx.saved_tensors
        mesh = ctx.mesh
        N = Y.shape[0]
        if gY is None:
            gY, gY2 = gY2, None
        if gY is None:
            return (None,) * 5
        gin = _c(gY.float())
        gin2 = _c(gY2.float()) if gY2 is not None else None
        G = torch.empty_like(gin)
        # (qt_act_bwd writes whole rows of the residual's ROW STRIDE -- column 0 = gY[:, 0], the rest 0 -- so a column view of a
        # wider matrix gets a buffer of that stride and its gradient is the matching column view)
        gbuf = res.new_empty(N, _row_stride(res)) if ctx.needs_input_grad[1] else None
        gres = gbuf
        if N > 0:
            _lib.call('qt_act_bwd', ptr(gin), ptr(Y), ptr(res), _row_stride(res), ptr(drop), ACT_TANH_RES, N, ptr(mesh.n_dev), 4,
                      ptr(G), ptr(gres), ptr(gin2))
            # G[:, 0] = g = dL/dy_pre;  gu_1 = L^ g -> column 1;  gu_2 = 2 L^ gu_1 - g -> column 2  (L^ symmetric)
            g = G.data_ptr()
            _spmm1(mesh, g, 4, 1.0, g + 4, 4)
            _spmm1(mesh, g + 4, 4, 2.0, g + 8, 4, p=g, ldp=4, beta=-1.0)
        return G, (gbuf[:, :res.shape[1]] if gbuf is not None else None), None, None, None


def scalar_cheb3(U, res, drop, mesh, alias=False):
    """alias=True: returns (Y, Y again) -- see _ScalarCheb3.forward."""
    return _ScalarCheb3.apply(U, res, drop, mesh, alias)


def pad_bias_rows(W, Ks):
    """Zero-pad the Ks bias rows of a packed weight to a multiple of 4 (they pair with mesh.cheb_ones columns)."""
    pad = (-Ks) % 4
    return torch.nn.functional.pad(W, (0, 0, 0, pad)) if pad else W


def cheb_poly(Z, W, mesh, K, Ks, act=ACT_NONE, res=None, drop=None, acc=None, post=None):
    """Z: one (N, C) matrix or a pair (Za, Zb) standing for [Za | Zb].  W: ((K*C + Ks [padded to a multiple of 4]), Co).
    acc: GradAcc shared by all uses of W in this pass.  post = (W2 (Co + 1 [padded to Co + 4], 4), acc2): returns (Y, U) with
    U = cheb_poly(Y, W2, mesh, 1, 1, acc=acc2) computed by the same launch (Co = 16 on the GPU; else two calls)."""
    Za, Zb = Z if isinstance(Z, (tuple, list)) else (Z, None)
    C = Za.shape[1] + (Zb.shape[1] if Zb is not None else 0)
    if W.shape[0] == K * C + Ks and Ks % 4:
        W = pad_bias_rows(W, Ks)
    if post is not None:
        W2, acc2 = post
        Co = W.shape[1]
        if W2.shape[0] == Co + 1:
            W2 = pad_bias_rows(W2, 1)
        if _HEAD_FUSE and Co == 16 and W2.shape == (Co + 4, 4) and act in (ACT_NONE, ACT_RELU) and Za.is_cuda:
            return _ChebPoly.apply(Za, Zb, W, res, drop, mesh, K, Ks, act, acc, W2, acc2)
        Y = _ChebPoly.apply(Za, Zb, W, res, drop, mesh, K, Ks, act, acc)
        return Y, _ChebPoly.apply(Y, None, W2, None, None, mesh, 1, 1, ACT_NONE, acc2)
    return _ChebPoly.apply(Za, Zb, W, res, drop, mesh, K, Ks, act, acc)


def compose_chebconvs(weights, biases):
    """Collapse stacked ChebConvs (no nonlinearity in between, model/model.py:95-96) into one
    Chebyshev series, using T_a T_b = (T_{a+b} + T_|a-b|) / 2.

    weights[l]: (G, K, in_l, out) stacked lins^T of layer l for G independent stacks;
    biases[l]: (G, out).  Returns M (G, 2L+1 .. , in_0, out) and beta (G, 2(L-1)+1, out).
    """
    P = weights[0]
    beta = biases[0].unsqueeze(1)
    for Wl, bl in zip(weights[1:], biases[1:]):
        # The bias series rides along as one extra input row of P (padded to P's orders), so that one pair of batched
        # products does the whole step: R[g, a, b] = P[g, a] @ W[g, b], then the order combination comb (J x ab) @ R.
        # (Two einsums here expanded into ~35 tiny kernels forward + backward per branch.)
        G_, ka, I, O = P.shape
        kb, Pd = Wl.shape[1], Wl.shape[3]
        kbeta = beta.shape[1]
        ext = torch.cat([P, torch.nn.functional.pad(beta, (0, 0, 0, ka - kbeta)).unsqueeze(2)], dim=2)     # (G, ka, I+1, O)
        W2 = Wl.permute(0, 2, 1, 3).reshape(G_, O, kb * Pd)
        R = torch.bmm(ext.reshape(G_, ka * (I + 1), O), W2).view(G_, ka, I + 1, kb, Pd)
        R = R.permute(0, 1, 3, 2, 4).reshape(G_, ka * kb, (I + 1) * Pd)
        comb = _comb(ka, kb, P.device).view(ka + kb - 1, ka * kb).to(P.dtype)
        out = torch.matmul(comb, R).view(G_, ka + kb - 1, I + 1, Pd)
        P = out[:, :, :I]
        nb = kbeta + kb - 1                                               # orders the bias series reaches
        beta = torch.cat([out[:, :1, I] + bl.unsqueeze(1), out[:, 1:nb, I]], dim=1)
    return P, beta


def _series3(B):
    return B.unsqueeze(1) if B.dim() == 2 else B          # a layer's bias (4, h) is a one-order bias series


class _ComposeStep(Function):
    """Series (P0 (4, Ka, in, h), bias series B0 (4, Kb0, h)) through the layer (P1 (4, K, h, h), B1 (4, h)): one product of
    the weight-space composition, natural layout (qt_compose_step_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, P0, B0, P1, B1):
        ins = [_c(P0.float()), _c(_series3(B0).float()), _c(P1.float()), _c(B1.float())]
        _lib.require_cuda(ins[0], 'weight stacks')
        _, Ka, cin, h = ins[0].shape
        Kb0, K = ins[1].shape[1], ins[2].shape[1]
        P = ins[0].new_empty(4, Ka + K - 1, cin, h)
        B = ins[0].new_empty(4, Kb0 + K - 1, h)
        _lib.call('qt_compose_step_fwd', *[ptr(t) for t in ins], Ka, Kb0, K, cin, h, ptr(P), ptr(B))
        ctx.save_for_backward(*ins)
        ctx.b0_shape = B0.shape
        return P, B

    @staticmethod
    def backward(ctx, gP, gB):
        ins = ctx.saved_tensors
        _, Ka, cin, h = ins[0].shape
        Kb0, K = ins[1].shape[1], ins[2].shape[1]
        gP = _c(gP) if gP is not None else torch.zeros(4, Ka + K - 1, cin, h, device=ins[0].device)
        gB = _c(gB) if gB is not None else torch.zeros(4, Kb0 + K - 1, h, device=ins[0].device)
        outs = [torch.empty_like(t) for t in ins]
        _lib.call('qt_compose_step_bwd', *[ptr(t) for t in ins], Ka, Kb0, K, cin, h, ptr(gP), ptr(gB), *[ptr(t) for t in outs])
        outs[1] = outs[1].view(ctx.b0_shape)
        return tuple(outs)


class _Compose2(Function):
    """Packed gate matrices of a GConvLSTM from the (folded) series of its x and h branches and their LAST layers, straight
    into the layout the gate GEMM multiplies with (qt_compose2_fwd / _bwd: one launch each way).  Same algebra as
    compose_chebconvs + the row / column layout of GConvLSTM._assemble, which remain the reference implementation.
    variants: tuple of with_h flags; returns W per variant, then W^T per variant."""

    @staticmethod
    def forward(ctx, Px0, Bx0, Px1, Bx1, Ph0, Bh0, Ph1, Bh1, cin_pad, variants):
        ins = [_c(t.float()) for t in (Px0, _series3(Bx0), Px1, Bx1, Ph0, _series3(Bh0), Ph1, Bh1)]
        _lib.require_cuda(ins[0], 'weight stacks')
        _, Ka, cin, h = ins[0].shape
        Kb0, K = ins[1].shape[1], ins[2].shape[1]
        assert ins[2].shape == (4, K, h, h) and ins[4].shape == (4, Ka, h, h) and ins[6].shape == (4, K, h, h)
        assert ins[5].shape[1] == Kb0
        K2, ksp = Ka + K - 1, (Kb0 + K - 1 + 3) // 4 * 4
        W1 = ins[0].new_empty(K2 * (cin_pad + h) + ksp, 4 * h) if True in variants else None
        W0 = ins[0].new_empty(K2 * cin_pad + ksp, 4 * h) if False in variants else None
        WT1 = W1.new_empty(W1.shape[1], W1.shape[0]) if W1 is not None else None
        WT0 = W0.new_empty(W0.shape[1], W0.shape[0]) if W0 is not None else None
        _lib.call('qt_compose2_fwd', *[ptr(t) for t in ins], Ka, Kb0, K, cin, cin_pad, h, ptr(W1), ptr(W0), ptr(WT1), ptr(WT0))
        ctx.save_for_backward(*ins)
        ctx.cin_pad, ctx.variants, ctx.b_shapes = cin_pad, tuple(variants), (Bx0.shape, Bh0.shape)
        ctx.set_materialize_grads(False)
        outs = tuple(W1 if v else W0 for v in variants) + tuple(WT1 if v else WT0 for v in variants)
        ctx.mark_non_differentiable(*outs[len(variants):])
        return outs

    @staticmethod
    def backward(ctx, *gWs):
        ins = ctx.saved_tensors
        _, Ka, cin, h = ins[0].shape
        Kb0, K = ins[1].shape[1], ins[2].shape[1]
        g = {v: (_c(gw) if gw is not None else None) for v, gw in zip(ctx.variants, gWs)}        # (the transposes have none)
        if all(v is None for v in g.values()):
            return (None,) * 10
        outs = [torch.empty_like(t) for t in ins]
        _lib.call('qt_compose2_bwd', *[ptr(t) for t in ins], Ka, Kb0, K, cin, ctx.cin_pad, h, ptr(g.get(True)), ptr(g.get(False)),
                  *[ptr(t) for t in outs])
        outs[1], outs[5] = outs[1].view(ctx.b_shapes[0]), outs[5].view(ctx.b_shapes[1])
        return (*outs, None, None)


def compose_pack(Px, Bx, Ph, Bh, cin_pad, variants):
    """([W per variant], [W^T per variant], K', Ks) from the per-layer stacks of the x and h branches (L >= 2 layers each):
    Px[l] (4, K, in_l, h), Bx[l] (4, h), likewise Ph, Bh.  Layers 0 .. L-2 fold into one series per branch
    (qt_compose_step), the last product lands in the packed layout (qt_compose2)."""
    def fold(Ps, Bs):
        P, B = Ps[0], Bs[0]
        for Wl, bl in zip(Ps[1:-1], Bs[1:-1]):
            P, B = _ComposeStep.apply(P, B, Wl, bl)
        return P, B
    (Pxf, Bxf), (Phf, Bhf) = fold(Px, Bx), fold(Ph, Bh)
    outs = _Compose2.apply(Pxf, Bxf, Px[-1], Bx[-1], Phf, Bhf, Ph[-1], Bh[-1], cin_pad, tuple(variants))
    nv = len(variants)
    K = Px[-1].shape[1]
    return outs[:nv], outs[nv:], Pxf.shape[1] + K - 1, _series3(Bxf).shape[1] + K - 1


_COMB = {}


def _comb(ka, kb, device):
    key = (ka, kb, str(device))
    if key not in _COMB:
        c = torch.zeros(ka + kb - 1, ka, kb)
        for a in range(ka):
            for b in range(kb):
                c[a + b, a, b] += 0.5
                c[abs(a - b), a, b] += 0.5
        _COMB[key] = c.to(device)
    return _COMB[key]


# ------------------------------------------------------------------------------ TransformerConv attention
_ATTN_CALLS = [0]
_ATTN_EPOCH = {}        # device -> int32[1] step counter mixed into every attention-dropout seed on the device


def dropout_epoch(device):
    key = str(device)
    if key not in _ATTN_EPOCH:
        _ATTN_EPOCH[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _ATTN_EPOCH[key]


def advance_dropout_epoch(device):
    """Once per training step (Seq2Seq.process_inputs): inside a captured step the launch arguments -- the host-side seeds
    included -- are frozen, the counter on the device is what makes every replay draw new attention-dropout masks."""
    ep = _ATTN_EPOCH.get(str(device))
    if ep is not None:
        ep.add_(1)


class _Attention(Function):
    """out = edge-softmax attention over the mesh adjacency + skip, from the fused projection proj = [q | k | v | skip]
    (N, 4C) and the edge-feature weight We (C, 2)  (PyG TransformerConv as configured by model/model.py:51).
    With heads = G > 1: G convolutions on the same mesh in one launch -- proj (N, G 4C), We (G, C, 2), out (N, G C), head g in
    column block g; gmod < G: the incoming gradient has only gmod column blocks and head g reads block g % gmod (the caller
    summed groups of gmod heads).
    acc: GradAcc shared by all uses of We in this forward pass (the dWe partials of every use add up in one slab and the
    pass's last backward reduces it), or None."""

    @staticmethod
    def forward(ctx, proj, We, mesh, c_real, keep, seed, acc, heads=1, gmod=0):
        proj, We = _c(proj.float()), _c(We.float())
        ctx.epoch = dropout_epoch(proj.device) if keep < 1.0 else None
        G = heads
        N, C = proj.shape[0], proj.shape[1] // (4 * G)
        xy, selfpair, eattr, _ = mesh.attn_geometry()
        out = proj.new_empty(N, G * C)
        stats = proj.new_empty(G, N, 2)
        _lib.call('qt_attn_fwd', ptr(mesh.rowptr), ptr(mesh.col), ptr(xy), ptr(eattr), ptr(selfpair), ptr(proj), 4 * C * G, ptr(We), C, c_real,
                  N, ptr(mesh.n_dev), keep, seed, ptr(ctx.epoch), ptr(out), ptr(stats), G, 0, 0, 0, 0)
        ctx.save_for_backward(proj, We, stats, out)
        ctx.mesh, ctx.c_real, ctx.keep, ctx.seed, ctx.acc, ctx.G, ctx.gmod = mesh, c_real, keep, seed, acc, G, gmod or G
        ctx.use_idx = acc.enter() if acc is not None else 0
        if ctx.gmod < G:            # the sum over the head groups (conv_x + conv_h of a gate, model/model.py:394-424)
            return out.view(N, G // ctx.gmod, ctx.gmod * C).sum(dim=1)
        return out

    @staticmethod
    def backward(ctx, g):
        proj, We, stats, out = ctx.saved_tensors
        mesh, acc, G, gmod = ctx.mesh, ctx.acc, ctx.G, ctx.gmod
        N, C = proj.shape[0], proj.shape[1] // (4 * G)
        xy, selfpair, eattr, rev = mesh.attn_geometry()
        coef = proj.new_empty(G, rev.numel() + N, 2)       # per message: target pass -> source pass
        g, ld_g = _rows(g.float())                  # a column block of the gates' gradient is read in place
        gproj = torch.empty_like(proj)
        if acc is None:
            nblk = max(_lib.value('qt_attn_blocks', N, C), 1)
            part = proj.new_empty(nblk, G * 2 * C) if N > 0 else proj.new_zeros(nblk, G * 2 * C)
        else:
            nblk = max(_lib.value('qt_attn_blocks', max(mesh.B * mesh.P, N), C), 1)
            part = acc.slab(proj, nblk, G * 2 * C)
        if N > 0:
            _lib.call('qt_attn_bwd', ptr(mesh.rowptr), ptr(mesh.col), ptr(xy), ptr(eattr), ptr(selfpair), ptr(proj), 4 * C * G, ptr(We), C,
                      ctx.c_real, N, ptr(mesh.n_dev), ctx.keep, ctx.seed, ptr(ctx.epoch), ptr(g), ld_g, ptr(stats), ptr(out), 0, ptr(gproj),
                      ptr(part), 0 if acc is None else 1, ptr(rev), ptr(coef), rev.numel(), G, gmod, 0, 0, 0, 0)
        else:
            gproj.zero_()
        none = (None,) * 7
        if acc is not None and not acc.leave(ctx.use_idx):
            return (gproj, None) + none
        psum = proj.new_empty(G * 2 * C)
        _lib.call('qt_colsum', ptr(part), nblk, G * 2 * C, ptr(psum))         # part (block, head, 2C)
        gWe = psum.view(G, 2, C).transpose(1, 2).contiguous()
        return (gproj, gWe[0] if We.dim() == 2 else gWe) + none


def attention(proj, We, mesh, c_real, dropout_p=0.0, training=False, acc=None, heads=1, gmod=0):
    keep = 1.0 - dropout_p if (training and dropout_p > 0) else 1.0
    _ATTN_CALLS[0] += 1
    seed = (_ATTN_CALLS[0] * 2654435761 + int(torch.initial_seed())) & 0xFFFFFFFF       # host-side counter: no device sync
    return _Attention.apply(proj, We, mesh, c_real, keep, seed, acc, heads, gmod)


_PROJ_BWD_FUSED = os.environ.get('QT_NO_PROJ_BWD_FUSED') != '1'      # (A/B switch)
_PROJ_BWD_SHARED = os.environ.get('QT_NO_PROJ_BWD_SHARED') != '1'    # (A/B switch: the first-layer segments too)
_STATS = {'skip_alias': 0}      # (how often a layer's gradient array was completed in place: tests look at it)


class _MultiConv(Function):
    """One layer of the G attention-convolution stacks of a recurrent cell (model/model.py:394-424 with TransformerConv :51) in
    three launches instead of 2 G: the projections [q | k | v | skip] of all stacks go into ONE array P (qt_proj_group, one
    launch per input segment), the edge softmax of the G heads is one launch (qt_attn_fwd, G heads).

    P is (G, 4, N, C): one dense (N, C) plane per head and block -- the layout the GEMMs write and read as planes, with gathered
    k / v rows that are whole 128-byte lines (measured against rows side by side in one (N, G 4C) matrix: the same 224 us per
    8-head forward launch, so the layout is a convenience, not a speed-up).
    segments: [(A_s, W_s (Gin_s, Cin_s + 4, Co_s))]: group g of segment s multiplies A_s (N, Cin_s) (Gin_s = 1) or A_s[g] of
    (Gin_s, N, Cin_s) with W_s[g] (bias in row Cin_s) and fills the next Co_s / C planes of P.  Layer 0 of a cell has the segments
    (X, Wx (1, ., 4 4C)) and (H, Wh (1, ., 4 4C)), deeper layers one segment (previous layer's output (8, N, C), W (8, C + 4, 4C)).
    Output: (G, N, C), or with gmod < G the rows (N, G / gmod, gmod C) whose head groups the consumer sums (conv_x + conv_h per
    gate: ops.lstm_cell adds them inside its kernel and hands back ONE gradient for both, as a stride-0 view).  acc: GradAcc of this layer's weights for the pass (dWe slab + deferred grouped weight gradients), or None."""

    @staticmethod
    def forward(ctx, mesh, c_real, keep, seed, acc, gmod, nseg, *args):
        As = [_c(a.float()) if a.dim() == 3 else _rows(a.float())[0] for a in args[:nseg]]
        Ws = [_c(w.float()) for w in args[nseg:2 * nseg]]
        We = _c(args[2 * nseg].float())
        G, C = We.shape[0], We.shape[1]
        N = As[0].shape[-2]
        ctx.epoch = dropout_epoch(We.device) if keep < 1.0 else None
        P = As[0].new_empty(G, 4, N, C)
        ones = mesh.cheb_ones(1)
        hoff, segs = 0, []
        for A, W in zip(As, Ws):
            gin, kin, co = W.shape
            cin = kin - 4
            assert A.shape[-1] == cin and (A.dim() == 3) == (gin > 1) and co % (4 * C) == 0, (A.shape, W.shape)
            lda, gsa = (cin, N * cin) if gin > 1 else (_ld(A), 0)
            if N > 0:
                _lib.call('qt_proj_group', ptr(A), lda, gsa, 1, cin, ptr(ones), ptr(W), None, kin * co, gin, co // C, C,
                          P.data_ptr() + 4 * hoff * 4 * N * C, C, (co // C) * N * C, 0, N, ptr(mesh.n_dev))
            segs.append((hoff, gin, cin, co, lda, gsa))
            hoff += gin * co // (4 * C)
        assert hoff == G, (hoff, G)
        xy, selfpair, eattr, _ = mesh.attn_geometry()
        gmod = gmod or G
        summed = gmod < G
        out = P.new_empty(N, G * C) if summed else P.new_empty(G, N, C)
        stats = P.new_empty(G, N, 2)
        _lib.call('qt_attn_fwd', ptr(mesh.rowptr), ptr(mesh.col), ptr(xy), ptr(eattr), ptr(selfpair), ptr(P), C, ptr(We), C, c_real,
                  N, ptr(mesh.n_dev), keep, seed, ptr(ctx.epoch), ptr(out), ptr(stats), G, G * C if summed else C, N * C, 4 * N * C,
                  C if summed else N * C)
        ctx.save_for_backward(P, We, stats, out, *As, *Ws)
        ctx.mesh, ctx.c_real, ctx.keep, ctx.seed, ctx.acc, ctx.G, ctx.gmod, ctx.segs = mesh, c_real, keep, seed, acc, G, gmod, segs
        ctx.use_idx = acc.enter() if acc is not None else 0
        if summed:                  # (N, G / gmod, gmod C): the consumer adds the head groups (lstm_cell: inside its kernel)
            return out.view(N, G // gmod, gmod * C)
        return out

    @staticmethod
    def backward(ctx, g):
        P, We, stats, out = ctx.saved_tensors[:4]
        nseg = len(ctx.segs)
        As, Ws = ctx.saved_tensors[4:4 + nseg], ctx.saved_tensors[4 + nseg:]
        mesh, acc, G, gmod = ctx.mesh, ctx.acc, ctx.G, ctx.gmod
        N, C = P.shape[2], P.shape[3]
        xy, selfpair, eattr, rev = mesh.attn_geometry()
        coef = P.new_empty(G, rev.numel() + N, 2)
        summed = gmod < G
        if summed and g.stride(1) == 0:       # the head groups were summed downstream: one gradient for all of them
            g, ld_g = _rows(g[:, 0].float())
            hs_g = C
        elif summed:
            g, ld_g, hs_g, gmod = _c(g.float()), G * C, C, G
        gP, alias = None, 0
        if not summed:
            # the next layer's backward writes this gradient straight into the skip block of a (G, 4, N, C) array (see the data
            # gradient below): that array becomes gP -- the skip block's gradient IS g, nothing to store again
            base = g._base if (g.dim() == 3 and N > 0 and g.dtype == P.dtype) else None
            if (base is not None and base.shape == P.shape and base.dtype == P.dtype and base.is_contiguous()
                    and g.data_ptr() == base.data_ptr() + 4 * 3 * N * C and g.stride() == (4 * N * C, C, 1)):
                gP, alias, ld_g, hs_g = base, 2, C, 4 * N * C
                _STATS['skip_alias'] += 1
            else:
                g, ld_g, hs_g = _c(g.float()), C, N * C
        if gP is None:
            gP = torch.empty_like(P)
        if acc is None:
            nblk = max(_lib.value('qt_attn_blocks', N, C), 1)
            part = P.new_empty(nblk, G * 2 * C) if N > 0 else P.new_zeros(nblk, G * 2 * C)
        else:
            nblk = max(_lib.value('qt_attn_blocks', max(mesh.B * mesh.P, N), C), 1)
            part = acc.slab(P, nblk, G * 2 * C)
        if N > 0:
            _lib.call('qt_attn_bwd', ptr(mesh.rowptr), ptr(mesh.col), ptr(xy), ptr(eattr), ptr(selfpair), ptr(P), C, ptr(We), C,
                      ctx.c_real, N, ptr(mesh.n_dev), ctx.keep, ctx.seed, ptr(ctx.epoch), ptr(g), ld_g, ptr(stats), ptr(out),
                      G * C if summed else C, ptr(gP), ptr(part), (0 if acc is None else 1) | alias, ptr(rev), ptr(coef), rev.numel(), G,
                      gmod, N * C, 4 * N * C, hs_g, C if summed else N * C)
        else:
            gP.zero_()
        gAs = []
        pslab = acc.pslab if acc is not None else {}
        fused = set()
        for s, ((hoff, gin, cin, co, lda, gsa), A, W) in enumerate(zip(ctx.segs, As, Ws)):
            if not ctx.needs_input_grad[7 + s]:
                gAs.append(None)
                continue
            heads = co // (4 * C)
            if (_PROJ_BWD_FUSED and N > 0 and cin == 32 and C == 32 and ctx.needs_input_grad[7 + nseg + s] and W.is_contiguous()
                    and ((gin > 1 and heads == 1 and A.is_contiguous()) or (_PROJ_BWD_SHARED and gin == 1 and heads > 1 and lda % 4 == 0 and A.data_ptr() % 16 == 0))):
                # data gradient AND this use's weight gradient in one pass over the gradient planes (csrc/projbwd.hip): the planes are
                # read once and need not be kept for the deferred grouped weight gradient
                G_s = gin * heads
                if s not in pslab:
                    pslab[s] = W.new_zeros(_lib.value('qt_proj_bwd_blocks', G_s), gin, cin + 4, co)
                if gin > 1:
                    nxt = A.new_empty(gin, 4, N, cin)
                    _lib.call('qt_proj_bwd', gP.data_ptr() + 4 * hoff * 4 * N * C, 4 * N * C, N * C, ptr(A), N * cin, cin, ptr(W), (cin + 4) * co, co,
                              nxt.data_ptr() + 4 * 3 * N * cin, 4 * N * cin, cin, ptr(pslab[s]), N, ptr(mesh.n_dev), gin, cin, C, 1, 1)
                    gAs.append(nxt[:, 3])
                else:
                    # a cell's first layer: the `heads` stacks of the segment share the input (gsA = 0) and sit side by side in one
                    # weight matrix; every head leaves its own partial data gradient, added here
                    partial = A.new_empty(heads, N, cin)
                    _lib.call('qt_proj_bwd', gP.data_ptr() + 4 * hoff * 4 * N * C, 4 * N * C, N * C, ptr(A), 0, lda, ptr(W), 4 * C, co,
                              ptr(partial), N * cin, cin, ptr(pslab[s]), N, ptr(mesh.n_dev), heads, cin, C, 1, 1)
                    gAs.append(partial.sum(dim=0))
                fused.add(s)
                continue
            if gin > 1 and N > 0:
                # the input is the previous layer's output (gin, N, cin): its gradient goes into block 3 of a fresh (gin, 4, N, cin)
                # array, which that layer's backward then completes as ITS gP (see above)
                nxt = A.new_empty(gin, 4, N, cin)
                gA = nxt[:, 3]
                go, gso = nxt.data_ptr() + 4 * 3 * N * cin, 4 * N * cin
            else:
                gA = A.new_empty(A.shape)
                go, gso = ptr(gA), 0
            if N > 0:         # gA_g = gP_g W_g[:cin]^T: the forward weight's own rows are the transposed operand
                _lib.call('qt_proj_group', gP.data_ptr() + 4 * hoff * 4 * N * C, C, (co // C) * N * C, co // C, C, None, None, ptr(W),
                          (cin + 4) * co, gin, 1, cin, go, cin, gso, 1, N, ptr(mesh.n_dev))       # (groups last to first: see csrc/attn.hip)
            else:
                gA.zero_()
            gAs.append(gA)
        last = acc is None or acc.leave(ctx.use_idx)
        need_w = any(ctx.needs_input_grad[7 + nseg:7 + 2 * nseg])
        pending = acc.pending if acc is not None else []
        if need_w and N > 0 and len(fused) < nseg:          # (the segments in `fused` have their share in the slabs already)
            pending.append((As, gP, N, mesh.n_dev, mesh.cheb_ones(1), fused))
        if not last:
            return (None,) * 7 + tuple(gAs) + (None,) * (nseg + 1)
        gWs = [None] * nseg
        if need_w:
            for s, (seg, W) in enumerate(zip(ctx.segs, Ws)):
                uses = [u for u in pending if s not in u[5]]
                if s in pslab:           # every fused use of the pass added its partial sums into the same slabs
                    gWs[s] = torch.empty_like(W)
                    _lib.call('qt_colsum', ptr(pslab[s]), pslab[s].shape[0], W.numel(), ptr(gWs[s]))
                    if uses:
                        gWs[s] += _wgrad_groups(uses, s, seg, C, W)
                else:
                    gWs[s] = _wgrad_groups(uses, s, seg, C, W)
        if acc is not None:
            acc.pending = []
        psum = P.new_empty(G * 2 * C)
        _lib.call('qt_colsum', ptr(part), nblk, G * 2 * C, ptr(psum))
        gWe = psum.view(G, 2, C).transpose(1, 2).contiguous()
        return (None,) * 7 + tuple(gAs) + tuple(gWs) + (gWe,)


def _wgrad_groups(uses, s, seg, C, W):
    """(gin, cin + 4, co) gradient of segment s's weights, summed over the uses [(As, gP (G, 4, N, C), N, n_dev, ones)] of a pass."""
    import ctypes
    hoff, gin, cin, co, lda, gsa = seg
    if not uses:
        return torch.zeros_like(W)
    chunks = [uses[i:i + 16] for i in range(0, len(uses), 16)]
    counts = []
    for ch in chunks:
        Ns = (ctypes.c_int * len(ch))(*[u[2] for u in ch])
        counts.append((Ns, _lib.value('qt_wgrad_group_blocks', len(ch), Ns)))
    part = W.new_empty(sum(c for _, c in counts), gin, cin + 4, co)
    off = 0
    for ch, (Ns, nb) in zip(chunks, counts):
        vp, ip = ctypes.c_void_p * len(ch), ctypes.c_int * len(ch)
        _lib.call('qt_wgrad_groups', len(ch), vp(*[u[0][s].data_ptr() for u in ch]),
                  ip(*[(cin if gin > 1 else _ld(u[0][s])) for u in ch]),
                  vp(*[u[4].data_ptr() for u in ch]), vp(*[u[1].data_ptr() + 4 * hoff * 4 * u[2] * C for u in ch]), Ns,
                  vp(*[ptr(u[3]) for u in ch]), cin, 4, co, C, C, gin, cin if gin > 1 else 0, co, 1, ptr(part[off:]))
        off += nb
    gW = torch.empty_like(W)
    _lib.call('qt_colsum', ptr(part), part.shape[0], W.numel(), ptr(gW))
    return gW


def multi_conv(segments, We, mesh, c_real, dropout_p=0.0, training=False, acc=None, gmod=0):
    """segments: [(A, W)] as in _MultiConv; We (G, C, 2).  Returns (G, N, C), or with gmod < G the rows (N, G / gmod, gmod C)
    whose head groups the consumer adds (ops.lstm_cell does, inside its kernel)."""
    keep = 1.0 - dropout_p if (training and dropout_p > 0) else 1.0
    _ATTN_CALLS[0] += 1
    seed = (_ATTN_CALLS[0] * 2654435761 + int(torch.initial_seed())) & 0xFFFFFFFF
    return _MultiConv.apply(mesh, c_real, keep, seed, acc, gmod, len(segments), *[a for a, _ in segments], *[w for _, w in segments], We)


# ------------------------------------------------------------------------------ LSTM cell
class _LstmCell(Function):
    """(O, LayerNorm_h(H'), LayerNorm_c(C')) from gate pre-activations (model/model.py:394-428,
    model/seq2seq.py:64-75)."""

    @staticmethod
    def forward(ctx, G, Cprev, wc, b, ln, mesh, acc):
        G = _c(G)
        ctx.pair = G.dim() == 3             # (N, 2, 4h): the conv_x and conv_h sums side by side, added inside the kernel
        assert not ctx.pair or G.shape[1] == 2
        N, h4 = G.shape[0], G.shape[-1]
        h = h4 // 4
        wc, b, ln = _c(wc), _c(b), _c(ln)
        Cprev, ld_c = _rows(Cprev)
        O, Hn, Cn = (G.new_empty(N, h) for _ in range(3))
        gates = G.new_empty(N, h4)
        _lib.call('qt_lstm_fwd', ptr(G), G.data_ptr() + 4 * h4 if ctx.pair else None, 2 * h4 if ctx.pair else h4, ptr(Cprev), ld_c, ptr(wc),
                  ptr(b), ptr(ln), N, ptr(mesh.n_dev), h, ptr(O), ptr(Hn), ptr(Cn), ptr(gates))
        ctx.save_for_backward(gates, Cprev, wc, ln)
        ctx.mesh, ctx.acc = mesh, acc
        ctx.use_idx = acc.enter() if acc is not None else 0
        ctx.set_materialize_grads(False)         # an unused output arrives as None, not as a zero-filled (N, h) buffer
        return O, Hn, Cn

    @staticmethod
    def backward(ctx, gO, gHn, gCn):
        gates, Cprev, wc, ln = ctx.saved_tensors
        res = _lstm_backward(gO, gHn, gCn, gates, Cprev, wc, ln, ctx.mesh, ctx.acc, ctx.use_idx)
        if ctx.pair:            # both addends receive the same gradient: a stride-0 view, no copy
            res = (res[0].unsqueeze(1).expand(-1, 2, -1),) + tuple(res[1:])
        return (*res, None, None)


def _lstm_backward(gO, gHn, gCn, gates, Cprev, wc, ln, mesh, acc, use_idx, dgrad=None, gHn2=None, add0=None):
    """(gG, gCprev, g_wc, g_b, g_ln) of the cell; the parameter gradients are None until the pass's last backward.
    gHn2: a second gradient of H' (added to gHn: by qt_lstm_bwd_dgrad on load, else here); add0: (N, c) added to plane 0 of
    the first part of `dgrad`'s planes (qt_lstm_bwd_dgrad only: the caller checks that this launch serves the use).
    dgrad = (Wrows, K, [part widths], [plane tensors (K, N, c)], out_sm): the data gradient of the gate GEMM, gG @ Wrows^T, is
    computed by the same launch (qt_lstm_bwd_dgrad) into the given planes (out_sm: planes 1.. slice-major); a tuple in the
    fifth place instead selects the opt-in launch that also accumulates the weight gradient (qt_lstm_bwd_fused)."""
    N, h = gates.shape[0], gates.shape[1] // 4
    fused_w = dgrad is not None and isinstance(dgrad[4], tuple)
    if gHn2 is not None and (dgrad is None or fused_w or N == 0):
        gHn, gHn2 = (gHn2 if gHn is None else gHn + gHn2), None          # (the other launches take one gradient of H')
    assert add0 is None or (dgrad is not None and not fused_w)
    (gHn, ld_gh), (gCn, ld_gc) = _rows(gHn), _rows(gCn)
    (gHn2, ld_gh2) = _rows(gHn2)
    (gO, ld_go), (Cprev, ld_c) = _rows(gO), _rows(Cprev)
    gG = None if (fused_w and N > 0) else torch.empty_like(gates)
    gCp = gates.new_empty(N, h) if Cprev is not None else None
    # (one slab row per workgroup of whichever kernel serves a use: k_lstm_bwd, k_dgrad_cell or the persistent fused launch)
    rows_of = lambda n: max(_lib.value('qt_lstm_bwd_blocks', n, h), _lib.value('qt_lstm_dgrad_blocks', n),
                            min(_lib.value('qt_lstm_fused_blocks'), -(n // -64)), 1)
    if acc is None:
        nblk = rows_of(N)
        part = gates.new_empty(nblk, 11 * h) if dgrad is None else gates.new_zeros(nblk, 11 * h)
    else:
        nblk = rows_of(max(mesh.B * mesh.P, N))      # one slab for every use of the pass, whichever kernel serves it
        part = acc.slab(gates, nblk, 11 * h)
    if N > 0 and fused_w:
        Wrows, K, Cl, planes, (Zs, TZs, S, ksp, Cs, wslab) = dgrad
        gG = None                                   # the gate gradients stay inside the launch
        _lib.call('qt_lstm_bwd_fused', ptr(gO), ld_go, ptr(gHn), ld_gh, ptr(gCn), ld_gc, ptr(gates), ptr(Cprev), ld_c,
                  ptr(wc), ptr(ln), N, ptr(mesh.n_dev), h, ptr(gCp), ptr(part), 0 if acc is None else 1,
                  ptr(Wrows), K, Cl[0], Cl[1] if len(Cl) > 1 else 0, ptr(planes[0]), ptr(planes[1]) if len(Cl) > 1 else None,
                  *_plane_args(Zs, TZs), K, Cs[0], Cs[1] if len(Cs) > 1 else 0, ptr(S), ksp, ptr(wslab), wslab.shape[0])
    elif N > 0 and dgrad is not None:
        Wrows, K, Cl, planes, out_sm = dgrad
        split = Wrows.__dict__.get('_qt_split') if hasattr(Wrows, '__dict__') else None
        _lib.call('qt_lstm_bwd_dgrad', ptr(gO), ld_go, ptr(gHn), ld_gh, ptr(gCn), ld_gc, ptr(gates), ptr(Cprev), ld_c,
                  ptr(wc), ptr(ln), N, ptr(mesh.n_dev), h, ptr(gG), ptr(gCp), ptr(part), 0 if acc is None else 1,
                  ptr(Wrows), ptr(split[0]) if split else None, ptr(split[1]) if split else None, K, Cl[0],
                  Cl[1] if len(Cl) > 1 else 0, ptr(planes[0]), ptr(planes[1]) if len(Cl) > 1 else None, int(out_sm),
                  ptr(gHn2), ld_gh2, ptr(add0))
    elif N > 0:
        if acc is None and nblk > _lib.value('qt_lstm_bwd_blocks', N, h):
            part.zero_()
        _lib.call('qt_lstm_bwd', ptr(gO), ld_go, ptr(gHn), ld_gh, ptr(gCn), ld_gc, ptr(gates), ptr(Cprev),
                  ld_c, ptr(wc), ptr(ln), N, ptr(mesh.n_dev), h, ptr(gG), ptr(gCp), ptr(part), 0 if acc is None else 1)
    if acc is not None and not acc.leave(use_idx):
        return gG, gCp, None, None, None
    psum = gates.new_empty(11 * h)
    if N > 0 or acc is not None:
        _lib.call('qt_colsum', ptr(part), nblk, 11 * h, ptr(psum))
    else:
        psum.zero_()
    psum = psum.view(11, h)
    return gG, gCp, psum[0:3], psum[3:7], (psum[7:11] if ln is not None else None)


class _GateCell(Function):
    """cheb_poly (no activation) + lstm_cell as one op for hidden sizes 8, 16, 32: the gate GEMM runs the cell in its epilogue
    (qt_dense_lstm), so the (N, 4h) pre-activations are never written.  Z = [Za | Zb] (Zb may be None).  The raw output
    gate O is returned as a column view of the saved gate activations.  Backward = the two backward passes."""

    @staticmethod
    def forward(ctx, Za, Zb, W, Cprev, wc, b, ln, mesh, K, Ks, acc_w, acc_p, alias_h=False, pass_x=False):
        _lib.require_cuda(Za, 'node features')
        Zs = _zparts(Za, Zb)
        W, wc, b, ln = _c(W.float()), _c(wc), _c(b), _c(ln)
        N = Zs[0].shape[0]
        Cs = [Z.shape[1] for Z in Zs]
        h = W.shape[1] // 4
        ksp = (Ks + 3) // 4 * 4
        assert W.shape[0] == K * sum(Cs) + ksp, f'weight rows {W.shape[0]} != {K}*{sum(Cs)}+{ksp}'
        TZs, sm = _cheb_planes(Zs, mesh, K)
        S = mesh.cheb_ones(Ks) if Ks else None
        Cprev, ld_c = _rows(Cprev)
        Hn, Cn = (Zs[0].new_empty(N, h) for _ in range(2))
        gates = Zs[0].new_empty(N, 4 * h)
        _lib.call('qt_dense_lstm', *_plane_args(Zs, TZs), K, Cs[0], Cs[1] if len(Cs) > 1 else 0, ptr(W), ptr(_w_t(W, acc_w)), ptr(S), ksp,
                  ptr(W[K * sum(Cs):]) if Ks else None, h, N, ptr(mesh.n_dev), ptr(Cprev), ld_c, ptr(wc), ptr(b), ptr(ln),
                  None, ptr(Hn), ptr(Cn), ptr(gates), sm)
        ctx.save_for_backward(*Zs, *TZs, W, gates, Cprev, wc, ln)
        ctx.mesh, ctx.K, ctx.Ks, ctx.acc_w, ctx.acc_p, ctx.nz, ctx.sm = mesh, K, Ks, acc_w, acc_p, len(Zs), sm
        ctx.use_w = acc_w.enter() if acc_w is not None else 0
        ctx.use_p = acc_p.enter() if acc_p is not None else 0
        ctx.set_materialize_grads(False)
        ctx.extras = (alias_h, pass_x)
        # alias_h: H' again as a second output (the same storage) for a second consumer; pass_x: Za again as an output (for a
        # consumer that would otherwise read Za itself).  A tensor with two consumers costs an elementwise gradient sum per use
        # in autograd; this way both gradients arrive HERE and the backward launch adds them where it reads / writes anyway.
        out = (gates[:, 3 * h:], Hn, Cn)
        if alias_h:
            out += (Hn.view_as(Hn),)
        if pass_x:
            out += (Za.view_as(Za),)
        return out

    @staticmethod
    def backward(ctx, gO, gHn, gCn, *gextra):
        alias_h, pass_x = ctx.extras
        gextra = list(gextra)
        gHn2 = gextra.pop(0) if alias_h else None
        gXp = gextra.pop(0) if pass_x else None
        nz = ctx.nz
        saved = ctx.saved_tensors
        Zs, TZs = list(saved[:nz]), list(saved[nz:2 * nz])
        W, gates, Cprev, wc, ln = saved[2 * nz:]
        need = list(ctx.needs_input_grad[:nz])
        K, N, h = ctx.K, gates.shape[0], gates.shape[1] // 4
        Cs = [Z.shape[1] for Z in Zs]
        live = [i for i, f in enumerate(need) if f]
        NB = K * sum(Cs[i] for i in live)
        dgrad = planes = None
        w_fused = False
        if N > 0 and live and h in (8, 16) and 16 < NB <= 128 and os.environ.get('QT_NO_DGRAD_FUSION') != '1':
            # the cell backward and the data gradient of the gate GEMM in one launch: gG feeds the MFMA loop from LDS
            Wb, _ = _dgrad_weight(W, K, Cs, live, ctx.acc_w)
            planes = [Zs[0].new_empty(K, N, Cs[i]) for i in live]
            osm = int(_clip_resident(ctx.mesh, [Cs[i] for i in live], K))      # planes 1.. slice-major for the fused Clenshaw
            dgrad = (Wb, K, [Cs[i] for i in live], planes, osm)
            # ... and, on request, the weight gradient too (qt_lstm_bwd_fused: gG never leaves the launch).  OFF by default: at
            # the bench shape the persistent launch takes 71 us against 47 + 21 us for this launch plus its share of the
            # deferred weight gradient -- fp32 MFMA issues on the vector pipe, so its 18 us of MFMA time, the cell arithmetic
            # and the memory phases add up instead of overlapping (HISTORY.md section C); 9.17 vs 8.98 ms per step.
            if (os.environ.get('QT_WGRAD_FUSION') == '1' and not ctx.sm and ctx.needs_input_grad[2] and ctx.acc_w is not None
                    and W.shape[0] <= 128 and NB <= (128 if h == 16 else 64)):
                ksp = (ctx.Ks + 3) // 4 * 4
                S = ctx.mesh.cheb_ones(ctx.Ks) if ctx.Ks else None
                wslab = ctx.acc_w.weight_slab(W, W.shape[0], W.shape[1])
                dgrad = dgrad[:4] + ((Zs, TZs, S, ksp, Cs, wslab),)
                osm = 0
                w_fused = True
        # (the pass-through gradient of Za rides into plane 0 of part a when the fused launch writes that plane)
        add0 = None
        if gXp is not None and dgrad is not None and not w_fused and live and live[0] == 0 and N > 0:
            add0 = _c(gXp.float())
            assert add0.shape == (N, Cs[0])
        gG, gCp, gwc, gb, gln = _lstm_backward(gO, gHn, gCn, gates, Cprev, wc, ln, ctx.mesh, ctx.acc_p, ctx.use_p, dgrad,
                                               gHn2=gHn2, add0=add0)
        gZs, gW = _cheb_backward(Zs, TZs, W, gG, ctx.mesh, K, ctx.Ks, ctx.acc_w, ctx.use_w, need, ctx.needs_input_grad[2],
                                 gTs_pre=None if planes is None else (planes, osm), w_fused=w_fused, sm=ctx.sm)
        gZa = gZs[0] if gZs is not None else None
        gZb = gZs[1] if gZs is not None and nz > 1 else None
        if gXp is not None and add0 is None and need[0]:
            gZa = gXp if gZa is None else gZa + gXp
        return gZa, gZb, gW, gCp, gwc, gb, gln, None, None, None, None, None, None, None


def gate_cell(X, H, W, Cprev, wc, b, ln, mesh, K, Ks, acc_w=None, acc_p=None, alias_h=False, pass_x=False):
    """(O, LayerNorm_h(H'), LayerNorm_c(C')) of one GConvLSTM update from Z = [X | H] (H may be None) and the packed
    gate weights W; X and H are passed as they are (column views of wider matrices included), never concatenated.
    alias_h / pass_x: further outputs -- H' once more (for its second consumer) / X once more (for a second consumer of X):
    the same values; on the fused path the same storage, with both gradients summed inside the backward launch."""
    C = X.shape[1] + (H.shape[1] if H is not None else 0)
    if W.shape[0] == K * C + Ks and Ks % 4:
        W = pad_bias_rows(W, Ks)
    if W.shape[1] in (32, 64, 128) and X.is_cuda:
        return _GateCell.apply(X, H, W, Cprev, wc, b, ln, mesh, K, Ks, acc_w, acc_p, alias_h, pass_x)
    G = cheb_poly((X, H), W, mesh, K, Ks, acc=acc_w)
    out = lstm_cell(G, Cprev, wc, b, ln, mesh, acc_p)
    return tuple(out) + ((out[1],) if alias_h else ()) + ((X,) if pass_x else ())


def lstm_cell(G, Cprev, wc, b, ln, mesh, acc=None):
    return _LstmCell.apply(G, Cprev, wc, b, ln, mesh, acc)


# ------------------------------------------------------------------------------ decoder head input
class _Head(Function):
    """[relu(LayerNorm_o(O)) | concat | 0-pad] (model/seq2seq.py:160-165) as two matrices: (N, h) and (N, hp - h).
    O may be a column view (the raw output gate inside the saved gate activations)."""

    @staticmethod
    def forward(ctx, O, ln_o, concat, hp, mesh, acc):
        (O, ld_o), ln_o, concat = _rows(O), _c(ln_o), _c(concat)
        N, h = O.shape
        Za, Zb = O.new_empty(N, h), O.new_empty(N, hp - h)
        _lib.call('qt_head_fwd', ptr(O), ld_o, ptr(ln_o), ptr(concat), N, ptr(mesh.n_dev), h, hp, ptr(Za), ptr(Zb))
        ctx.save_for_backward(O, ln_o)
        ctx.hp, ctx.has_concat, ctx.mesh, ctx.acc = hp, concat is not None, mesh, acc
        ctx.use_idx = acc.enter() if acc is not None else 0
        return Za, Zb

    @staticmethod
    def backward(ctx, gZa, gZb):
        O, ln_o = ctx.saved_tensors
        O, ld_o = _rows(O)
        N, h = O.shape
        gZa, gZb = _c(gZa), _c(gZb)
        gO = O.new_empty(N, h)
        gcat = O.new_empty(N, 1) if ctx.has_concat else None
        acc, mesh = ctx.acc, ctx.mesh
        if acc is None:
            nblk = max(_lib.value('qt_lstm_bwd_blocks', N, h), 1)
            part = O.new_empty(nblk, 2 * h)
        else:
            nblk = max(_lib.value('qt_lstm_bwd_blocks', mesh.B * mesh.P, h), 1)
            part = acc.slab(O, nblk, 2 * h)
        if N > 0:
            _lib.call('qt_head_bwd', ptr(gZa), ptr(gZb), ptr(O), ld_o, ptr(ln_o), N, ptr(mesh.n_dev), h, ctx.hp, ptr(gO),
                      ptr(gcat), ptr(part), 0 if acc is None else 1)
        if acc is not None and not acc.leave(ctx.use_idx):
            return gO, None, gcat, None, None, None
        psum = O.new_empty(2 * h)
        if N > 0 or acc is not None:
            _lib.call('qt_colsum', ptr(part), nblk, 2 * h, ptr(psum))
        else:
            psum.zero_()
        return gO, psum.view(2, h), gcat, None, None, None


def head_input(O, ln_o, concat, hp, mesh, acc=None):
    """-> (Za (N, h), Zb (N, hp - h)), the two column parts of the head's input."""
    return _Head.apply(O, ln_o, concat, hp, mesh, acc)


# ------------------------------------------------------------------------------ mesh <-> image
def _pool_raw(mesh, C, out, out_stride, out_coff, mean, img=None, S=1, src_val=None, src_mesh=None, src_inv=False,
              img_clip_stride=0):
    if _CLIP_REMESH and img is not None and C <= 8 and S * C <= 64:
        # a few scalar channels per pixel (the shapes it was measured on: the encoder's input frames S = 10, C = 4 and the
        # decoder's concat layer S = C = 1): one workgroup per (clip, 64 x 64 tile, frame, channel), LDS pyramid.  Wide rows
        # (C up to 64 in a gather's backward) would read scalars strided by C floats from every workgroup: they stay on
        # qt_pool's float4 path (round-3 advisor finding)
        _lib.call('qt_pool_clip', ptr(img), S, img_clip_stride, C, ptr(mesh.labels), ptr(mesh.level), ptr(mesh.npix), int(mean),
                  mesh.B, mesh.n, mesh.m, mesh.N, ptr(out), out_stride, out_coff)
        return
    _lib.call('qt_pool', ptr(img), S, img_clip_stride, ptr(src_val), ptr(src_mesh.labels) if src_mesh is not None else None,
              ptr(src_mesh.npix) if src_mesh is not None else None, int(src_inv), C, ptr(mesh.labels), ptr(mesh.level),
              ptr(mesh.npix), int(mean), mesh.B, mesh.n, mesh.m, mesh.N, ptr(mesh.cell) if C >= 4 else None, ptr(mesh.n_dev),
              ptr(out), out_stride, out_coff)       # 1-channel transfers: the tile kernel alone is one launch and faster


def _gather_raw(mesh, val, C, inv_npix, img):
    _lib.call('qt_gather', ptr(val), C, ptr(mesh.labels), ptr(mesh.npix) if inv_npix else None,
              mesh.B * mesh.P, ptr(img))


class _PoolImage(Function):
    """flatten (model/graph_functions.py:391-419): img (B, S, n*m, C) -> node means (S, N, C)."""

    @staticmethod
    def forward(ctx, img, mesh, mean):
        img = img.float()
        B, S, P, C = img.shape
        clip_stride = 0
        if B > 1 and img[0].is_contiguous() and img.stride(0) >= S * P * C:
            clip_stride = img.stride(0)          # e.g. one time step of a (B, T, P, C) tensor: no copy
        else:
            img = _c(img)
        out = img.new_empty(S, mesh.N, C)
        if mesh.N > 0:
            _pool_raw(mesh, C, out, C, 0, mean, img=img, S=S, img_clip_stride=clip_stride)
        ctx.mesh, ctx.mean, ctx.shape = mesh, mean, img.shape
        return out

    @staticmethod
    def backward(ctx, g):
        mesh = ctx.mesh
        B, S, P, C = ctx.shape
        g = _c(g)
        gi = g.new_empty(S, B, P, C)
        for s in range(S):
            _gather_raw(mesh, g[s], C, ctx.mean, gi[s])
        return gi.transpose(0, 1), None, None


def pool_image(img, mesh, mean=True):
    return _PoolImage.apply(img, mesh, mean)


def pool_image_into(img, mesh, out, coff=0, mean=True):
    """Node means of img (B, S, n*m, C) written into columns [coff, coff + C) of out (S, N, W) -- data only (no autograd):
    the encoder's input rows [frame means | position | size] are assembled in place instead of concatenated."""
    assert not img.requires_grad and out.dim() == 3 and out.is_contiguous()
    img = img.float()
    B, S, P, C = img.shape
    clip_stride = 0
    if B > 1 and img[0].is_contiguous() and img.stride(0) >= S * P * C:
        clip_stride = img.stride(0)
    else:
        img = _c(img)
    if mesh.N > 0:
        _pool_raw(mesh, C, out, out.shape[2], coff, mean, img=img, S=S, img_clip_stride=clip_stride)
    return out


class _Gather(Function):
    """unflatten (model/graph_functions.py:451-458): node values (N, C) -> pixels (B, n*m, C)."""

    @staticmethod
    def forward(ctx, val, mesh):
        val = _c(val.float())
        C = val.shape[1]
        img = val.new_empty(mesh.B, mesh.P, C)
        _gather_raw(mesh, val, C, False, img)
        ctx.mesh = mesh
        return img

    @staticmethod
    def backward(ctx, g):
        mesh = ctx.mesh
        g = _c(g)
        C = g.shape[-1]
        out = g.new_empty(1, mesh.N, C)
        if mesh.N > 0:
            _pool_raw(mesh, C, out, C, 0, False, img=g.view(mesh.B, 1, mesh.P, C), S=1)
        return out[0], None


def gather_pixels(val, mesh):
    return _Gather.apply(val, mesh)


_CLIP_REMESH = os.environ.get('QT_NO_CLIP_REMESH') != '1'      # (A/B switch: 1 = the general node / tile kernels everywhere)


def _remesh_raw(dst, src, parts, outs, src_inv, mean, posfeat=None, first_only=False):
    """outs (dense (dst.N, w) matrices, side by side) = per-node reduction over dst's pixels of [parts...][src.labels[p]]
    (qt_remesh)."""
    import ctypes
    n, no = len(parts), len(outs)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in parts])
    widths = (ctypes.c_int * n)(*[t.shape[1] for t in parts])
    lds = (ctypes.c_int * n)(*[_ld(t) for t in parts])
    optrs = (ctypes.c_void_p * no)(*[t.data_ptr() for t in outs])
    owidths = (ctypes.c_int * no)(*[t.shape[1] for t in outs])
    if _CLIP_REMESH and getattr(src, 'cell_off', None) is not None:
        # the transfer of one 64 x 64 tile and one 4-channel slice runs inside one workgroup's LDS (the source mesh's nodes of a
        # tile are one row range: cell_off)
        _lib.call('qt_remesh_clip', ptrs, widths, lds, n, ptr(src.labels), ptr(src.npix), int(src_inv), ptr(src.cell_off),
                  ptr(dst.labels), ptr(dst.level), ptr(dst.npix), int(mean), dst.B, dst.n, dst.m, optrs, owidths, no,
                  ptr(posfeat), int(first_only))
        return
    assert posfeat is None and not first_only, 'the decoder-input assembly is part of the tile-resident transfer only'
    # the direct row index of the single-pixel nodes, when one mesh of the pair was decomposed from the other
    direct = None
    if dst.built_from is not None and dst.built_from() is src:
        direct = dst.fwd_src
    elif src.built_from is not None and src.built_from() is dst:
        direct = src.bwd_src
    _lib.call('qt_remesh', ptrs, widths, lds, n, ptr(src.labels), ptr(src.npix), int(src_inv), ptr(dst.labels), ptr(dst.level),
              ptr(dst.npix), int(mean), dst.B, dst.n, dst.m, dst.N, ptr(dst.cell), ptr(dst.n_dev), optrs, owidths, no, ptr(direct))


class _Remesh(Function):
    """State transfer between meshes: new node = mean over its pixels of the old node value (unflatten + flatten of
    model/seq2seq.py:440-442, 474-477 fused; no image is materialised).  The state is given as up to 8 matrices side by
    side (row-strided column views welcome) and comes back as dense matrices split by `out_widths` (gradients likewise):
    neither direction concatenates anything, and every consumer of a state part reads dense rows (as 64-byte column slices
    of one 272-byte-pitch matrix the same parts cost the gate kernels 5 % more)."""

    @staticmethod
    def forward(ctx, old, new, out_widths, dec_input, *vals):
        # dec_input: the first output part (4 wide) comes back as the decoder's next input [value | position, size]
        # (model/seq2seq.py:484-487), assembled by the transfer kernel; its gradient counts column 0 only
        parts = [_rows(v.float())[0] for v in vals]
        C = sum(t.shape[1] for t in parts)
        assert sum(out_widths) == C and len(parts) <= 8 and len(out_widths) <= 8
        assert not dec_input or (out_widths[0] == 4 and parts[0].shape[1] == 4)
        outs = [parts[0].new_empty(new.N, w) for w in out_widths]
        if new.N > 0:
            _remesh_raw(new, old, parts, outs, False, True, posfeat=new.posfeat if dec_input else None)
        ctx.old, ctx.new, ctx.in_widths, ctx.out_widths, ctx.dec = old, new, [t.shape[1] for t in parts], out_widths, dec_input
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        old, new = ctx.old, ctx.new
        ref = next(g for g in gs if g is not None)
        parts = [_rows(g)[0] if g is not None else ref.new_zeros(new.N, w) for g, w in zip(gs, ctx.out_widths)]
        gvals = [ref.new_empty(old.N, w) for w in ctx.in_widths]
        if old.N > 0:
            _remesh_raw(old, new, parts, gvals, True, False, first_only=ctx.dec)
        return (None, None, None, None, *gvals)


_DEC_FOLD = os.environ.get('QT_NO_DEC_FOLD') != '1'           # (A/B switch)


def clip_remesh_ok(old, new):
    """True when the transfer old -> new (and its backward) runs on the tile-resident kernel (both meshes keep cell_off)."""
    return _CLIP_REMESH and getattr(old, 'cell_off', None) is not None and getattr(new, 'cell_off', None) is not None


def remesh_transfer(vals, old, new, out_widths=None, dec_input=False):
    """vals: one (N_old, C) matrix or a list of column parts (widths multiples of 4); returns the transferred state as
    one matrix (out_widths None) or as the list of its column parts.  dec_input (needs clip_remesh_ok): the first part
    (4 wide) comes back as [transferred value | new.posfeat], the decoder's next input."""
    single = not isinstance(vals, (list, tuple))
    vals = [vals] if single else list(vals)
    win = [v.shape[1] for v in vals]
    C = sum(win)
    wout = list(out_widths) if out_widths is not None else [C]
    if sum(wout) != C:
        raise ValueError(f'remesh_transfer: out_widths {wout} do not add up to the {C} columns of the state')
    if any(w % 4 for w in win + wout):                   # odd widths: one matrix through qt_pool, then column views
        v = vals[0] if len(vals) == 1 else torch.cat(vals, dim=1)
        out = _RemeshOne.apply(v, old, new)
        return out if out_widths is None else list(torch.split(out, wout, dim=1))
    # qt_remesh takes <= 8 parts in and out per launch; more parts (n_layers >= 4) go across in balanced runs
    nmax = max(len(win), len(wout))
    groups = _remesh_groups(win, wout, -(-nmax // -(-nmax // 8))) or _remesh_groups(win, wout)
    outs = []
    if groups is None:       # e.g. 9 parts into one matrix: across in runs of 8 with their own widths, regrouped afterwards
        for a in range(0, len(vals), 8):
            outs += _Remesh.apply(old, new, tuple(win[a:a + 8]), dec_input and a == 0, *vals[a:a + 8])
        outs = torch.split(concat_cols(outs, new), wout, dim=1)
    else:
        for i0, i1, o0, o1 in groups:
            outs += _Remesh.apply(old, new, tuple(wout[o0:o1]), dec_input and i0 == 0 and o0 == 0, *vals[i0:i1])
    return outs[0] if out_widths is None else list(outs)


def _remesh_groups(win, wout, limit=8):
    """[(i0, i1, o0, o1)]: consecutive runs of input parts [i0, i1) and output parts [o0, o1) that cover the same columns
    with at most `limit` parts on either side (one qt_remesh launch each), chosen greedily; None when some stretch between
    two column boundaries the two splits share needs more parts than that."""
    def cum(ws):
        out = [0]
        for w in ws:
            out.append(out[-1] + w)
        return out
    ci, co = cum(win), cum(wout)
    at_i, at_o = {c: i for i, c in enumerate(ci)}, {c: i for i, c in enumerate(co)}
    common = sorted(set(ci) & set(co))
    groups, i0, o0 = [], 0, 0
    while i0 < len(win):
        best = None
        for c in common:
            if c <= ci[i0]:
                continue
            if at_i[c] - i0 > limit or at_o[c] - o0 > limit:
                break
            best = (at_i[c], at_o[c])
        if best is None:
            return None
        groups.append((i0, best[0], o0, best[1]))
        i0, o0 = best
    return groups


class _RemeshOne(Function):
    """The same for one matrix of any width (qt_pool)."""

    @staticmethod
    def forward(ctx, val, old, new):
        val = _c(val.float())
        C = val.shape[1]
        out = val.new_empty(new.N, C)
        if new.N > 0:
            _pool_raw(new, C, out, C, 0, True, src_val=val, src_mesh=old)
        ctx.old, ctx.new = old, new
        return out

    @staticmethod
    def backward(ctx, g):
        old, new = ctx.old, ctx.new
        g = _c(g)
        C = g.shape[1]
        out = g.new_empty(old.N, C)
        if old.N > 0:
            _pool_raw(old, C, out, C, 0, False, src_val=g, src_mesh=new, src_inv=True)
        return out, None, None


class _DecoderInput(Function):
    """[val4[:, 0] | posfeat] (N, 4): the decoder's next input (model/seq2seq.py:484-487) from the transferred 4-wide
    output; one kernel each way instead of slice + cat and slice_backward's zero-fill + copy."""

    @staticmethod
    def forward(ctx, val4, mesh):
        val4, ld = _rows(val4)                     # a column view of the transferred state is read in place
        out = val4.new_empty(val4.shape)           # (empty_like would keep the odd strides of a 1-row view)
        if mesh.N > 0:
            _lib.call('qt_decoder_input', ptr(val4), ld, ptr(mesh.posfeat), mesh.N, ptr(mesh.n_dev), ptr(out))
        ctx.mesh = mesh
        return out

    @staticmethod
    def backward(ctx, g):
        mesh = ctx.mesh
        g = _c(g)
        gv = g.new_empty(g.shape)
        if mesh.N > 0:
            _lib.call('qt_decoder_input', ptr(g), 4, None, mesh.N, ptr(mesh.n_dev), ptr(gv))
        return gv, None


def decoder_input(val4, mesh):
    return _DecoderInput.apply(val4, mesh)


# ------------------------------------------------------------------------------ loss
class _StepSSE(Function):
    """Per-block partial sums over clips and unmasked pixels of (out[label] - y)^2 for one output step
    (unflatten + MSELoss numerator, model/mpnnlstm.py:243-246).  The caller sums the partials of all steps once."""

    @staticmethod
    def forward(ctx, out, y, mesh):
        assert out.dtype == torch.float32 and out.is_contiguous(), 'out must be a contiguous fp32 matrix (column 0 is used)'
        # y: one output step of a (B, T, W, H, 1) tensor -- a strided view is fine (clip stride passed to the kernel)
        yv = y.reshape(mesh.B, mesh.P) if y.is_contiguous() else y
        if not (yv.dim() >= 2 and yv[0].is_contiguous() and yv.dtype == torch.float32):
            yv = y.float().contiguous().view(mesh.B, mesh.P)
        nt = -(mesh.P // -1024)
        part = out.new_empty(mesh.B * nt)
        _lib.call('qt_sse', ptr(out), out.stride(0), ptr(mesh.labels), ptr(yv), yv.stride(0), mesh.B, mesh.n, mesh.m, ptr(part))
        sy = out.new_empty(1, mesh.N, 1)
        if mesh.N > 0:
            _pool_raw(mesh, 1, sy, 1, 0, False, img=yv, S=1, img_clip_stride=yv.stride(0))
        ctx.save_for_backward(out, sy)
        ctx.mesh = mesh
        return part

    @staticmethod
    def backward(ctx, g):
        out, sy = ctx.saved_tensors
        mesh = ctx.mesh
        # every partial has the same upstream gradient (they are only ever summed); full rows: column 0 = value
        gout = torch.empty_like(out)
        if mesh.N > 0:
            _lib.call('qt_sse_bwd', ptr(out), out.stride(0), ptr(mesh.npix), ptr(sy), ptr(g.reshape(-1)[:1]), mesh.N, ptr(mesh.n_dev),
                      out.shape[1], ptr(gout))
        return gout, None, None


def _full_rows(out):
    """`out` (N, 1) as the contiguous matrix whose column 0 it is (the head's 4-wide output), if it is such a view."""
    base = out._base
    if (base is not None and base.dim() == 2 and out.dim() == 2 and out.shape[1] == 1 and base.is_contiguous()
            and base.shape[0] == out.shape[0] and out.storage_offset() == base.storage_offset()
            and out.stride(0) == base.shape[1] and base.dtype == torch.float32):
        return base
    return out


class _RolloutSSE(Function):
    """_StepSSE for all output steps of a rollout in one launch per kernel (qt_sse_rollout / _bwd): the loss touches every
    step once, after the last one, so the 3 x T_out small launches (squared error, per-node target sums for the gradient,
    the gradient rows) become 3 per 16 steps.  y: (B, T, P) contiguous fp32; outs[t]: contiguous (N_t, W) matrices whose
    column 0 is the prediction."""

    @staticmethod
    def forward(ctx, y, meshes, *outs):
        import ctypes
        B, T, P = y.shape
        m0 = meshes[0]
        nt = -(P // -1024)
        part = outs[0].new_empty(T, B * nt)
        sys_ = [o.new_empty(ms.N) for o, ms in zip(outs, meshes)]
        vp, ip = ctypes.c_void_p, ctypes.c_int
        for z0 in range(0, T, 16):
            sl = slice(z0, min(z0 + 16, T))
            n = sl.stop - sl.start
            _lib.call('qt_sse_rollout', n, (vp * n)(*[o.data_ptr() for o in outs[sl]]), (ip * n)(*[o.stride(0) for o in outs[sl]]),
                      (vp * n)(*[ms.labels.data_ptr() for ms in meshes[sl]]), (vp * n)(*[ms.level.data_ptr() for ms in meshes[sl]]),
                      (ip * n)(*[ms.N for ms in meshes[sl]]), (vp * n)(*[t.data_ptr() for t in sys_[sl]]),
                      y.data_ptr() + 4 * z0 * P, T * P, P, B, m0.n, m0.m, ptr(part[z0:]))
        ctx.save_for_backward(*outs, *sys_)
        ctx.meshes = meshes
        return part

    @staticmethod
    def backward(ctx, g):
        import ctypes
        meshes = ctx.meshes
        T = len(meshes)
        outs, sys_ = ctx.saved_tensors[:T], ctx.saved_tensors[T:]
        gouts = [torch.empty_like(o) for o in outs]
        g1 = g.reshape(-1)[:1].contiguous()      # every partial has the same upstream gradient (they are only ever summed)
        vp, ip = ctypes.c_void_p, ctypes.c_int
        W = outs[0].shape[1]
        for z0 in range(0, T, 16):
            sl = slice(z0, min(z0 + 16, T))
            n = sl.stop - sl.start
            _lib.call('qt_sse_rollout_bwd', n, (vp * n)(*[o.data_ptr() for o in outs[sl]]), (ip * n)(*[o.stride(0) for o in outs[sl]]),
                      (vp * n)(*[ms.npix.data_ptr() for ms in meshes[sl]]), (vp * n)(*[t.data_ptr() for t in sys_[sl]]),
                      (ip * n)(*[ms.N for ms in meshes[sl]]), (vp * n)(*[ms.n_dev.data_ptr() if ms.n_dev is not None else None for ms in meshes[sl]]),
                      ptr(g1), W, (vp * n)(*[t.data_ptr() for t in gouts[sl]]))
        return (None, None, *gouts)


def rollout_sse_partials(outputs, y, meshes):
    """Partial sums of the squared error of every output step (their total is the MSE numerator), or None when the steps
    cannot share the launches (masked-per-pixel preset meshes, odd layouts): the caller then goes step by step."""
    if not outputs or any(ms.loss_mask is not None or ms.N == 0 for ms in meshes):
        return None
    outs = [_full_rows(o) for o in outputs]
    W = outs[0].shape[1]
    if any(o.dtype != torch.float32 or not o.is_contiguous() or o.shape[1] != W or not o.is_cuda for o in outs):
        return None
    B, T = meshes[0].B, len(outs)
    if y.dtype != torch.float32 or not y.is_contiguous() or y.numel() != B * T * meshes[0].P:
        return None
    return _RolloutSSE.apply(y.view(B, T, meshes[0].P), tuple(meshes), *outs)


def step_sse_partials(out, y, mesh):
    """`out` (N, 1); when it is column 0 of a wider contiguous matrix (the head's 4-wide output) the op runs on that
    matrix, so the gradient is written once as full rows instead of slice_backward's zero-fill + copy."""
    if mesh.loss_mask is not None:
        # homogeneous preset mesh: partly masked cells keep all their pixels, so the mask is applied per pixel
        # (composition of differentiable ops; this preset-mesh path is not the tuned one)
        keep = (mesh.loss_mask == 0).float().view(1, mesh.P)
        img = gather_pixels(out[:, :1], mesh).view(mesh.B, mesh.P)
        d = (img - y.reshape(mesh.B, mesh.P).float()) * keep
        return (d * d).sum().view(1)
    base = out._base
    if (base is not None and base.dim() == 2 and out.dim() == 2 and out.shape[1] == 1 and base.is_contiguous()
            and base.shape[0] == out.shape[0] and out.storage_offset() == base.storage_offset()
            and out.stride(0) == base.shape[1] and base.dtype == torch.float32):
        out = base
    return _StepSSE.apply(out, y, mesh)


def step_sse(out, y, mesh):
    return _StepSSE.apply(out, y, mesh).sum()
