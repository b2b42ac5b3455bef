"""Graph-recurrent cells of the hot path: the reference's model/model.py names (GraphConv,
GConvLSTM, CONVOLUTIONS, CONVOLUTION_KWARGS ...) with the same parameters / state-dict keys, computed
by fused HIP kernels (qtmpnn.ops) on a `Mesh` instead of per-module PyG calls.

Where the reference passes (edge_index, edge_weight) these modules take the Mesh in the edge_index
slot; the Mesh already holds the ChebConv normalisation, which PyG recomputes in every call.
"""
import os

import torch
import torch.nn as nn

from qtmpnn import ops
from qtmpnn.mesh import Mesh

_ATTN_PLAN = os.environ.get('QT_NO_ATTN_PLAN') != '1'        # (A/B switch: attention models pack per tensor, torch Adam)
_MULTI_CONV = os.environ.get('QT_NO_MULTI_CONV') != '1'      # (diagnostics: one projection + attention launch pair per convolution)


class ChebConv(nn.Module):
    """Parameter layout of torch_geometric ChebConv (lins.{k}.weight (out, in) glorot, bias zeros);
    reference kwargs K=3, normalization='sym', bias=True (model/model.py:53)."""

    def __init__(self, in_channels, out_channels, K=3, normalization='sym', bias=True):
        super().__init__()
        assert normalization == 'sym', 'only the symmetric normalisation is used by the reference'
        self.in_channels, self.out_channels, self.K = in_channels, out_channels, K
        self.lins = nn.ModuleList([nn.Linear(in_channels, out_channels, bias=False) for _ in range(K)])
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        for lin in self.lins:
            nn.init.xavier_uniform_(lin.weight)

    def cheb_coeffs(self):
        """Coefficient matrices (K, in, out) of the Chebyshev series this layer applies, and its bias."""
        return torch.stack([lin.weight for lin in self.lins]).transpose(1, 2), self.bias

    def packed(self, in_pad=None, out_pad=None):
        """[W_0^T; ...; W_{K-1}^T; bias; 0 0 0] as one ((K*in_pad)+4, out_pad) matrix, zero padded."""
        cin, cout = in_pad or self.in_channels, out_pad or self.out_channels
        w, _ = self.cheb_coeffs()                                                           # (K, in, out)
        w = nn.functional.pad(w, (0, cout - self.out_channels, 0, cin - self.in_channels))
        b = self.bias if self.bias is not None else w.new_zeros(self.out_channels)
        tail = nn.functional.pad(b.unsqueeze(0), (0, cout - self.out_channels, 0, 3))        # bias row + 3 zero rows
        return torch.cat([w.reshape(self.K * cin, cout), tail], dim=0)

    # -- the same matrix as a data-movement layout over stand-in tensors (ops.PackPlan) -------------------------------
    def plan_params(self):
        return [lin.weight for lin in self.lins] + [self.bias]

    def plan_layout(self, T, fill, in_pad=None, out_pad=None):
        cin, cout = in_pad or self.in_channels, out_pad or self.out_channels
        w = torch.stack(T[:self.K]).transpose(1, 2)                                          # (K, in, out)
        w = nn.functional.pad(w, (0, cout - self.out_channels, 0, cin - self.in_channels), value=fill)
        tail = nn.functional.pad(T[self.K].unsqueeze(0), (0, cout - self.out_channels, 0, 3), value=fill)
        return torch.cat([w.reshape(self.K * cin, cout), tail], dim=0)

    def plan_layout_projected(self, T, fill):
        """A K = 3 layer with ONE output channel as the (in + 4, 4) matrix [w_0 w_1 w_2 0 ; b 0 0 0 ; 0 ...]: the operand of
        `project first, then propagate` (ops.scalar_cheb3) -- U = z @ this gives the three coefficient products and the bias."""
        assert self.K == 3 and self.out_channels == 1 and self.in_channels % 4 == 0
        w = nn.functional.pad(torch.stack(T[:3])[:, 0, :].t(), (0, 1), value=fill)             # (in, 4)
        tail = nn.functional.pad(T[3].view(1, 1), (0, 3, 0, 3), value=fill)                      # (4, 4): [b 0 0 0] + 3 zero rows
        return torch.cat([w, tail], dim=0)

    def forward(self, x, edge_index, edge_weight=None):
        mesh = _need_mesh(edge_index, x)
        pad = (-x.shape[1]) % 4
        xin = nn.functional.pad(x, (0, pad)) if pad else x
        opad = (-self.out_channels) % 4
        y = ops.cheb_poly(xin, self.packed(xin.shape[1], self.out_channels + opad), mesh, self.K, 1)
        return y[:, :self.out_channels] if opad else y


class GCNConv(ChebConv):
    """torch_geometric GCNConv(add_self_loops=False) (model/model.py:50; parameters lin.weight (out, in), bias):
    out = A^ (x W^T) + b with A^_ij = d_i^-1/2 w_ij d_j^-1/2.  The mesh weights are symmetric and its self pairs have
    weight 0, so A^ = -L^ off the diagonal: a GCNConv IS the Chebyshev series [0, -W^T] and reuses the ChebConv
    kernels, including the weight-space composition of stacked layers."""

    def __init__(self, in_channels, out_channels, add_self_loops=False):
        nn.Module.__init__(self)
        assert not add_self_loops, 'the reference uses add_self_loops=False'
        self.in_channels, self.out_channels, self.K = in_channels, out_channels, 2
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        nn.init.xavier_uniform_(self.lin.weight)

    def cheb_coeffs(self):
        wt = self.lin.weight.t()
        return torch.stack([torch.zeros_like(wt), -wt]), self.bias


class TransformerConv(nn.Module):
    """torch_geometric TransformerConv(heads=1, edge_dim=2, dropout=0.1, concat=False) (model/model.py:51): parameters
    lin_key / lin_query / lin_value (.weight (out, in), .bias), lin_edge.weight (out, 2), lin_skip (.weight, .bias).
    One projection GEMM produces [q | k | v | skip]; the edge softmax runs in the fused attention kernel, which
    recomputes the [angle, dist] edge attributes from the node centroids."""

    def __init__(self, in_channels, out_channels, heads=1, edge_dim=2, dropout=0.0, concat=False):
        super().__init__()
        assert heads == 1 and not concat and edge_dim == 2, 'the reference uses heads=1, concat=False, edge_dim=2'
        self.in_channels, self.out_channels, self.dropout = in_channels, out_channels, dropout
        self.lin_key = nn.Linear(in_channels, out_channels)
        self.lin_query = nn.Linear(in_channels, out_channels)
        self.lin_value = nn.Linear(in_channels, out_channels)
        self.lin_edge = nn.Linear(edge_dim, out_channels, bias=False)
        self.lin_skip = nn.Linear(in_channels, out_channels)
        for lin in (self.lin_key, self.lin_query, self.lin_value, self.lin_edge, self.lin_skip):
            nn.init.xavier_uniform_(lin.weight)
            if lin.bias is not None:
                nn.init.zeros_(lin.bias)

    def pack(self):
        """(W, We, accumulator) for one forward pass: the fused projection [q | k | v | skip] with its bias row, the padded
        edge weight, and the gradient accumulator all uses of this convolution in the pass share (a recurrent cell calls it
        once per time step: packing per call cost ~40 tiny kernels each time, forward + backward)."""
        cin, cout = self.in_channels, self.out_channels
        cin_p, cp = cin + (-cin) % 4, cout + (-cout) % 4
        blocks = [self.lin_query, self.lin_key, self.lin_value, self.lin_skip]
        w = torch.cat([nn.functional.pad(l.weight.t(), (0, cp - cout, 0, cin_p - cin)) for l in blocks], dim=1)   # (cin_p, 4 cp)
        b = torch.cat([nn.functional.pad(l.bias, (0, cp - cout)) for l in blocks]).unsqueeze(0)
        W = torch.cat([w, nn.functional.pad(b, (0, 0, 0, 3))], dim=0)                   # bias row + 3 zero rows
        We = nn.functional.pad(self.lin_edge.weight, (0, 0, 0, cp - cout))
        return PackedConv(W, We, ops.GradAcc(), ops.GradAcc())

    def plan_params(self):
        """The parameters in module order (= the stand-ins proj_layout receives)."""
        return [self.lin_key.weight, self.lin_key.bias, self.lin_query.weight, self.lin_query.bias, self.lin_value.weight,
                self.lin_value.bias, self.lin_edge.weight, self.lin_skip.weight, self.lin_skip.bias]

    @staticmethod
    def proj_layout(Ts, cin, cout, fill):
        """(W (n, cin_p + 4, 4 cp), We (n, cp, 2)) of n convolutions of one (in, out) shape from their plan_params() lists (or index
        stand-ins of them: data movement only, padding = `fill`, see ops.PackPlan): W[i] = [q | k | v | skip] of convolution i
        with its bias row + 3 padding rows."""
        n = len(Ts)
        cin_p, cp = cin + (-cin) % 4, cout + (-cout) % 4
        w = torch.stack([t for T in Ts for t in (T[2], T[0], T[4], T[7])]).view(n, 4, cout, cin)           # (n, 4, cout, cin)
        w = nn.functional.pad(w.permute(0, 3, 1, 2), (0, cp - cout, 0, 0, 0, cin_p - cin), value=fill)     # (n, cin_p, 4, cp)
        b = torch.stack([t for T in Ts for t in (T[3], T[1], T[5], T[8])]).view(n, 1, 4, cout)
        b = nn.functional.pad(b, (0, cp - cout, 0, 0, 0, 3), value=fill)                                    # bias row + 3 padding rows
        W = torch.cat([w, b], dim=1).reshape(n, cin_p + 4, 4 * cp)
        We = nn.functional.pad(torch.stack([T[6] for T in Ts]), (0, 0, 0, cp - cout), value=fill)          # (n, cp, 2)
        return W, We

    @staticmethod
    def stack_proj(convs):
        """(W (n, cin_p + 4, 4 cp), We (n, cp, 2)) of n convolutions of one (in, out) shape: W[i] = [q | k | v | skip] of
        convolution i with its bias row (the matrices pack() builds, in three stack / pad / cat launches)."""
        return TransformerConv.proj_layout([c.plan_params() for c in convs], convs[0].in_channels, convs[0].out_channels, 0.0)

    @staticmethod
    def pack_many(convs):
        """[PackedConv] for a list of convolutions, one batched packing per (in, out) shape: the same matrices as pack()."""
        out = [None] * len(convs)
        groups = {}
        for i, c in enumerate(convs):
            groups.setdefault((c.in_channels, c.out_channels), []).append(i)
        for (cin, cout), idxs in groups.items():
            cin_p, cp = cin + (-cin) % 4, cout + (-cout) % 4
            cs = [convs[i] for i in idxs]
            blocks = [[c.lin_query, c.lin_key, c.lin_value, c.lin_skip] for c in cs]
            w = torch.stack([l.weight for b4 in blocks for l in b4]).view(len(cs), 4, cout, cin)          # (G, 4, cout, cin)
            w = nn.functional.pad(w.permute(0, 3, 1, 2), (0, cp - cout, 0, 0, 0, cin_p - cin))             # (G, cin_p, 4, cp)
            b = torch.stack([l.bias for b4 in blocks for l in b4]).view(len(cs), 1, 4, cout)
            b = nn.functional.pad(b, (0, cp - cout, 0, 0, 0, 3))                                            # bias row + 3 zero rows
            W = torch.cat([w, b], dim=1).reshape(len(cs), cin_p + 4, 4 * cp)
            We = nn.functional.pad(torch.stack([c.lin_edge.weight for c in cs]), (0, 0, 0, cp - cout))     # (G, cp, 2)
            for i, Wk, Wek in zip(idxs, W.unbind(0), We.unbind(0)):          # (unbind: one stack in the backward)
                out[i] = PackedConv(Wk, Wek, ops.GradAcc(), ops.GradAcc())
        return out

    def forward(self, x, edge_index, edge_weight=None, packed=None):
        mesh = _need_mesh(edge_index, x)
        cin, cout = self.in_channels, self.out_channels
        cin_p, cp = cin + (-cin) % 4, cout + (-cout) % 4
        x = x[:, :cin] if x.shape[1] > cin_p else x
        if x.shape[1] < cin_p:
            x = nn.functional.pad(x, (0, cin_p - x.shape[1]))
        pc = packed if packed is not None else self.pack()
        proj = ops.cheb_poly(x, pc.W, mesh, 1, 1, acc=pc.acc if packed is not None else None)     # one GEMM: [q | k | v | skip]
        out = ops.attention(proj, pc.We, mesh, cout, self.dropout, self.training, pc.acc_e if packed is not None else None)
        return out[:, :cout] if cp != cout else out


class PackedConv:
    """Packed weights of one attention convolution for one forward pass."""
    __slots__ = ('W', 'We', 'acc', 'acc_e')

    def __init__(self, W, We, acc, acc_e):
        self.W, self.We, self.acc, self.acc_e = W, We, acc, acc_e


def _need_mesh(edge_index, *node_tensors):
    """The Mesh a module received in the reference's edge_index slot; node_tensors: (N, c) operands whose rows must be the mesh's
    nodes -- the kernels walk the mesh's rows and read these buffers unchecked."""
    if not isinstance(edge_index, Mesh):
        raise TypeError('pass the Mesh (graph_structure["mapping"]) where the reference passes edge_index: '
                        'the HIP path keeps adjacency and normalisation in the Mesh')
    for t in node_tensors:
        if t is not None and t.shape[0] != edge_index.N:
            raise ValueError(f'a node tensor of {t.shape[0]} rows for a mesh of {edge_index.N} nodes')
    return edge_index


CONVOLUTIONS = {
    'ChebConv': ChebConv,
    'GCNConv': GCNConv,
    'TransformerConv': TransformerConv,
    'MHTransformerConv': None,
    'GATConv': None,
    'GATv2Conv': None,
    'Dummy': None,
}

CONVOLUTION_KWARGS = {
    'GCNConv': dict(add_self_loops=False),
    'TransformerConv': dict(heads=1, edge_dim=2, dropout=0.1, concat=False),
    'MHTransformerConv': dict(heads=3, edge_dim=2, dropout=0.1),
    'ChebConv': dict(K=3, normalization='sym', bias=True),
    'GATConv': dict(heads=1, edge_dim=2),
    'GATv2Conv': dict(heads=1, edge_dim=2),
    'Dummy': dict(),
}


def _conv_class(convolution_type):
    assert convolution_type in CONVOLUTIONS, f'unknown convolution {convolution_type}'
    cls = CONVOLUTIONS[convolution_type]
    if cls is None:
        raise NotImplementedError(f'{convolution_type} is not built yet on the HIP path (ChebConv is; SURVEY.md 8(f))')
    return cls


class GraphConv(nn.Module):
    """n stacked convolutions with no nonlinearity in between (model/model.py:59-97)."""

    def __init__(self, convolution_type, in_channels, out_channels, n_layers):
        super().__init__()
        self.convolution_type, self.n_layers = convolution_type, n_layers
        cls, kw = _conv_class(convolution_type), CONVOLUTION_KWARGS[convolution_type]
        chans = [in_channels] + [out_channels] * n_layers
        self.convolutions = nn.ModuleList([cls(a, b, **kw) for a, b in zip(chans[:-1], chans[1:])])

    def forward(self, x, edge_index, edge_attr=None, return_attention_weights=False):
        for conv in self.convolutions:
            x = conv(x, edge_index, edge_attr)
        return x


class GConvLSTM(nn.Module):
    """Peephole graph-LSTM (model/model.py:263-463); forward returns (O, H', C') like the reference (:463).

    All eight GraphConv stacks of the reference are evaluated by ONE Chebyshev pass over Z = [X | H]:
    the stacked convolutions are pre-composed in weight space (ops.compose_chebconvs) and the four
    gates share the recurrence T_k(L^) Z.
    """

    GATES = 'ifco'

    def __init__(self, in_channels, out_channels, n_conv_layers=1, convolution_type='GCNConv', name='GConvLSTM'):
        super().__init__()
        assert convolution_type in CONVOLUTIONS
        self.convolution_type, self.n_conv_layers, self.name = convolution_type, n_conv_layers, name
        self.in_channels, self.out_channels = in_channels, out_channels
        for g in self.GATES:        # creation order = the reference's state-dict order
            setattr(self, f'conv_x_{g}', GraphConv(convolution_type, in_channels, out_channels, n_conv_layers))
            setattr(self, f'conv_h_{g}', GraphConv(convolution_type, out_channels, out_channels, n_conv_layers))
            if g != 'c':
                setattr(self, f'w_c_{g}', nn.Parameter(torch.zeros(1, out_channels)))
            setattr(self, f'b_{g}', nn.Parameter(torch.zeros(1, out_channels)))

    # -- weight packing (tiny, differentiable torch ops; once per forward pass) ----------
    def _branch(self, prefix):
        weights, biases = [], []
        for l in range(self.n_conv_layers):
            convs = [getattr(self, f'{prefix}_{g}').convolutions[l] for g in self.GATES]
            if isinstance(convs[0], GCNConv):
                weights.append(torch.stack([c.cheb_coeffs()[0] for c in convs]))
            else:
                w = torch.stack([lin.weight for c in convs for lin in c.lins])             # one copy: (4*K, h, in)
                weights.append(w.view(4, len(convs[0].lins), *w.shape[1:]).transpose(-1, -2))
            biases.append(torch.stack([c.bias for c in convs]))
        return ops.compose_chebconvs(weights, biases)          # (4, K, in, h), (4, Ks, h)

    @property
    def is_series(self):
        """True when every convolution is a Chebyshev series (ChebConv, GCNConv): the stacks compose in weight space."""
        return hasattr(self.conv_x_i.convolutions[0], 'cheb_coeffs')

    def pack(self, in_pad=None, ln=None, variants=(True,)):
        """One PackedCell per requested variant (with_h True / False); the variants share the peephole / bias
        tensors and their gradient accumulator.  W: ((K*C + Ks_padded), 4h) for Z = [X (padded to in_pad) | H]."""
        h = self.out_channels
        if not self.is_series:          # attention convolutions are nonlinear: no weight-space composition; TransformerConv stacks run
                                        # layer by layer (_pack_multi), other kinds one convolution after another
            wc = torch.cat([self.w_c_i, self.w_c_f, self.w_c_o], dim=0)
            b = torch.cat([self.b_i, self.b_f, self.b_c, self.b_o], dim=0)
            acc_p = ops.GradAcc()
            names = [f'{br}_{g}' for br in ('conv_x', 'conv_h') for g in self.GATES]
            if isinstance(self.conv_x_i.convolutions[0], TransformerConv) and _MULTI_CONV and h % 4 == 0:
                cells = [PackedCell(None, 0, 0, wc, b, ln, None, acc_p) for _ in variants]
                multi = self._pack_multi(names)
                for c in cells:
                    c.multi = multi
                return cells
            if isinstance(self.conv_x_i.convolutions[0], TransformerConv):
                # all convolutions of one shape are packed together: a handful of stack / pad / cat launches per shape
                # instead of a dozen per convolution (24+ convolutions per cell)
                flat = [(n, l, c) for n in names for l, c in enumerate(getattr(self, n).convolutions)]
                packed = TransformerConv.pack_many([c for _, _, c in flat])
                convs = {n: [None] * self.n_conv_layers for n in names}
                for (n, l, _), pc in zip(flat, packed):
                    convs[n][l] = pc
            else:
                convs = {n: [c.pack() for c in getattr(self, n).convolutions] for n in names}
            cells = [PackedCell(None, 0, 0, wc, b, ln, None, acc_p) for _ in variants]
            for c in cells:
                c.convs = convs             # the variants share the packed convolutions (and their accumulators)
            return cells
        Px, bx = self._branch('conv_x')
        Ph, bh = self._branch('conv_h')
        wc = torch.cat([self.w_c_i, self.w_c_f, self.w_c_o], dim=0)
        b = torch.cat([self.b_i, self.b_f, self.b_c, self.b_o], dim=0)
        return self._assemble(Px, bx, Ph, bh, wc, b, in_pad, ln, variants, ops.GradAcc())

    def _pack_multi(self, names):
        """Per layer ([W segments], We (8, C, 2), accumulator) for ops.multi_conv: the eight stacks run layer by layer, stack g =
        head g in the order conv_x_{i,f,c,o}, conv_h_{i,f,c,o}.  Layer 0 has two input segments (X and H: the four stacks of a
        branch share their input, so their projections are ONE matrix with 4 x 4C columns), deeper layers one (head g reads column
        block g of the previous layer's output)."""
        layers = []
        for l in range(self.n_conv_layers):
            convs = [getattr(self, n).convolutions[l] for n in names]
            if l == 0:
                Ws, Wes = [], []
                for part in (convs[:4], convs[4:]):
                    W, We = TransformerConv.stack_proj(part)                    # (4, cin_p + 4, 4C)
                    Ws.append(W.permute(1, 0, 2).reshape(1, W.shape[1], 4 * W.shape[2]))
                    Wes.append(We)
                layers.append((Ws, torch.cat(Wes, dim=0), ops.GradAcc()))
            else:
                W, We = TransformerConv.stack_proj(convs)                       # (8, C + 4, 4C)
                layers.append(([W], We, ops.GradAcc()))
        return layers

    # -- packing through one gather (ops.PackPlan): plain ChebConv stacks, and TransformerConv stacks on the layer-by-layer path ----
    @property
    def _attention_plan(self):
        return (_ATTN_PLAN and _MULTI_CONV and self.out_channels % 4 == 0 and
                all(type(c) is TransformerConv for g in self.GATES for br in ('conv_x', 'conv_h')
                    for c in getattr(self, f'{br}_{g}').convolutions))

    @property
    def plannable(self):
        return self._attention_plan or all(type(c) is ChebConv and c.bias is not None for g in self.GATES for br in ('conv_x', 'conv_h')
                                           for c in getattr(self, f'{br}_{g}').convolutions)

    def plan_params(self):
        ps = []
        for br in ('conv_x', 'conv_h'):
            for g in self.GATES:
                for conv in getattr(self, f'{br}_{g}').convolutions:
                    ps += conv.plan_params()
        return ps + [self.w_c_i, self.w_c_f, self.w_c_o, self.b_i, self.b_f, self.b_c, self.b_o]

    def _plan_layout_attention(self, T, fill, prefix):
        """The matrices of _pack_multi from stand-ins: M0x / M0h (1, cin_p + 4, 4 x 4C) and E0 (8, C, 2) for layer 0, M<l> (8, C + 4,
        4C) and E<l> for the deeper layers; wc (3, h), b (4, h)."""
        L, h, per = self.n_conv_layers, self.out_channels, 9
        conv = lambda bi, gi, l: T[((bi * 4 + gi) * L + l) * per:((bi * 4 + gi) * L + l + 1) * per]
        tail = T[2 * 4 * L * per:]
        out = {prefix + 'wc': torch.cat(tail[0:3], dim=0), prefix + 'b': torch.cat(tail[3:7], dim=0)}
        for l in range(L):
            if l == 0:
                Wes = []
                for bi, br in enumerate('xh'):
                    W, We = TransformerConv.proj_layout([conv(bi, gi, 0) for gi in range(4)], self.in_channels if bi == 0 else h, h, fill)
                    out[f'{prefix}M0{br}'] = W.permute(1, 0, 2).reshape(1, W.shape[1], 4 * W.shape[2])
                    Wes.append(We)
                out[prefix + 'E0'] = torch.cat(Wes, dim=0)
            else:
                W, We = TransformerConv.proj_layout([conv(bi, gi, l) for bi in range(2) for gi in range(4)], h, h, fill)
                out[f'{prefix}M{l}'], out[f'{prefix}E{l}'] = W, We
        return out

    def plan_layout(self, T, fill, prefix, in_pad=None, variants=(True,)):
        """Outputs (named with `prefix`): wc (3, h), b (4, h) and, for one conv layer per stack, the gate matrix W of
        every requested variant as (x-bias member, h-bias member) sums; for deeper stacks the per-layer weight / bias
        stacks of both branches, which compose_chebconvs then combines."""
        if self._attention_plan:
            return self._plan_layout_attention(T, fill, prefix)
        L, h = self.n_conv_layers, self.out_channels
        K = len(self.conv_x_i.convolutions[0].lins)
        per = K + 1

        def Wt(bi, l):       # (4, K, in, h)
            w = torch.stack([T[((bi * 4 + gi) * L + l) * per + k] for gi in range(4) for k in range(K)])
            return w.view(4, K, *w.shape[1:]).transpose(-1, -2)

        def Bs(bi, l):       # (4, h)
            return torch.stack([T[((bi * 4 + gi) * L + l) * per + K] for gi in range(4)])

        tail = T[2 * 4 * L * per:]
        out = {prefix + 'wc': torch.cat(tail[0:3], dim=0), prefix + 'b': torch.cat(tail[3:7], dim=0)}
        if L > 1:
            for bi, br in enumerate('xh'):
                for l in range(L):
                    out[f'{prefix}P{br}{l}'] = Wt(bi, l)
                    out[f'{prefix}B{br}{l}'] = Bs(bi, l)
            return out
        Px, Ph = Wt(0, 0), Wt(1, 0)
        cin = in_pad or self.in_channels
        if cin > self.in_channels:
            Px = nn.functional.pad(Px, (0, 0, 0, cin - self.in_channels), value=fill)
        rows = lambda bs: nn.functional.pad(bs.unsqueeze(1).permute(1, 0, 2).reshape(1, 4 * h), (0, 0, 0, 3), value=fill)
        for with_h in variants:
            M = torch.cat([Px, Ph], dim=2) if with_h else Px
            Wm = M.permute(1, 2, 0, 3).reshape(K * M.shape[2], 4 * h)
            out[f'{prefix}W{int(with_h)}'] = (torch.cat([Wm, rows(Bs(0, 0))], dim=0),
                                              torch.cat([torch.full_like(Wm, fill), rows(Bs(1, 0))], dim=0))
        return out

    def pack_from(self, outs, prefix, in_pad=None, ln=None, variants=(True,)):
        """PackedCells from the outputs of a plan built with plan_layout (same arguments)."""
        L, h = self.n_conv_layers, self.out_channels
        wc, b = outs[prefix + 'wc'], outs[prefix + 'b']
        acc_p = ops.GradAcc()
        if self._attention_plan:
            multi = [([outs[prefix + 'M0x'], outs[prefix + 'M0h']] if l == 0 else [outs[f'{prefix}M{l}']], outs[f'{prefix}E{l}'], ops.GradAcc())
                     for l in range(L)]
            cells = [PackedCell(None, 0, 0, wc, b, ln, None, acc_p) for _ in variants]
            for c in cells:
                c.multi = multi
            return cells
        if L == 1:
            K = len(self.conv_x_i.convolutions[0].lins)
            return [PackedCell(outs[f'{prefix}W{int(v)}'], K, 1, wc, b, ln, ops.GradAcc(), acc_p) for v in variants]
        if wc.is_cuda:           # composition and layout on the device: one launch per product (ops.compose_pack)
            stacks = [[outs[f'{prefix}{n}{br}{l}'] for l in range(L)] for n, br in (('P', 'x'), ('B', 'x'), ('P', 'h'), ('B', 'h'))]
            Ws, WTs, Kc, Ksc = ops.compose_pack(*stacks, in_pad or self.in_channels, variants)
            cells = []
            for W, WT in zip(Ws, WTs):
                acc_w = ops.GradAcc()
                acc_w.wt['T'] = WT                   # the gate GEMM stages its weight chunk from the transpose
                cells.append(PackedCell(W, Kc, Ksc, wc, b, ln, acc_w, acc_p))
            return cells
        Px, bx = ops.compose_chebconvs([outs[f'{prefix}Px{l}'] for l in range(L)], [outs[f'{prefix}Bx{l}'] for l in range(L)])
        Ph, bh = ops.compose_chebconvs([outs[f'{prefix}Ph{l}'] for l in range(L)], [outs[f'{prefix}Bh{l}'] for l in range(L)])
        return self._assemble(Px, bx, Ph, bh, wc, b, in_pad, ln, variants, acc_p)

    def _assemble(self, Px, bx, Ph, bh, wc, b, in_pad, ln, variants, acc_p):
        h = self.out_channels
        cin = in_pad or self.in_channels
        if cin > self.in_channels:
            Px = nn.functional.pad(Px, (0, 0, 0, cin - self.in_channels))
        K, Ks = Px.shape[1], bx.shape[1]
        bias_rows = nn.functional.pad((bx + ops.unalias(bh)).permute(1, 0, 2).reshape(Ks, 4 * h), (0, 0, 0, (-Ks) % 4))
        out = []
        for with_h in variants:
            M = torch.cat([Px, Ph], dim=2) if with_h else Px           # (4, K, C, h)
            W = torch.cat([M.permute(1, 2, 0, 3).reshape(K * M.shape[2], 4 * h), bias_rows], dim=0)
            out.append(PackedCell(W, K, Ks, wc, b, ln, ops.GradAcc(), acc_p))
        return out

    def step(self, X, mesh, H, C, pk, alias_h=False, pass_x=False):
        """One cell update with packed weights `pk`; pk.ln = (4, h) LayerNorm parameters fused onto H', C' or None.
        alias_h / pass_x: return H' / X once more after (O, H', C') -- see ops.gate_cell."""
        if pk.W is None and (alias_h or pass_x):
            out = tuple(self.step(X, mesh, H, C, pk))
            return out + ((out[1],) if alias_h else ()) + ((X,) if pass_x else ())
        if pk.W is None and pk.multi is not None:
            Hz = H if H is not None else X.new_zeros(X.shape[0], self.out_channels)     # conv_h(0) is not 0 (biases)
            c0 = self.conv_x_i.convolutions[0]
            cin_p = c0.in_channels + (-c0.in_channels) % 4
            if X.shape[1] != cin_p:
                X = X[:, :c0.in_channels] if X.shape[1] > cin_p else X
                X = nn.functional.pad(X, (0, cin_p - X.shape[1])) if X.shape[1] < cin_p else X
            y, L = None, len(pk.multi)
            for l, (Ws, We, acc) in enumerate(pk.multi):
                segs = [(X, Ws[0]), (Hz, Ws[1])] if l == 0 else [(y, Ws[0])]
                y = ops.multi_conv(segs, We, mesh, self.out_channels, c0.dropout, self.training, acc, gmod=4 if l == L - 1 else 0)
            return ops.lstm_cell(y, C, pk.wc, pk.b, pk.ln, mesh, pk.acc_p)
        if pk.W is None:
            Hz = H if H is not None else X.new_zeros(X.shape[0], self.out_channels)     # conv_h(0) is not 0 (biases)

            def stack(name, x):
                for conv, pc in zip(getattr(self, name).convolutions, pk.convs[name] if pk.convs else [None] * self.n_conv_layers):
                    x = conv(x, mesh, packed=pc) if pc is not None else conv(x, mesh)
                return x
            G = torch.cat([stack(f'conv_x_{g}', X) + stack(f'conv_h_{g}', Hz) for g in self.GATES], dim=1)
            return ops.lstm_cell(G, C, pk.wc, pk.b, pk.ln, mesh, pk.acc_p)
        return ops.gate_cell(X, H, pk.W, C, pk.wc, pk.b, pk.ln, mesh, pk.K, pk.Ks, pk.acc_w, pk.acc_p, alias_h, pass_x)

    def forward(self, X, edge_index, edge_weight=None, H=None, C=None):
        pad = (-X.shape[1]) % 4
        if pad:
            X = nn.functional.pad(X, (0, pad))
        return self.step(X, _need_mesh(edge_index, X, H, C), H, C, self.pack(X.shape[1], None, (H is not None,))[0])


class PackedCell:
    """Packed weights of one GConvLSTM for one forward pass (+ the gradient accumulators of that pass)."""
    __slots__ = ('W', 'K', 'Ks', 'wc', 'b', 'ln', 'acc_w', 'acc_p', 'convs', 'multi')

    def __init__(self, W, K, Ks, wc, b, ln, acc_w, acc_p):
        self.W, self.K, self.Ks, self.wc, self.b, self.ln, self.acc_w, self.acc_p = W, K, Ks, wc, b, ln, acc_w, acc_p
        self.convs = None
        self.multi = None


def _not_built(name):
    class _Missing(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f'{name} is a non-default variant of the reference and is outside the '
                                      'hot path built here (SURVEY.md section 2)')
    _Missing.__name__ = name
    return _Missing


GConvGRU = _not_built('GConvGRU')
GConvLSTM_Simple = _not_built('GConvLSTM_Simple')
SplitGConvLSTM = _not_built('SplitGConvLSTM')
DummyLSTM = _not_built('DummyLSTM')
MPNNLSTM = _not_built('MPNNLSTM')
MPNNLSTMI = _not_built('MPNNLSTMI')
MHTransformerConv = _not_built('MHTransformerConv')
