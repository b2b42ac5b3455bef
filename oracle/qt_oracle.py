"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain numpy / CPU-PyTorch restatement of the reference algorithm for the hot
path (one clip at a time, exactly like the reference).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file;
the product package (quadtree-mpnnlstm_amd/) never does.

Pinning (DESIGN.md "Oracle"):
  * graph build, flatten/unflatten, GConvLSTM / Encoder / Decoder / Seq2Seq
    control flow and the train-step loss are pinned against golden vectors
    produced by executing the reference's own modules (tests/golden/make_golden.py).
  * ChebConv / GCNConv arithmetic lives in the un-vendored third-party package
    torch-geometric==2.2.0 (requirements.txt:12), which is absent here:
    **parity unpinned** for that arithmetic.  `cheb_conv` / `gcn_conv` restate
    the library's published definitions and are cross-checked
    only against an independent dense formulation (self-consistency).

All file:line citations are into the reference tree.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

CONDITIONS = ('max_larger_than', 'max_smaller_than', 'min_larger_than', 'min_smaller_than')


# --------------------------------------------------------------------------- R0
def positional_encoding(w, h):
    """model/utils.py:37-45 -- channel 0 = col/h, channel 1 = row/w (float64)."""
    pe = np.empty((w, h, 2), dtype=np.float64)
    pe[..., 0] = (np.arange(h, dtype=np.float64) / h)[None, :]
    pe[..., 1] = (np.arange(w, dtype=np.float64) / w)[:, None]
    return pe


def add_positional_encoding(x):
    """model/utils.py:30-52.  x: (n, w, h, c) torch tensor."""
    assert x.dim() == 4
    n, w, h, _ = x.shape
    pe = torch.from_numpy(positional_encoding(w, h)).to(x.dtype)
    return torch.cat([x, pe.unsqueeze(0).expand(n, w, h, 2)], dim=-1)


# --------------------------------------------------------------------------- R1
def quadtree_decompose(img, thresh=0.05, max_size=64, mask=None, high_interest_region=None,
                       transform_func=None, condition='max_larger_than'):
    """model/graph_functions.py:145-259, restated.

    Depth-first traversal with an explicit LIFO stack; base cells are stacked
    row-major (so the last one is visited first), children are stacked
    (x,y),(x+s,y),(x,y+s),(x+s,y+s).  The split test looks at a (size+1)^2
    window whose two upper limits are BOTH clamped with the padded column
    count (:222-225), mask / high-interest windows likewise (:239-242).
    """
    assert max_size & (max_size - 1) == 0
    assert condition in CONDITIONS
    img = np.asarray(img)
    n, m = img.shape
    n_pad, m_pad = -(n // -max_size) * max_size, -(m // -max_size) * max_size
    labels = np.full((n_pad, m_pad), -1, dtype=np.int64)
    crit = np.pad(img, ((0, n_pad - n), (0, m_pad - m)), mode='edge')
    if transform_func is not None:
        crit = transform_func(crit)
    use_max = condition.startswith('max')
    larger = condition.endswith('larger_than')

    todo = [(bi * max_size, bj * max_size, max_size)
            for bi in range(n_pad // max_size) for bj in range(m_pad // max_size)]
    next_label = 0
    while todo:
        x, y, s = todo.pop()
        if x >= n or y >= m:
            continue
        if s == 1:
            if mask is not None and mask[x, y]:
                continue
            labels[x, y] = next_label
            next_label += 1
            continue
        hi_r, hi_c = min(x + s + 1, m_pad), min(y + s + 1, m_pad)
        win = crit[x:hi_r, y:hi_c]
        if win.size == 0:
            raise IndexError('empty split window (tall image: padded rows > padded cols)')
        v = win.max() if use_max else win.min()
        split = bool(v > thresh) if larger else bool(v < thresh)
        if not split and mask is not None:
            split = bool(mask[x:hi_r, y:hi_c].any())
        if not split and high_interest_region is not None:
            split = bool(high_interest_region[x:hi_r, y:hi_c].any())
        if split:
            h = s // 2
            todo.extend([(x, y, h), (x + h, y, h), (x, y + h, h), (x + h, y + h, h)])
        else:
            labels[x:x + s, y:y + s] = next_label
            next_label += 1
    return labels[:n, :m]


# --------------------------------------------------------------------------- R2
def pixel_counts(labels):
    """n_pixels_per_node of get_mapping (model/graph_functions.py:577-587)."""
    flat = labels.reshape(-1)
    flat = flat[flat >= 0]
    return np.bincount(flat).astype(np.float32)


def dense_mapping(labels):
    """The (N, P) 0/1 matrix of get_mapping + to_dense (:586, :649); small cases only."""
    flat = labels.reshape(-1)
    n_nodes = int(flat.max()) + 1
    m = np.zeros((n_nodes, flat.size), dtype=np.float32)
    valid = flat >= 0
    m[flat[valid], np.nonzero(valid)[0]] = 1.0
    return m


# --------------------------------------------------------------------------- R3 / R4
def flatten(img, labels, npix):
    """model/graph_functions.py:391-419 by labels: node value = mean of its pixels.

    img (ns, w, h, c) tensor -> (ns, N, c).  Masked pixels (label -1) take no part.
    """
    ns, w, h, c = img.shape
    lab = torch.as_tensor(labels.reshape(-1), dtype=torch.long)
    valid = lab >= 0
    flat = img.reshape(ns, w * h, c)
    out = torch.zeros(ns, npix.shape[0], c, dtype=img.dtype)
    out.index_add_(1, lab[valid], flat[:, valid, :])
    return out / torch.as_tensor(npix, dtype=img.dtype).view(1, -1, 1)


def unflatten(data, labels, image_shape):
    """model/graph_functions.py:451-458 by labels: pixel value = its node's value.

    data (..., N, c) -> (..., w, h, c); masked pixels get 0 (their mapping column is 0).
    """
    lab = torch.as_tensor(labels.reshape(-1), dtype=torch.long)
    safe = lab.clamp(min=0)
    img = data.index_select(-2, safe)
    img = img * (lab >= 0).to(data.dtype).view(*([1] * (data.dim() - 2)), -1, 1)
    return img.reshape(*data.shape[:-2], *image_shape, data.shape[-1])


# --------------------------------------------------------------------------- R5 / R6
def adjacency_sorted(labels):
    """Edge SET of get_adj (model/graph_functions.py:261-345) in canonical (src, dst) order.

    Directed pairs between 4-adjacent cells, -1 dropped, self pairs kept (the
    removal is commented out at :329-333).  The reference's order (raster scan x
    CPython set order) is not reproduced; parity is on the sorted set.
    """
    lab = np.asarray(labels)
    pairs = []
    for a, b in ((lab[:-1, :], lab[1:, :]), (lab[:, :-1], lab[:, 1:])):
        a, b = a.reshape(-1), b.reshape(-1)
        ok = (a >= 0) & (b >= 0)
        pairs.append(np.stack([a[ok], b[ok]]))
        pairs.append(np.stack([b[ok], a[ok]]))
    if not pairs:
        return np.zeros((2, 0), dtype=np.int64)
    e = np.concatenate(pairs, axis=1)
    if e.shape[1] == 0:
        return e.astype(np.int64)
    n_nodes = int(lab.max()) + 1
    key = np.unique(e[0].astype(np.int64) * n_nodes + e[1])
    return np.stack([key // n_nodes, key % n_nodes]).astype(np.int64)


def edge_dist(src, dst, xx, yy):
    """model/graph_functions.py:358-363."""
    return torch.sqrt((yy[src] - yy[dst]) ** 2 + (xx[src] - xx[dst]) ** 2)


def edge_angle(src, dst, xx, yy):
    """model/graph_functions.py:365-370."""
    return torch.atan2(xx[src] - xx[dst], yy[src] - yy[dst]) % (2 * np.pi) / (2 * np.pi)


# --------------------------------------------------------------------------- R7
def image_to_graph(img, thresh=0.05, max_grid_size=64, mask=None, high_interest_region=None,
                   transform_func=None, condition='max_larger_than', use_edge_attrs=False,
                   resolution=0.25):
    """model/graph_functions.py:590-681 (quadtree branch), label based.

    img (ns, w, h, c) tensor whose last two channels are the positional encoding.
    Returns labels, npix, data (ns, N, c+1), edge_index (2, E) sorted, edge_attrs.
    """
    assert img.dim() == 4
    if torch.isnan(img).any():
        raise ValueError('Found NaNs in image data')
    img0 = img[..., 0].max(dim=0).values.detach().numpy()
    labels = quadtree_decompose(img0, thresh=thresh, max_size=max_grid_size, mask=mask,
                                high_interest_region=high_interest_region,
                                transform_func=transform_func, condition=condition)
    npix = pixel_counts(labels)
    data = flatten(img, labels, npix)
    if torch.isnan(data).any():
        raise ValueError('Found NaNs in graph data')
    shape = img0.shape
    xx = data[0, :, -2] * shape[1] * resolution
    yy = data[0, :, -1] * shape[0] * resolution
    size = torch.as_tensor(npix) / ((max_grid_size / 2) ** 2)
    data = torch.cat([data, size.view(1, -1, 1).expand(data.shape[0], -1, 1)], dim=-1)
    ei = torch.as_tensor(adjacency_sorted(labels))
    d = edge_dist(ei[0], ei[1], xx, yy)
    attrs = torch.stack([edge_angle(ei[0], ei[1], xx, yy), d]).T if use_edge_attrs else d
    return dict(labels=labels, n_pixels_per_node=torch.as_tensor(npix), data=data,
                edge_index=ei, edge_attrs=attrs)


def pixel_graph(img, mask=None, use_edge_attrs=True, resolution=0.25):
    """image_to_graph_pixelwise + get_adj_pixelwise (model/graph_functions.py:471-539): every unmasked pixel is a node in raster
    order, node size = resolution^2, 4-neighbour edges without self pairs; edge_attrs = [angle, dist] or None (-> the convolution
    runs unweighted).  Edge list in canonical (src, dst) order."""
    ns, w, h, c = img.shape
    keep = np.ones((w, h), dtype=bool) if mask is None else ~np.asarray(mask, dtype=bool)
    labels = np.where(keep, np.cumsum(keep.reshape(-1)).reshape(w, h) - 1, -1).astype(np.int64)
    n_nodes = int(keep.sum())
    data = img[:, torch.as_tensor(keep), :]
    xx, yy = data[0, :, -2] * h * resolution, data[0, :, -1] * w * resolution
    data = torch.cat([data, torch.full((ns, n_nodes, 1), resolution ** 2, dtype=img.dtype)], dim=-1)
    ei = torch.as_tensor(adjacency_sorted(labels))
    ei = ei[:, ei[0] != ei[1]]
    attrs = torch.stack([edge_angle(ei[0], ei[1], xx, yy), edge_dist(ei[0], ei[1], xx, yy)]).T if use_edge_attrs else None
    return dict(labels=labels, n_pixels_per_node=torch.ones(n_nodes), data=data, edge_index=ei, edge_attrs=attrs, pixelwise=True)


def static_graph(image_shape, max_grid_size, mask, high_interest_region=None, use_edge_attrs=True, resolution=0.25,
                 homogeneous=False):
    """create_static_heterogeneous_graph (model/graph_functions.py:683-699): the quadtree of an all-zero image at thresh = +inf,
    i.e. cells split only where they touch the mask / the high-interest region.  homogeneous=True:
    create_static_homogeneous_graph (:707-737) -- the same without a mask (uniform cells), then every cell that lies entirely
    under the mask is deleted together with its edges and the rest renumbered in order; a partly masked cell keeps ALL its pixels."""
    w, h = image_shape
    img = add_positional_encoding(torch.zeros(1, w, h, 1))
    g = image_to_graph(img, thresh=np.inf, max_grid_size=max_grid_size, mask=None if homogeneous else mask,
                       high_interest_region=None if homogeneous else high_interest_region, use_edge_attrs=use_edge_attrs,
                       resolution=resolution)
    del g['data']
    if not homogeneous:
        return g
    labels, npix = g['labels'], g['n_pixels_per_node'].numpy()
    unmasked = np.bincount(labels.reshape(-1), weights=(~np.asarray(mask, dtype=bool)).reshape(-1).astype(np.float64),
                           minlength=len(npix))
    kept = unmasked > 0                                                        # get_nan_nodes (:701-702), inverted
    new_id = np.cumsum(kept) - 1
    ei = g['edge_index'].numpy()
    ok = kept[ei[0]] & kept[ei[1]]
    return dict(labels=np.where(kept[labels], new_id[labels], -1), n_pixels_per_node=torch.as_tensor(npix[kept]),
                edge_index=torch.as_tensor(new_id[ei[:, ok]]), edge_attrs=g['edge_attrs'][torch.as_tensor(ok)])


# --------------------------------------------------------------------------- R10 / R11
def cheb_norm(edge_index, edge_weight, n_nodes):
    """torch_geometric 2.2.0 ChebConv.__norm__ (normalization='sym', lambda_max=2.0), restated:
    remove self loops; deg = scatter_add(w, row); L = I - D^-1/2 W D^-1/2; scale 2/lambda_max;
    add self loops with fill -1.  Returns the full (index, weight) propagate list in PyG order:
    off-diagonal entries, the +1 diagonal, then the -1 diagonal.  PARITY UNPINNED (module header)."""
    row, col = edge_index
    if edge_weight is None:                       # PyG: unweighted graph -> unit weights
        edge_weight = torch.ones(row.numel())
    keep = row != col
    row, col, w = row[keep], col[keep], edge_weight[keep]
    deg = torch.zeros(n_nodes, dtype=w.dtype).index_add_(0, row, w)
    dis = deg.pow(-0.5)
    dis[torch.isinf(dis)] = 0
    off = -(dis[row] * w * dis[col])
    loop = torch.arange(n_nodes)
    idx = torch.stack([torch.cat([row, loop, loop]), torch.cat([col, loop, loop])])
    lam = torch.tensor(2.0, dtype=w.dtype)
    val = torch.cat([off, torch.ones(n_nodes, dtype=w.dtype)])
    val = (2.0 * val) / lam
    val[torch.isinf(val)] = 0
    val = torch.cat([val, -torch.ones(n_nodes, dtype=w.dtype)])
    return idx, val


def _propagate(idx, val, x):
    """aggr='add' of val * x_j at the target index (flow source_to_target)."""
    out = torch.zeros_like(x)
    return out.index_add(0, idx[1], val.view(-1, 1) * x[idx[0]])


def cheb_conv(x, edge_index, edge_weight, weights, bias):
    """torch_geometric 2.2.0 ChebConv.forward, restated: T0 = x, T1 = L^x, Tk = 2 L^ T(k-1) - T(k-2),
    out = sum_k lins[k](T_k) + bias, lins bias-free with weight (out, in)."""
    idx, val = cheb_norm(edge_index, edge_weight, x.shape[0])
    tx0 = x
    out = F.linear(tx0, weights[0])
    tx1 = x
    if len(weights) > 1:
        tx1 = _propagate(idx, val, x)
        out = out + F.linear(tx1, weights[1])
    for w in weights[2:]:
        tx2 = 2.0 * _propagate(idx, val, tx1) - tx0
        out = out + F.linear(tx2, w)
        tx0, tx1 = tx1, tx2
    return out + bias if bias is not None else out


def cheb_conv_dense(x, edge_index, edge_weight, weights, bias):
    """Independent dense formulation for the self-consistency check: L^ = -D^-1/2 W D^-1/2."""
    n = x.shape[0]
    keep = edge_index[0] != edge_index[1]
    W = torch.zeros(n, n, dtype=torch.float64)
    W[edge_index[1][keep], edge_index[0][keep]] = edge_weight[keep].double()
    deg = W.sum(0)
    dis = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))
    L = -(dis.view(-1, 1) * W * dis.view(1, -1))
    xs = [x.double(), L @ x.double()]
    for _ in weights[2:]:
        xs.append(2 * L @ xs[-1] - xs[-2])
    out = sum(t @ w.double().T for t, w in zip(xs, weights))
    return (out + bias.double()) if bias is not None else out


def gcn_conv(x, edge_index, edge_weight, weight, bias):
    """torch_geometric 2.2.0 GCNConv(add_self_loops=False): gcn_norm with deg over the target
    index, out = A^ (x W^T) + b.  PARITY UNPINNED (module header)."""
    row, col = edge_index
    n = x.shape[0]
    if edge_weight is None:                       # PyG gcn_norm: unweighted graph -> unit weights (pixelwise meshes)
        edge_weight = torch.ones(row.numel(), dtype=x.dtype)
    deg = torch.zeros(n, dtype=edge_weight.dtype).index_add_(0, col, edge_weight)
    dis = deg.pow(-0.5)
    dis[torch.isinf(dis)] = 0
    norm = dis[row] * edge_weight * dis[col]
    out = _propagate(edge_index, norm, F.linear(x, weight))
    return out + bias if bias is not None else out


def transformer_conv(x, edge_index, edge_attr, p, dropout=0.0, training=False):
    """torch_geometric 2.2.0 TransformerConv(heads=1, concat=False, beta=False, edge_dim=2, root_weight=True), restated
    from the library's published definition: q = Wq x_i + bq, k = Wk x_j + bk, v = Wv x_j + bv,
    e = We edge_attr (no bias), alpha = softmax_j(q_i . (k_j + e) / sqrt(C)) over the incoming edges of i (PyG softmax:
    exp(a - max) / (sum + 1e-16)), dropout on alpha, out_i = sum_j alpha (v_j + e) + Wskip x_i + bskip.
    PARITY UNPINNED (module header).  p: dict of the nine parameter tensors."""
    src, dst = edge_index
    n, c = x.shape[0], p['lin_query.weight'].shape[0]
    q = F.linear(x, p['lin_query.weight'], p['lin_query.bias'])
    k = F.linear(x, p['lin_key.weight'], p['lin_key.bias'])
    v = F.linear(x, p['lin_value.weight'], p['lin_value.bias'])
    e = F.linear(edge_attr, p['lin_edge.weight'])
    a = (q[dst] * (k[src] + e)).sum(-1) / np.sqrt(c)
    amax = torch.full((n,), -float('inf'), dtype=a.dtype).scatter_reduce(0, dst, a.detach(), 'amax', include_self=True)
    ex = torch.exp(a - amax[dst])
    alpha = ex / (torch.zeros(n, dtype=a.dtype).index_add(0, dst, ex)[dst] + 1e-16)
    alpha = F.dropout(alpha, dropout, training)
    out = torch.zeros(n, c, dtype=x.dtype).index_add(0, dst, alpha.unsqueeze(-1) * (v[src] + e))
    return out + F.linear(x, p['lin_skip.weight'], p['lin_skip.bias'])


class TransformerConv(nn.Module):
    """State-dict layout of PyG TransformerConv: lin_key / lin_query / lin_value (.weight, .bias), lin_edge.weight,
    lin_skip (.weight, .bias), created in that order."""

    def __init__(self, in_channels, out_channels, heads=1, edge_dim=2, dropout=0.0, concat=False):
        super().__init__()
        assert heads == 1 and not concat and edge_dim == 2
        self.dropout = dropout
        self.lin_key = nn.Linear(in_channels, out_channels)
        self.lin_query = nn.Linear(in_channels, out_channels)
        self.lin_value = nn.Linear(in_channels, out_channels)
        self.lin_edge = nn.Linear(edge_dim, out_channels, bias=False)
        self.lin_skip = nn.Linear(in_channels, out_channels)

    def forward(self, x, edge_index, edge_attr=None):
        return transformer_conv(x, edge_index, edge_attr, dict(self.named_parameters()), self.dropout, self.training)


class ChebConv(nn.Module):
    """State-dict layout of PyG ChebConv: lins.{k}.weight (out, in), bias (out,)."""

    def __init__(self, in_channels, out_channels, K=3, normalization='sym', bias=True):
        super().__init__()
        assert normalization == 'sym'
        self.lins = nn.ModuleList([nn.Linear(in_channels, out_channels, bias=False) for _ in range(K)])
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        for lin in self.lins:
            nn.init.xavier_uniform_(lin.weight)

    def forward(self, x, edge_index, edge_weight=None):
        return cheb_conv(x, edge_index, edge_weight, [lin.weight for lin in self.lins], self.bias)


class GCNConv(nn.Module):
    """State-dict layout of PyG GCNConv: lin.weight (out, in), bias (out,)."""

    def __init__(self, in_channels, out_channels, add_self_loops=False):
        super().__init__()
        assert not add_self_loops
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        nn.init.xavier_uniform_(self.lin.weight)

    def forward(self, x, edge_index, edge_weight=None):
        return gcn_conv(x, edge_index, edge_weight, self.lin.weight, self.bias)


CONVS = {'ChebConv': (ChebConv, dict(K=3, normalization='sym', bias=True)),
         'GCNConv': (GCNConv, dict(add_self_loops=False)),
         'TransformerConv': (TransformerConv, dict(heads=1, edge_dim=2, dropout=0.1, concat=False))}


# --------------------------------------------------------------------------- R9
class GraphConv(nn.Module):
    """model/model.py:59-97: n stacked convs, no nonlinearity in between."""

    def __init__(self, convolution_type, in_channels, out_channels, n_layers):
        super().__init__()
        cls, kw = CONVS[convolution_type]
        chans = [in_channels] + [out_channels] * n_layers
        self.convolutions = nn.ModuleList([cls(a, b, **kw) for a, b in zip(chans[:-1], chans[1:])])

    def forward(self, x, edge_index, edge_attr):
        for conv in self.convolutions:
            x = conv(x, edge_index, edge_attr)
        return x


# --------------------------------------------------------------------------- R12
class GConvLSTM(nn.Module):
    """model/model.py:263-463: peephole graph-LSTM; returns (O, H', C') (:463)."""

    def __init__(self, in_channels, out_channels, n_conv_layers=1, convolution_type='ChebConv'):
        super().__init__()
        for g in 'ifco':
            setattr(self, f'conv_x_{g}', GraphConv(convolution_type, in_channels, out_channels, n_conv_layers))
            setattr(self, f'conv_h_{g}', GraphConv(convolution_type, out_channels, out_channels, n_conv_layers))
            if g != 'c':
                setattr(self, f'w_c_{g}', nn.Parameter(torch.zeros(1, out_channels)))
            setattr(self, f'b_{g}', nn.Parameter(torch.zeros(1, out_channels)))
        self.out_channels = out_channels

    def forward(self, X, edge_index, edge_weight, H=None, C=None):
        if H is None:
            H = torch.zeros(X.shape[0], self.out_channels)
        if C is None:
            C = torch.zeros(X.shape[0], self.out_channels)
        g = lambda n: getattr(self, 'conv_x_' + n)(X, edge_index, edge_weight) + \
            getattr(self, 'conv_h_' + n)(H, edge_index, edge_weight)
        I = torch.sigmoid(g('i') + self.w_c_i * C + self.b_i)
        Fg = torch.sigmoid(g('f') + self.w_c_f * C + self.b_f)
        T = torch.tanh(g('c') + self.b_c)
        C = Fg * C + I * T
        O = torch.sigmoid(g('o') + self.w_c_o * C + self.b_o)
        return O, O * torch.tanh(C), C


# --------------------------------------------------------------------------- R13 / R14
class Encoder(nn.Module):
    """model/seq2seq.py:21-82.  Upper layers restart from zero state every call (:71)."""

    def __init__(self, input_features, hidden_size, n_layers, convolution_type, n_conv_layers):
        super().__init__()
        dims = [input_features] + [hidden_size] * n_layers
        self.rnns = nn.ModuleList([GConvLSTM(a, hidden_size, n_conv_layers, convolution_type) for a in dims[:-1]])
        self.norm_h = nn.LayerNorm(hidden_size)
        self.norm_c = nn.LayerNorm(hidden_size)

    def forward(self, X, edge_index, edge_weight, H=None, C=None):
        hs, cs = [], []
        inp = X
        for i, rnn in enumerate(self.rnns):
            _, h, c = rnn(inp, edge_index, edge_weight, H if i == 0 else None, C if i == 0 else None)
            h, c = self.norm_h(h), self.norm_c(c)
            hs.append(h)
            cs.append(c)
            inp = h
        return torch.stack(hs), torch.stack(cs)


class Decoder(nn.Module):
    """model/seq2seq.py:84-187.  Decoder LSTMs always use one conv layer (:106)."""

    def __init__(self, input_features, hidden_size, dropout, n_layers, concat_layers_dim, convolution_type):
        super().__init__()
        dims = [input_features] + [hidden_size] * n_layers
        self.rnns = nn.ModuleList([GConvLSTM(a, hidden_size, 1, convolution_type) for a in dims[:-1]])
        cls, kw = CONVS[convolution_type]
        self.fc_out1 = cls(hidden_size + concat_layers_dim, hidden_size, **kw)
        self.fc_out2 = cls(hidden_size, 1, **kw)
        self.norm_o = nn.LayerNorm(hidden_size)
        self.norm_h = nn.LayerNorm(hidden_size)
        self.norm_c = nn.LayerNorm(hidden_size)
        self.dropout = nn.Dropout(dropout)

    def forward(self, X, edge_index, edge_weight, concat_layers, H, C):
        hs, cs = [], []
        inp = X
        for i, rnn in enumerate(self.rnns):
            out, h, c = rnn(inp, edge_index, edge_weight, H[i], C[i])
            h, c = self.norm_h(h), self.norm_c(c)
            hs.append(h)
            cs.append(c)
            inp = h
        out = F.relu(self.norm_o(out))
        if concat_layers is not None:
            out = torch.cat([out, concat_layers], dim=-1)
        out = F.relu(self.fc_out1(out, edge_index, edge_weight))
        out = self.dropout(self.fc_out2(out, edge_index, edge_weight))
        return torch.tanh(out) + X[:, [0]], torch.stack(hs), torch.stack(cs)


# --------------------------------------------------------------------------- R15
class Seq2Seq(nn.Module):
    """model/seq2seq.py:190-527, quadtree path, teacher_forcing_ratio = 0, remesh_input = False.

    One clip per call, like the reference.  `concat_layers` is required (HEAD crashes
    without it: SURVEY.md 3.5).  forward returns (outputs, label maps, trace).
    """

    def __init__(self, hidden_size, dropout, thresh, input_timesteps=3, input_features=4,
                 output_timesteps=5, n_layers=4, n_conv_layers=2, transform_func=None,
                 condition='max_larger_than', convolution_type='ChebConv'):
        super().__init__()
        self.encoder = Encoder(input_features, hidden_size, n_layers, convolution_type, n_conv_layers)
        self.decoder = Decoder(4, hidden_size, dropout, n_layers, 1, convolution_type)
        self.thresh, self.transform_func, self.condition = thresh, transform_func, condition
        self.input_timesteps, self.output_timesteps = input_timesteps, output_timesteps
        self.use_edge_attrs = convolution_type in ('TransformerConv',)          # seq2seq.py:244-247

    def _graph(self, img, mask, hir):
        return image_to_graph(img, thresh=self.thresh, mask=mask, high_interest_region=hir,
                              transform_func=self.transform_func, condition=self.condition,
                              use_edge_attrs=self.use_edge_attrs)

    def forward(self, x, concat_layers, mask=None, high_interest_region=None, remesh_every=1,
                skip_last_remesh=True, graph_structure=None):
        w, h = x.shape[1:3]
        if graph_structure is not None:
            # preset mesh (seq2seq.py:288-294): flatten onto it, node size = n_pixels_per_node / 4 ("Don't assume 4 !!")
            npx = graph_structure['n_pixels_per_node']
            data = flatten(add_positional_encoding(x), graph_structure['labels'], npx.numpy())
            g = dict(graph_structure, data=torch.cat([data, (npx / 4.0).view(1, -1, 1).expand(data.shape[0], -1, 1)], dim=-1))
        elif self.thresh == -np.inf:                                            # graph_functions.py:629-630
            g = pixel_graph(add_positional_encoding(x), mask, self.use_edge_attrs)
        else:
            g = self._graph(add_positional_encoding(x), mask, high_interest_region)
        trace = dict(labels=[g['labels']], images=[], edge_index=[g['edge_index']])
        feats, hidden, cell = g['data'], None, None
        for t in range(self.input_timesteps):                                   # seq2seq.py:308-330
            hidden, cell = self.encoder(feats[t], g['edge_index'], g['edge_attrs'],
                                        None if hidden is None else hidden[-1],
                                        None if cell is None else cell[-1])
        xcur = feats[-1][:, [0, -3, -2, -1]]                                    # seq2seq.py:336
        outputs, maps = [], []
        for t in range(self.output_timesteps):                                  # seq2seq.py:345-396
            cl = flatten(concat_layers[t][None], g['labels'], g['n_pixels_per_node'].numpy())[0]
            out, hidden, cell = self.decoder(xcur, g['edge_index'], g['edge_attrs'], cl, hidden, cell)
            outputs.append(out)
            maps.append(g['labels'])
            last = t == self.output_timesteps - 1
            if self.thresh != -np.inf and (t + 1) % remesh_every == 0 and not (last and skip_last_remesh):  # seq2seq.py:393, 434-491
                img = unflatten(out, g['labels'], (w, h))
                h_img = unflatten(hidden, g['labels'], (w, h))
                c_img = unflatten(cell, g['labels'], (w, h))
                trace['images'].append(img.detach().numpy()[..., 0].copy())
                g = self._graph(add_positional_encoding(img[None]), mask, high_interest_region)
                trace['labels'].append(g['labels'])
                trace['edge_index'].append(g['edge_index'])
                npx = g['n_pixels_per_node'].numpy()
                hidden = flatten(h_img.swapaxes(0, -1), g['labels'], npx).swapaxes(0, -1)
                cell = flatten(c_img.swapaxes(0, -1), g['labels'], npx).swapaxes(0, -1)
                xcur = g['data'][0]
            else:                                                                # seq2seq.py:420-431
                xcur = torch.cat([out, xcur[:, 1:]], dim=-1)
        return outputs, maps, trace


# --------------------------------------------------------------------------- R16
def clip_loss(outputs, maps, y, image_shape, mask):
    """model/mpnnlstm.py:243-246: unflatten every step with its own mesh, MSE over ~mask."""
    y_hat = torch.stack([unflatten(o, lab, image_shape) for o, lab in zip(outputs, maps)])
    keep = torch.as_tensor(~np.asarray(mask, dtype=bool))
    return F.mse_loss(y_hat[:, keep], y[:, keep])


def train_step(model, optimizer, x, y, concat_layers, mask, max_norm=10.0, **forward_kw):
    """model/mpnnlstm.py:229-257 for one clip; returns the loss value."""
    optimizer.zero_grad()
    outputs, maps, _ = model(x, concat_layers, mask=mask, **forward_kw)
    loss = clip_loss(outputs, maps, y, x.shape[1:3], mask)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=max_norm)
    optimizer.step()
    return float(loss.detach())
