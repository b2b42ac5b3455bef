"""GPU parity of the individual HIP ops against the CPU oracle and the golden vectors."""
import numpy as np
import pytest
import torch

from helpers import close, dev, golden, grad_close, load_state

pytestmark = pytest.mark.gpu


def _mesh_64(seed, noise=0.0, B=1, thresh=0.1):
    from qtmpnn import synthetic
    from qtmpnn.mesh import build_mesh
    img = np.stack([synthetic.make_clip(seed + i, n_frames=1, pixel_noise=noise)[0, ..., 0] for i in range(B)])
    return build_mesh(src=torch.from_numpy(img).to(dev()), thresh=thresh), img


def _oracle_graph(mesh):
    ei = mesh.edge_index(True).cpu()
    return ei, mesh.edge_attrs(False).cpu()


def test_flatten_unflatten_golden():
    from model.graph_functions import flatten, unflatten
    from qtmpnn.mesh import build_mesh
    t = golden('transfer.npz')
    img = torch.from_numpy(t['img']).to(dev())
    mesh = build_mesh(src=img[..., 0].amax(dim=0, keepdim=True), thresh=0.1)
    assert np.array_equal(mesh.labels[0].cpu().numpy(), t['labels'])
    img.requires_grad_(True)
    flat = flatten(img, mesh, mesh.npix)
    close(flat, t['flat'])
    (gx,) = torch.autograd.grad(flat, img, torch.from_numpy(t['flat_gy']).to(dev()))
    close(gx, t['flat_gx'])
    data = torch.from_numpy(t['data']).to(dev()).requires_grad_(True)
    im = unflatten(data, mesh, (64, 64))
    close(im, t['unflat'])
    (gd,) = torch.autograd.grad(im, data, torch.from_numpy(t['unflat_gi']).to(dev()))
    close(gd, t['unflat_gd'], atol=1e-4)


@pytest.mark.parametrize('C', [1, 4, 20])
def test_spmm_vs_dense(C):
    from qtmpnn.mesh import spmm
    mesh, _ = _mesh_64(3, noise=0.02, B=2)
    rp = mesh.rowptr.long()
    rows = torch.repeat_interleave(torch.arange(mesh.N, device=dev()), rp[1:] - rp[:-1])      # CSR order
    L = torch.zeros(mesh.N, mesh.N, dtype=torch.float64, device=dev())
    L[rows, mesh.col[:mesh.E].long()] = mesh.nrm[:mesh.E].double()
    x = torch.randn(mesh.N, C, device=dev())
    p, q = torch.randn_like(x), torch.randn_like(x)
    out = torch.empty_like(x)
    spmm(mesh, x, 2.0, p, -1.0, q, 0.5, out, C)
    ref = 2.0 * (L @ x.double()) - p.double() + 0.5 * q.double()
    close(out, ref.float(), atol=1e-5)
    assert mesh.E == int((L != 0).sum())


def test_remesh_transfer_vs_oracle():
    from oracle import qt_oracle as O
    from qtmpnn import ops
    old, _ = _mesh_64(11, noise=0.0)
    new, _ = _mesh_64(12, noise=0.03)
    val = torch.randn(old.N, 8, device=dev(), requires_grad=True)
    out = ops.remesh_transfer(val, old, new)
    vc = val.detach().cpu().requires_grad_(True)
    img = O.unflatten(vc, old.labels[0].cpu().numpy(), (64, 64))
    ref = O.flatten(img[None], new.labels[0].cpu().numpy(), new.npix.cpu().numpy())[0]
    close(out, ref)
    g = torch.randn(new.N, 8)
    (gv,) = torch.autograd.grad(out, val, g.to(dev()))
    (gr,) = torch.autograd.grad(ref, vc, g)
    close(gv, gr, atol=1e-4)
    # the same state given as row-strided column parts and returned as parts: bit-identical, no concatenation
    wide = torch.randn(old.N, 20, device=dev())
    wide[:, 4:12] = val.detach()
    pa, pb = wide[:, 4:8].requires_grad_(True), wide[:, 8:12].requires_grad_(True)
    oa, ob = ops.remesh_transfer([pa, pb], old, new, [4, 4])
    assert torch.equal(torch.cat([oa, ob], dim=1), out.detach())
    ga, gb = torch.autograd.grad([oa, ob], [pa, pb], [g[:, :4].to(dev()), g[:, 4:].to(dev())])
    assert torch.equal(torch.cat([ga, gb], dim=1), gv)


@pytest.mark.parametrize('shape,B', [((64, 64), 3), ((48, 64), 2), ((24, 32), 1), ((64, 40), 2), ((128, 128), 2), ((100, 150), 1)])
def test_clip_resident_remesh_equals_general_kernels(shape, B):
    """csrc/remeshclip.hip (a clip's transfer of one 4-channel slice inside one workgroup's LDS: staged source rows, LDS gathers,
    a 64 x 64 sum pyramid) against the general node / tile kernels of transfer.hip, forward and backward (the transposed transfer:
    sums scaled by 1 / source pixel count): frames smaller than 64 x 64 and frames of several 64 x 64 tiles (a tile's source rows
    are one label range: Mesh.cell_off), a land mask, cells from 1 x 1 to a whole tile, several source and output parts with
    row-strided views.  Same sums in another association: 1e-6 relative."""
    from qtmpnn import ops
    from qtmpnn.mesh import build_mesh
    n, m = shape
    rng = np.random.default_rng(n + m + B)
    img_a = np.zeros((B, n, m), np.float32)
    img_b = np.zeros((B, n, m), np.float32)
    for b in range(B):
        img_a[b, rng.integers(0, n - 6):, :][:5, rng.integers(0, m - 8):][:, :7] = 1.0           # a few fine patches
        img_b[b] = (rng.random((n, m)) < 0.02 + 0.1 * b).astype(np.float32)
    img_b[-1] = 0.0                                                     # last clip of the new mesh: unsplit base cells (level 6 / 5)
    mask = np.zeros((n, m), bool)
    mask[n // 2:n // 2 + 5, 2:m // 2] = True
    old = build_mesh(src=torch.from_numpy(img_a).to(dev()), thresh=0.5, mask=mask)
    new = build_mesh(src=torch.from_numpy(img_b).to(dev()), thresh=0.5, mask=mask)
    assert int(new.level.max()) >= (5 if min(n, m) >= 48 else 3) and int(old.level.min()) == 0
    torch.manual_seed(0)
    wide = torch.randn(old.N, 40, device=dev())
    parts = [wide[:, 4:8].requires_grad_(True), torch.randn(old.N, 16, device=dev(), requires_grad=True), wide[:, 12:24].requires_grad_(True)]
    gouts = [torch.randn(new.N, w, device=dev()) for w in (4, 8, 20)]

    def run():
        outs = ops.remesh_transfer(parts, old, new, [4, 8, 20])
        grads = torch.autograd.grad(outs, parts, gouts)
        return [o.detach() for o in outs] + list(grads)
    assert ops._CLIP_REMESH
    fast = run()
    prev, ops._CLIP_REMESH = ops._CLIP_REMESH, False
    try:
        ref = run()
    finally:
        ops._CLIP_REMESH = prev
    for a, r in zip(fast, ref):
        close(a, r, rtol=1e-6, atol=1e-6 * float(r.abs().max()))


def test_transfer_assembles_the_next_decoder_input():
    """remesh_transfer(..., dec_input=True): the first 4-wide part comes back as [transferred value | position, size] of the new
    mesh (model/seq2seq.py:484-487) from the transfer launch itself == remesh_transfer + ops.decoder_input, bit for bit, values
    and gradients (only column 0 of that part carries a gradient back)."""
    from qtmpnn import ops
    old, _ = _mesh_64(31, noise=0.03, B=2)
    new, _ = _mesh_64(33, noise=0.0, B=2)
    assert ops.clip_remesh_ok(old, new)
    torch.manual_seed(2)
    v4 = torch.randn(old.N, 4, device=dev(), requires_grad=True)
    hs = [torch.randn(old.N, 16, device=dev(), requires_grad=True) for _ in range(2)]
    gx, gh = torch.randn(new.N, 4, device=dev()), [torch.randn(new.N, 16, device=dev()) for _ in range(2)]

    def run(fold):
        x, *parts = ops.remesh_transfer([v4, *hs], old, new, [4, 16, 16], dec_input=fold)
        if not fold:
            x = ops.decoder_input(x, new)
        grads = torch.autograd.grad([x, *parts], [v4, *hs], [gx, *gh])
        return [x.detach(), *[p.detach() for p in parts], *grads]
    a, b = run(True), run(False)
    for u, w in zip(a, b):
        assert torch.equal(u, w)
    assert torch.equal(a[0][:, 1:], new.posfeat) and bool((a[3][:, 1:] == 0).all())


@pytest.mark.parametrize('shape,B,S,C', [((64, 64), 3, 4, 1), ((48, 64), 2, 1, 5), ((24, 32), 1, 2, 3), ((100, 150), 2, 2, 2)])
def test_clip_resident_pooling_equals_general_kernels(shape, B, S, C):
    """qt_pool_clip (image -> mesh pooling of scalar channels through a 64 x 64 LDS pyramid, one workgroup per clip, frame and
    channel) against the general tile / node kernels: node means of (B, S, n*m, C) frames and the gradient back onto the pixels,
    masked meshes, frames smaller than 64 x 64.  Same sums in another association: 1e-6 relative."""
    from qtmpnn import ops
    from qtmpnn.mesh import build_mesh
    n, m = shape
    rng = np.random.default_rng(n * 7 + m + C)
    crit = (rng.random((B, n, m)) < 0.04).astype(np.float32)
    crit[-1, : n // 2] = 0.0
    mask = np.zeros((n, m), bool)
    mask[3:9, m // 2:m - 2] = True
    mesh = build_mesh(src=torch.from_numpy(crit).to(dev()), thresh=0.5, mask=mask)
    img = torch.randn(B, S, n * m, C, device=dev(), requires_grad=True)
    g = torch.randn(S, mesh.N, C, device=dev())

    def run():
        out = ops.pool_image(img, mesh, True)
        (gi,) = torch.autograd.grad(out, img, g)
        return out.detach(), gi
    fast = run()
    prev, ops._CLIP_REMESH = ops._CLIP_REMESH, False
    try:
        ref = run()
    finally:
        ops._CLIP_REMESH = prev
    for a, r in zip(fast, ref):
        close(a, r, rtol=1e-6, atol=1e-6 * float(r.abs().max()))


@pytest.mark.parametrize('n_conv', [1, 2, 3])
def test_graphconv_stack_vs_oracle(n_conv):
    """Composed Chebyshev polynomial (one kernel pass) == the oracle's sequential ChebConv stack, fwd + grads."""
    from model.model import GraphConv
    from oracle import qt_oracle as O
    from qtmpnn import ops
    mesh, _ = _mesh_64(21, noise=0.0)
    ei, ew = _oracle_graph(mesh)
    torch.manual_seed(0)
    ref = O.GraphConv('ChebConv', 8, 12, n_conv)
    for p in ref.parameters():
        p.data.normal_(0, 0.3)
    mine = GraphConv('ChebConv', 8, 12, n_conv).to(dev())
    mine.load_state_dict(ref.state_dict())
    x = torch.randn(mesh.N, 8)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr, ei, ew)
    # sequential on the GPU (module forward) and composed (one cheb_poly call)
    xg = x.to(dev()).requires_grad_(True)
    yg = mine(xg, mesh)
    close(yg, yr, atol=1e-4)
    W = [torch.stack([torch.stack([lin.weight.t() for lin in c.lins])]) for c in mine.convolutions]
    b = [c.bias.unsqueeze(0) for c in mine.convolutions]
    P, beta = ops.compose_chebconvs(W, b)
    Wfull = torch.cat([P[0].reshape(-1, 12), beta[0]], dim=0)
    xc = x.to(dev()).requires_grad_(True)
    yc = ops.cheb_poly(xc, Wfull, mesh, P.shape[1], beta.shape[1])
    close(yc, yr, atol=1e-4)
    gy = torch.randn_like(yr)
    gr = torch.autograd.grad(yr, [xr] + list(ref.parameters()), gy)
    gg = torch.autograd.grad(yc, [xc] + list(mine.parameters()), gy.to(dev()))
    for a, b_, name in zip(gg, gr, ['x'] + [k for k, _ in ref.named_parameters()]):
        grad_close(a, b_, msg=name)


@pytest.mark.parametrize('n_conv', [1, 2, 3])
def test_gconvlstm_cell_golden(n_conv):
    from model.model import GConvLSTM
    from qtmpnn import synthetic
    from qtmpnn.mesh import build_mesh
    g = golden('cells.npz')
    tag = f'nc{n_conv}'
    c = synthetic.make_clip(31, canvas=(64, 64), n_digits=1, n_frames=1, pixel_noise=0.0)
    mesh = build_mesh(src=torch.from_numpy(c[..., 0]).to(dev()), thresh=0.1)
    assert np.array_equal(mesh.labels[0].cpu().numpy(), g['labels'])
    cell = GConvLSTM(4, 8, n_conv, 'ChebConv')
    load_state(cell, g, f'{tag}_w/')
    cell.to(dev())
    X, H, C = (torch.from_numpy(g[f'{tag}_{k}']).to(dev()).requires_grad_(True) for k in 'XHC')
    Oo, Hn, Cn = cell(X, mesh, None, H, C)
    for got, name in ((Oo, 'O'), (Hn, 'Hn'), (Cn, 'Cn')):
        close(got, g[f'{tag}_{name}'], msg=name)
    names = [k for k, _ in cell.named_parameters()]
    grads = torch.autograd.grad([Oo, Hn, Cn], [X, H, C] + list(cell.parameters()),
                                [torch.from_numpy(g[f'{tag}_g{k}']).to(dev()) for k in 'OHC'])
    for got, name in zip(grads[:3], ('gX', 'gHin', 'gCin')):
        grad_close(got, g[f'{tag}_{name}'], msg=name)
    for got, k in zip(grads[3:], names):
        grad_close(got, g[f'{tag}_g/{k}'], msg=k)


def test_encoder_decoder_step_golden():
    from model.seq2seq import Decoder, Encoder
    from qtmpnn import synthetic
    from qtmpnn.mesh import build_mesh
    g = golden('cells.npz')
    c = synthetic.make_clip(31, canvas=(64, 64), n_digits=1, n_frames=1, pixel_noise=0.0)
    mesh = build_mesh(src=torch.from_numpy(c[..., 0]).to(dev()), thresh=0.1)
    enc = Encoder(4, 8, 0.0, n_layers=2, convolution_type='ChebConv', rnn_type='LSTM', n_conv_layers=2)
    load_state(enc, g, 'enc_w/')
    enc.to(dev())
    t = lambda k: torch.from_numpy(g[k]).to(dev())
    hid, cel = enc(t('enc_X'), mesh, None, H=t('enc_H'), C=t('enc_C'))
    close(hid, g['enc_hidden'])
    close(cel, g['enc_cell'])
    dec = Decoder(4, 8, 0.0, n_layers=2, concat_layers_dim=1, convolution_type='ChebConv', rnn_type='LSTM')
    load_state(dec, g, 'dec_w/')
    dec.to(dev()).eval()
    out, hid, cel = dec(t('dec_X'), mesh, None, t('dec_concat'), t('dec_H'), t('dec_C'))
    close(out, g['dec_out'])
    close(hid, g['dec_hidden'])
    close(cel, g['dec_cell'])


@pytest.mark.parametrize('h', [8, 16, 32])
def test_lstm_kernel_vs_torch(h):
    """Cell + fused LayerNorm against a plain fp32 torch reference (forward and every gradient)."""
    from qtmpnn import ops
    torch.manual_seed(h)
    N = 1000
    mk = lambda *s: torch.randn(*s, device=dev(), requires_grad=True)
    G, Cp, wc, b, ln = mk(N, 4 * h), mk(N, h), mk(3, h), mk(4, h), mk(4, h)

    def ref(G, Cp, wc, b, ln):
        gi, gf, gc, go = G.split(h, dim=1)
        I = torch.sigmoid(gi + wc[0] * Cp + b[0])
        F = torch.sigmoid(gf + wc[1] * Cp + b[1])
        T = torch.tanh(gc + b[2])
        Cr = F * Cp + I * T
        Og = torch.sigmoid(go + wc[2] * Cr + b[3])
        Hr = Og * torch.tanh(Cr)
        return (Og, torch.nn.functional.layer_norm(Hr, (h,), ln[0], ln[1], 1e-5),
                torch.nn.functional.layer_norm(Cr, (h,), ln[2], ln[3], 1e-5))
    from qtmpnn.mesh import Mesh
    outs = ops.lstm_cell(G, Cp, wc, b, ln, Mesh())
    refs = ref(G, Cp, wc, b, ln)
    for a, r in zip(outs, refs):
        close(a, r, atol=2e-5)
    gs = [torch.randn_like(o) for o in refs]
    ga = torch.autograd.grad(outs, [G, Cp, wc, b, ln], gs)
    gr = torch.autograd.grad(refs, [G, Cp, wc, b, ln], gs)
    for a, r, name in zip(ga, gr, ['G', 'Cprev', 'wc', 'b', 'ln']):
        grad_close(a, r, msg=name)


def test_head_kernel_vs_torch():
    from qtmpnn import ops
    torch.manual_seed(1)
    N, h = 777, 16
    O = torch.randn(N, h, device=dev(), requires_grad=True)
    ln = torch.randn(2, h, device=dev(), requires_grad=True)
    cc = torch.randn(N, 1, device=dev(), requires_grad=True)
    from qtmpnn.mesh import Mesh
    z = torch.cat(ops.head_input(O, ln, cc, h + 4, Mesh()), dim=1)        # two column parts: (N, h) | (N, 4)
    ref = torch.cat([torch.relu(torch.nn.functional.layer_norm(O, (h,), ln[0], ln[1], 1e-5)), cc,
                     torch.zeros(N, 3, device=dev())], dim=1)
    close(z, ref, atol=2e-5)
    g = torch.randn_like(ref)
    ga = torch.autograd.grad(z, [O, ln, cc], g)
    gr = torch.autograd.grad(ref, [O, ln, cc], g)
    for a, r, name in zip(ga, gr, ['O', 'ln_o', 'concat']):
        grad_close(a, r, msg=name)


def test_step_sse_vs_torch():
    from qtmpnn import ops
    mesh, _ = _mesh_64(41, noise=0.02, B=3)
    out = torch.randn(mesh.N, 1, device=dev(), requires_grad=True)
    y = torch.rand(3, 64 * 64, device=dev())
    sse = ops.step_sse(out, y, mesh)
    img = out[mesh.labels.reshape(3, -1).long(), 0]
    ref = ((img - y) ** 2).sum()
    close(sse, ref, rtol=1e-5)
    (ga,) = torch.autograd.grad(sse, out)
    (gr,) = torch.autograd.grad(ref, out)
    grad_close(ga, gr)


def test_static_mode_ignores_capacity_rows():
    """Static (hipGraph) mode: rows >= N of every node buffer are garbage by contract.  Poison them with NaN and
    check that cheb_poly / lstm / head / remesh results for the valid rows and every parameter gradient are
    bit-identical to the dynamic-mode results."""
    from qtmpnn import ops, synthetic
    from qtmpnn.mesh import build_mesh
    img = torch.from_numpy(np.stack([synthetic.make_clip(60 + i, n_frames=1, pixel_noise=0.02)[0, ..., 0] for i in range(2)])).to(dev())
    dyn = build_mesh(src=img, thresh=0.1)
    sta = build_mesh(src=img, thresh=0.1, static=True)
    N, cap = dyn.N, sta.N
    assert sta.n_valid == N and cap == 2 * 64 * 64 and torch.equal(dyn.labels, sta.labels)
    torch.manual_seed(0)
    C, Co, K, Ks = 20, 64, 5, 3
    Zd = torch.randn(N, C, device=dev())
    W0 = torch.randn(K * C + Ks, Co, device=dev()) * 0.2
    wc, b, ln = torch.randn(3, 16, device=dev()), torch.randn(4, 16, device=dev()), torch.randn(4, 16, device=dev())
    Cp = torch.randn(N, 16, device=dev())
    gO, gH, gC = (torch.randn(N, 16, device=dev()) for _ in range(3))

    def run(mesh, rows):
        def padn(t):
            out = torch.full((rows, *t.shape[1:]), float('nan'), device=dev())
            out[:N] = t
            return out
        Z = padn(Zd).requires_grad_(True)
        W = W0.clone().requires_grad_(True)
        params = [p.clone().requires_grad_(True) for p in (wc, b, ln)]
        cp = padn(Cp).requires_grad_(True)
        G = ops.cheb_poly(Z, W, mesh, K, Ks)
        O, Hn, Cn = ops.lstm_cell(G, cp, *params, mesh)
        loss_terms = [O, Hn, Cn]
        grads = torch.autograd.grad(loss_terms, [Z, W, cp] + params, [padn(gO), padn(gH), padn(gC)])
        return [O[:N], Hn[:N], Cn[:N], grads[0][:N], grads[1], grads[2][:N]] + list(grads[3:])
    a, s = run(dyn, N), run(sta, cap)
    names = ['O', 'Hn', 'Cn', 'gZ', 'gW', 'gCprev', 'g_wc', 'g_b', 'g_ln']
    for x, y, nm in zip(a, s, names):
        assert not torch.isnan(y).any(), f'{nm}: NaN leaked from the capacity rows'
        assert torch.equal(x, y) or torch.allclose(x, y, rtol=1e-5, atol=1e-6), nm


def test_gcnconv_cell_vs_oracle():
    """R11: GConvLSTM with GCNConv(add_self_loops=False) against the oracle's restatement (fwd + grads)."""
    from model.model import GConvLSTM
    from oracle import qt_oracle as O
    mesh, _ = _mesh_64(71, noise=0.0)
    ei, ew = _oracle_graph(mesh)
    torch.manual_seed(5)
    ref = O.GConvLSTM(4, 8, 2, 'GCNConv')
    for p in ref.parameters():
        p.data.normal_(0, 0.4)
    mine = GConvLSTM(4, 8, 2, 'GCNConv')
    mine.load_state_dict(ref.state_dict())
    mine.to(dev())
    X, H, C = torch.randn(mesh.N, 4), torch.randn(mesh.N, 8), torch.randn(mesh.N, 8)
    xr, hr, cr = (t.clone().requires_grad_(True) for t in (X, H, C))
    outs_r = ref(xr, ei, ew, hr, cr)
    xg, hg, cg = (t.to(dev()).requires_grad_(True) for t in (X, H, C))
    outs_g = mine(xg, mesh, None, hg, cg)
    for a, b in zip(outs_g, outs_r):
        close(a, b, atol=1e-4)
    gs = [torch.randn_like(o) for o in outs_r]
    gr = torch.autograd.grad(outs_r, [xr, hr, cr] + list(ref.parameters()), gs)
    gg = torch.autograd.grad(outs_g, [xg, hg, cg] + list(mine.parameters()), [g.to(dev()) for g in gs])
    for a, b, name in zip(gg, gr, ['X', 'H', 'C'] + [k for k, _ in ref.named_parameters()]):
        grad_close(a, b, msg=name)


def test_transformerconv_vs_oracle_dense():
    """Attention kernel (forward and every gradient) against the oracle's scatter formulation of TransformerConv on the
    mesh's own edge list (self pairs of multi-pixel cells included, edge attributes [angle, dist])."""
    from model.model import TransformerConv
    from oracle import qt_oracle as O
    mesh, _ = _mesh_64(81, noise=0.0, B=2)
    ei = mesh.edge_index(True).cpu()
    ea = mesh.edge_attrs(True).cpu()
    torch.manual_seed(7)
    for cin, cout in ((6, 8), (9, 1), (16, 16)):
        ref = O.TransformerConv(cin, cout)
        for p in ref.parameters():
            p.data.normal_(0, 0.5)
        mine = TransformerConv(cin, cout)
        mine.load_state_dict(ref.state_dict())
        mine.to(dev())
        x = torch.randn(mesh.N, cin)
        xr = x.clone().requires_grad_(True)
        yr = ref(xr, ei, ea)
        xg = x.to(dev()).requires_grad_(True)
        yg = mine(xg, mesh)
        close(yg, yr, atol=1e-4, msg=f'{cin}->{cout}')
        gy = torch.randn_like(yr)
        gr = torch.autograd.grad(yr, [xr] + list(ref.parameters()), gy)
        gg = torch.autograd.grad(yg, [xg] + list(mine.parameters()), gy.to(dev()))
        names = ['x'] + [k for k, _ in ref.named_parameters()]
        wscale = float(gr[names.index('lin_key.weight')].abs().max())
        for a, b, name in zip(gg, gr, names):
            if name == 'lin_key.bias':       # exactly zero in exact arithmetic (softmax shift invariance): rounding noise only
                assert float(a.abs().max()) <= 1e-4 * wscale and float(b.abs().max()) <= 1e-4 * wscale
                continue
            grad_close(a, b, msg=f'{cin}->{cout} {name}')


def test_transformer_gconvlstm_cell_golden():
    """GConvLSTM with TransformerConv stacks against the trace of the reference's cell code."""
    from model.model import GConvLSTM
    from qtmpnn import synthetic
    from qtmpnn.mesh import build_mesh
    g = golden('transformer_cell.npz')
    c = synthetic.make_clip(33, canvas=(64, 64), n_digits=1, n_frames=1, pixel_noise=0.0)
    mesh = build_mesh(src=torch.from_numpy(c[..., 0]).to(dev()), thresh=0.1)
    assert np.array_equal(mesh.labels[0].cpu().numpy(), g['labels'])
    assert np.array_equal(mesh.edge_index(True).cpu().numpy(), g['edges'])
    close(mesh.edge_attrs(True), g['attrs'], atol=2e-5)
    cell = GConvLSTM(4, 8, 2, 'TransformerConv')
    load_state(cell, g, 'w/')
    cell.to(dev()).eval()
    X, H, C = (torch.from_numpy(g[k]).to(dev()).requires_grad_(True) for k in 'XHC')
    Oo, Hn, Cn = cell(X, mesh, None, H, C)
    for got, name in ((Oo, 'O'), (Hn, 'Hn'), (Cn, 'Cn')):
        close(got, g[name], msg=name)
    grads = torch.autograd.grad([Oo, Hn, Cn], [X, H, C] + list(cell.parameters()),
                                [torch.from_numpy(g[k]).to(dev()) for k in ('gO', 'gH', 'gC')])
    for got, name in zip(grads[:3], ('gX', 'gHin', 'gCin')):
        grad_close(got, g[name], msg=name)
    for got, (k, _) in zip(grads[3:], cell.named_parameters()):
        grad_close(got, g['g/' + k], msg=k, floor=0.05 if k.endswith('lin_key.bias') else 1e-3)


@pytest.mark.parametrize('keep_h', [True, False])
def test_transformer_cell_layerwise_launches_equal_per_convolution_path(keep_h):
    """The eight TransformerConv stacks of a cell run layer by layer (qt_proj_group + one 8-head attention launch per layer,
    grouped weight gradients) must give what one projection + attention launch pair per convolution gives: same O, H', C' and the
    same gradient of every input and parameter (three layers deep, hidden 32 as ice_exp.py configures it; ragged node count)."""
    from model import model as M
    from qtmpnn import synthetic
    from qtmpnn.mesh import build_mesh
    c = synthetic.make_clip(7, canvas=(64, 64), n_digits=2, n_frames=1, pixel_noise=0.0)
    mesh = build_mesh(src=torch.from_numpy(c[..., 0]).to(dev()), thresh=0.1)
    torch.manual_seed(5)
    cell = M.GConvLSTM(5, 32, 3, 'TransformerConv').to(dev()).eval()
    for p in cell.parameters():
        torch.nn.init.normal_(p, std=0.2)
    N = mesh.N
    X = torch.randn(N, 8, device=dev(), requires_grad=True)        # (5 features padded to 8 columns, as Seq2Seq feeds them)
    H = torch.randn(N, 32, device=dev(), requires_grad=True) if keep_h else None
    C = torch.randn(N, 32, device=dev(), requires_grad=True)
    gs = [torch.randn(N, 32, device=dev()) for _ in range(3)]

    def run(multi):
        old, M._MULTI_CONV = M._MULTI_CONV, multi
        try:
            outs = cell(X, mesh, None, H, C)
            ins = [t for t in (X, H, C) if t is not None]
            grads = torch.autograd.grad(list(outs), ins + list(cell.parameters()), gs)
        finally:
            M._MULTI_CONV = old
        return [o.detach() for o in outs], grads

    from qtmpnn import ops
    hits = ops._STATS['skip_alias']
    (o1, g1), (o0, g0) = run(True), run(False)
    # the two inner layers complete the gradient array the layer above started (its skip block is the incoming gradient)
    assert ops._STATS['skip_alias'] - hits == 2
    for a, b, name in zip(o1, o0, ('O', 'Hn', 'Cn')):
        close(a, b, rtol=1e-5, atol=1e-6, msg=name)
    names = [n for n, t in (('gX', X), ('gH', H), ('gC', C)) if t is not None] + [k for k, _ in cell.named_parameters()]
    for a, b, name in zip(g1, g0, names):
        # (the key bias has an exactly-zero gradient -- a shift of every key moves all scores of a target equally --: what is computed
        # is rounding noise of the column sums, and the one-pass projection backward sums the rows in another order)
        grad_close(a, b, rtol=1e-5, rel_atol=2e-6, msg=name, floor=10.0 if name.endswith('lin_key.bias') else 1e-3)


def test_multi_head_attention_equals_separate_calls():
    """qt_attn_fwd / qt_attn_bwd with G heads in one launch, in both layouts (rows side by side; one dense plane per head and
    block): the results of G single-head calls on the same operands, bit for bit (dropout off: every head draws its own mask)."""
    from qtmpnn import _lib, synthetic
    from qtmpnn._lib import ptr
    from qtmpnn.mesh import build_mesh
    c = synthetic.make_clip(9, canvas=(64, 64), n_digits=2, n_frames=1, pixel_noise=0.0)
    mesh = build_mesh(src=torch.from_numpy(c[..., 0]).to(dev()), thresh=0.1)
    N, G, C = mesh.N, 3, 8
    xy, selfpair, eattr, rev = mesh.attn_geometry()
    E = rev.numel()
    torch.manual_seed(2)
    P = torch.randn(G, 4, N, C, device=dev())                       # planes
    Prow = P.permute(2, 0, 1, 3).reshape(N, G * 4 * C).contiguous()   # rows side by side
    We = torch.randn(G, C, 2, device=dev())
    g = torch.randn(G, N, C, device=dev())
    grow = g.permute(1, 0, 2).reshape(N, G * C).contiguous()
    nblk = _lib.value('qt_attn_blocks', N, C)
    geo = (ptr(mesh.rowptr), ptr(mesh.col), ptr(xy), ptr(eattr), ptr(selfpair))

    def run(proj, ld, ps, hs, gin, ld_g, hs_g, heads, out_shape, ld_o, hs_o, we):
        out, stats = torch.empty(out_shape, device=dev()), torch.empty(heads, N, 2, device=dev())
        _lib.call('qt_attn_fwd', *geo, ptr(proj), ld, ptr(we), C, C, N, ptr(mesh.n_dev), 1.0, 3, None, ptr(out), ptr(stats), heads, ld_o,
                  ps, hs, hs_o)
        gp, part, coef = torch.zeros_like(proj), torch.zeros(nblk, heads * 2 * C, device=dev()), torch.empty(heads, E + N, 2, device=dev())
        _lib.call('qt_attn_bwd', *geo, ptr(proj), ld, ptr(we), C, C, N, ptr(mesh.n_dev), 1.0, 3, None, ptr(gin), ld_g, ptr(stats), ptr(out),
                  ld_o, ptr(gp), ptr(part), 0, ptr(rev), ptr(coef), E, heads, 0, ps, hs, hs_g, hs_o)
        return out, gp, part.sum(0)

    o_pl, gp_pl, w_pl = run(P, C, N * C, 4 * N * C, g, C, N * C, G, (G, N, C), C, N * C, We)
    o_rw, gp_rw, w_rw = run(Prow, G * 4 * C, C, 4 * C, grow, G * C, C, G, (N, G * C), G * C, C, We)
    assert torch.equal(o_pl.permute(1, 0, 2).reshape(N, G * C), o_rw)
    assert torch.equal(gp_pl.permute(2, 0, 1, 3).reshape(N, G * 4 * C), gp_rw)
    assert torch.equal(w_pl, w_rw)
    for h in range(G):
        ph = P[h].permute(1, 0, 2).reshape(N, 4 * C).contiguous()
        o1, gp1, w1 = run(ph, 4 * C, C, 4 * C, g[h].contiguous(), C, C, 1, (N, C), C, C, We[h].contiguous())
        assert torch.equal(o1, o_pl[h]), h
        assert torch.equal(gp1.view(N, 4, C).permute(1, 0, 2), gp_pl[h]), h
        assert torch.equal(w1, w_pl[h * 2 * C:(h + 1) * 2 * C]), h


def test_multi_head_attention_dropout_draws_one_mask_per_head():
    """Heads of one launch must not share their attention-dropout mask (the reference's convolutions draw independently): with
    identical operands in every head the outputs agree without dropout and differ with it, and forward and backward use the
    same mask (the gradient of sum(out) w.r.t. v is the dropped attention weight: zero exactly where the forward dropped)."""
    from qtmpnn import ops, synthetic
    from qtmpnn.mesh import build_mesh
    c = synthetic.make_clip(11, canvas=(64, 64), n_digits=2, n_frames=1, pixel_noise=0.0)
    mesh = build_mesh(src=torch.from_numpy(c[..., 0]).to(dev()), thresh=0.1)
    N, G, C = mesh.N, 4, 8
    torch.manual_seed(6)
    one = torch.randn(N, 4 * C, device=dev())
    proj = one.repeat(1, G).requires_grad_(True)                      # rows side by side: every head sees the same q | k | v | skip
    We = torch.randn(C, 2, device=dev()).expand(G, C, 2).contiguous()
    out0 = ops.attention(proj, We, mesh, C, 0.0, False, heads=G).view(N, G, C)
    for h in range(1, G):
        assert torch.equal(out0[:, h], out0[:, 0])
    out1 = ops.attention(proj, We, mesh, C, 0.5, True, heads=G).view(N, G, C)
    assert all(not torch.equal(out1[:, h], out1[:, 0]) for h in range(1, G))
    # keep rate: mean attention mass that survives, x 1 / keep, stays near 1 (sum over edges of alpha d = 1 in expectation)
    ones_v = proj.detach().clone().view(N, G, 4, C)
    ones_v[:, :, 2] = 1.0                                             # v = 1, no edge term in the value: out - skip = sum alpha d
    zero_e = torch.zeros(G, C, 2, device=dev())
    o = ops.attention(ones_v.view(N, G * 4 * C), zero_e, mesh, C, 0.5, True, heads=G).view(N, G, C)
    mass = (o - ones_v[:, :, 3])[..., 0]                              # (N, G)
    assert abs(float(mass.mean()) - 1.0) < 0.05, float(mass.mean())


def test_proj_group_and_grouped_weight_gradient_equal_dense_calls():
    """qt_proj_group (G products in one launch, planes in / planes out) against G qt_dense2 calls, bit for bit; qt_wgrad_groups
    against the fp64 product."""
    import ctypes
    from qtmpnn import _lib
    from qtmpnn._lib import ptr
    torch.manual_seed(3)
    N, G, cin, C = 1000, 3, 8, 8
    co = 4 * C
    A = torch.randn(G, N, cin, device=dev())
    W = torch.randn(G, cin + 4, co, device=dev())
    ones = torch.zeros(N, 4, device=dev())
    ones[:, 0] = 1
    P = torch.empty(G, 4, N, C, device=dev())
    _lib.call('qt_proj_group', ptr(A), cin, N * cin, 1, cin, ptr(ones), ptr(W), None, (cin + 4) * co, G, 4, C, ptr(P), C, 4 * N * C, 0, N, None)
    for h in range(G):
        Y = torch.empty(N, co, device=dev())
        _lib.call('qt_dense2', ptr(A[h]), 0, None, None, 0, None, 1, cin, 0, ptr(W[h]), None, ptr(ones), 4, ptr(W[h][cin:]), 1, co, 0, N, None,
                  0, None, 0, None, ptr(Y), None, 0, None, None)
        assert torch.equal(P[h].permute(1, 0, 2).reshape(N, co), Y), h
    # data gradient: gA_g = gP_g W_g[:cin]^T from the planes of gP (the forward weight's rows are the transposed operand)
    gP = torch.randn_like(P)
    gA = torch.empty_like(A)
    _lib.call('qt_proj_group', ptr(gP), C, 4 * N * C, 4, C, None, None, ptr(W), (cin + 4) * co, G, 1, cin, ptr(gA), cin, N * cin, 1, N, None)
    for h in range(G):
        ref = gP[h].permute(1, 0, 2).reshape(N, co).double() @ W[h][:cin].double().t()
        close(gA[h], ref.float(), rtol=1e-5, atol=1e-5, msg=f'dgrad {h}')
    # weight gradient of all groups: [A_g | 1]^T gP_g
    Ns = (ctypes.c_int * 1)(N)
    nb = _lib.value('qt_wgrad_group_blocks', 1, Ns)
    part = torch.empty(nb, G, cin + 4, co, device=dev())
    vp = ctypes.c_void_p * 1
    _lib.call('qt_wgrad_groups', 1, vp(A.data_ptr()), (ctypes.c_int * 1)(cin), vp(ones.data_ptr()), vp(gP.data_ptr()), Ns, vp(None), cin, 4, co,
              C, C, G, cin, co, 1, ptr(part))
    gW = part.sum(0)
    for h in range(G):
        Ah = torch.cat([A[h], ones], dim=1).double()
        ref = Ah.t() @ gP[h].permute(1, 0, 2).reshape(N, co).double()
        close(gW[h], ref.float(), rtol=1e-5, atol=1e-4, msg=f'wgrad {h}')



@pytest.mark.parametrize('N,valid', [(1000, None), (4133, None), (5000, 3777), (100, None)])
def test_projection_backward_in_one_pass(N, valid):
    """qt_proj_bwd (csrc/projbwd.hip: data gradient + partial weight gradient of the grouped projection in ONE pass over the gradient
    planes, hidden size 32) against the fp64 products, and its data gradient against qt_proj_group on the same planes BIT FOR BIT
    (same reduction order on the same MFMA); two uses add into the same slabs; ragged row counts, a device-side row count
    (capacity rows beyond it are ignored and their outputs left alone)."""
    from qtmpnn import _lib
    from qtmpnn._lib import ptr
    torch.manual_seed(N)
    G, cin, C = 8, 32, 32
    co = 4 * C
    nv = valid or N
    n_dev = torch.tensor([nv], dtype=torch.int32, device=dev()) if valid else None
    nb = _lib.value('qt_proj_bwd_blocks', G)
    assert nb * G <= 2 * _lib.value('qt_num_cus')          # every workgroup resident: two per CU
    part = torch.zeros(nb, G, cin + 4, co, device=dev())
    W = torch.randn(G, cin + 4, co, device=dev()) * 0.3
    ref_w = torch.zeros(G, cin + 4, co, dtype=torch.float64, device=dev())
    for use in range(2):
        A = torch.randn(G, N, cin, device=dev())
        gP = torch.randn(G, 4, N, C, device=dev())
        nxt = torch.full((G, 4, N, cin), 7.0, device=dev())         # the data gradient lands in block 3 of the next layer's array
        _lib.call('qt_proj_bwd', ptr(gP), 4 * N * C, N * C, ptr(A), N * cin, cin, ptr(W), (cin + 4) * co, co,
                  nxt.data_ptr() + 4 * 3 * N * cin, 4 * N * cin, cin, ptr(part), N, ptr(n_dev), G, cin, C, 1, use)
        old = torch.full((G, N, cin), 7.0, device=dev())
        _lib.call('qt_proj_group', ptr(gP), C, 4 * N * C, 4, C, None, None, ptr(W), (cin + 4) * co, G, 1, cin, ptr(old), cin, N * cin, 1, N,
                  ptr(n_dev))
        assert torch.equal(nxt[:, 3], old), 'data gradient differs from qt_proj_group'
        assert (nxt[:, :3] == 7.0).all() and (nxt[:, 3, nv:] == 7.0).all()
        ones = torch.zeros(nv, 4, dtype=torch.float64, device=dev())
        ones[:, 0] = 1
        for h in range(G):
            g2 = gP[h, :, :nv].permute(1, 0, 2).reshape(nv, co).double()
            close(nxt[h, 3, :nv], (g2 @ W[h, :cin].double().t()).float(), rtol=1e-5, atol=1e-5, msg=f'dgrad {h}')
            ref_w[h] += torch.cat([A[h, :nv].double(), ones], dim=1).t() @ g2
    gW = torch.empty(G, cin + 4, co, device=dev())
    _lib.call('qt_colsum', ptr(part), nb, gW.numel(), ptr(gW))
    close(gW, ref_w.float(), rtol=1e-5, atol=2e-4, msg='weight gradient of two uses')
    assert not gW[:, cin + 1:].any()
    # accumulate == 0 overwrites the slabs
    _lib.call('qt_proj_bwd', ptr(gP), 4 * N * C, N * C, ptr(A), N * cin, cin, ptr(W), (cin + 4) * co, co,
              nxt.data_ptr() + 4 * 3 * N * cin, 4 * N * cin, cin, ptr(part), N, ptr(n_dev), G, cin, C, 0, 0)
    _lib.call('qt_colsum', ptr(part), nb, gW.numel(), ptr(gW))
    g2 = gP[:, :, :nv].permute(0, 2, 1, 3).reshape(G, nv, co).double()
    last = torch.cat([A[:, :nv].double(), ones.unsqueeze(0).expand(G, -1, -1)], dim=2).transpose(1, 2) @ g2
    close(gW, last.float(), rtol=1e-5, atol=2e-4, msg='weight gradient, overwrite mode')


@pytest.mark.parametrize('N,valid', [(1000, None), (4133, 3000)])
def test_projection_backward_in_one_pass_shared_input(N, valid):
    """qt_proj_bwd on a cell's FIRST layer: the four stacks of a segment share one input (gsA = 0) and sit side by side in one
    (36, 4 x 128) weight matrix (ldw = 512, gsW = 128), input rows possibly a column block of a wider matrix (lda); every head leaves a partial data gradient (their sum = the data gradient),
    the slabs are laid out like the weight matrix.  Against the fp64 products; two uses add into the same slabs."""
    from qtmpnn import _lib
    from qtmpnn._lib import ptr
    torch.manual_seed(N + 1)
    H, cin, C = 4, 32, 32
    co = H * 4 * C
    nv = valid or N
    n_dev = torch.tensor([nv], dtype=torch.int32, device=dev()) if valid else None
    nb = _lib.value('qt_proj_bwd_blocks', H)
    part = torch.zeros(nb, 1, cin + 4, co, device=dev())
    W = torch.randn(1, cin + 4, co, device=dev()) * 0.3
    ref_w = torch.zeros(cin + 4, co, dtype=torch.float64, device=dev())
    ones = torch.zeros(nv, 4, dtype=torch.float64, device=dev())
    ones[:, 0] = 1
    for use in range(2):
        lda = cin if use == 0 else 68               # (second use: the rows are a column block of a wider state matrix)
        A = torch.randn(N, lda, device=dev())[:, lda - cin:]
        gP = torch.randn(H, 4, N, C, device=dev())
        partial = torch.full((H, N, cin), 7.0, device=dev())
        _lib.call('qt_proj_bwd', ptr(gP), 4 * N * C, N * C, ptr(A), 0, lda, ptr(W), 4 * C, co, ptr(partial), N * cin, cin, ptr(part), N,
                  ptr(n_dev), H, cin, C, 1, use)
        assert (partial[:, nv:] == 7.0).all()
        g2 = gP[:, :, :nv].permute(2, 0, 1, 3).reshape(nv, co).double()             # row: head, block, channel = W's columns
        close(partial[:, :nv].sum(0), (g2 @ W[0, :cin].double().t()).float(), rtol=1e-5, atol=2e-5, msg='data gradient')
        ref_w += torch.cat([A[:nv].double(), ones], dim=1).t() @ g2
    gW = torch.empty(1, cin + 4, co, device=dev())
    _lib.call('qt_colsum', ptr(part), nb, gW.numel(), ptr(gW))
    close(gW[0], ref_w.float(), rtol=1e-5, atol=2e-4, msg='weight gradient of two uses')
    assert not gW[0, cin + 1:].any()


@pytest.mark.parametrize('cin_x', [8, 32])
def test_transformer_cell_hidden32_one_pass_projection_backward_equals_two_launch_path(cin_x):
    """A hidden-32 TransformerConv cell, three layers deep (what ice_exp.py:153-162 runs; cin_x = 32: an upper cell of the stack,
    whose X segment is fused as well): the backward with the one-pass projection backward (default) against the data-gradient
    launch + deferred grouped weight gradient (QT_NO_PROJ_BWD_FUSED).  Deeper layers: same reduction order, but a first-layer
    segment's data gradient is now the sum of four per-head partials -> input gradients at 1e-5 of their scale, like the weight
    gradients (another summation order over the rows)."""
    from model.model import GConvLSTM
    from qtmpnn import ops
    mesh, _ = _tile_mesh('mnist128_sparse', 1)
    N = mesh.N
    torch.manual_seed(4)
    cell = GConvLSTM(cin_x, 32, n_conv_layers=3, convolution_type='TransformerConv').to(dev()).eval()
    X, H, Cc = (torch.randn(N, w, device=dev(), requires_grad=True) for w in (cin_x, 32, 32))
    gO, gH, gC = (torch.randn(N, 32, device=dev()) for _ in range(3))
    res = {}
    for flag in (True, False):
        prev, ops._PROJ_BWD_FUSED = ops._PROJ_BWD_FUSED, flag
        try:
            Oo, Hn, Cn = cell(X, mesh, None, H, Cc)
            grads = torch.autograd.grad([Oo, Hn, Cn], [X, H, Cc] + list(cell.parameters()), [gO, gH, gC], allow_unused=True)
        finally:
            ops._PROJ_BWD_FUSED = prev
        res[flag] = grads
    names = ['X', 'H', 'C'] + [k for k, _ in cell.named_parameters()]
    for i, (a, b, name) in enumerate(zip(res[True], res[False], names)):
        if a is None or b is None:
            assert a is None and b is None
            continue
        if name == 'C':
            assert torch.equal(a, b), f'input gradient {name}'
        elif name.endswith('lin_key.bias'):       # an exact zero: both are rounding noise of column sums over N rows of O(1) terms
            assert float(a.abs().max()) < 2e-5 and float(b.abs().max()) < 2e-5, name
        else:
            close(a, b, rtol=1e-5, atol=1e-5 * max(float(b.abs().max()), 1e-3), msg=name)


@pytest.mark.parametrize('K,Cb,Cbb,Kb', [(128, 8, 32, 7), (64, 16, 0, 3), (128, 32, 0, 1)])
def test_split_bf16_data_gradient_gemm(K, Cb, Cbb, Kb):
    """qt_dense_sb (the data gradient of wide gate matrices as a split-bf16 product: gradients only) against the fp64 product:
    relative error of the 2-term split, ~2^-16 per product, far inside the gradient tolerance (1e-4); planes in two parts, ragged
    row count, device-side row count."""
    from qtmpnn import _lib
    from qtmpnn._lib import ptr
    torch.manual_seed(8)
    N, NB = 3001, Kb * (Cb + Cbb)
    A = torch.randn(N + 50, K, device=dev())
    Wt = torch.randn(NB, K, device=dev()) * 0.3                  # B^T: row j = the K coefficients of output column j
    hi, lo = (torch.empty(NB, K, dtype=torch.bfloat16, device=dev()) for _ in range(2))
    _lib.call('qt_split_bf16', ptr(Wt), Wt.numel(), ptr(hi), ptr(lo))
    out = torch.full((Kb, N + 50, Cb), 7.0, device=dev())
    outb = torch.full((Kb, N + 50, Cbb), 7.0, device=dev()) if Cbb else None
    n_dev = torch.tensor([N], dtype=torch.int32, device=dev())
    _lib.call('qt_dense_sb', ptr(A), 0, K, ptr(hi), ptr(lo), Kb, Cb, Cbb, N + 50, ptr(n_dev), ptr(out), ptr(outb))
    ref = (A[:N].double() @ Wt.double().t()).view(N, Kb, Cb + Cbb).permute(1, 0, 2)
    scale = float(ref.abs().max())
    err = float((out[:, :N].double() - ref[..., :Cb]).abs().max()) / scale
    assert err < 2e-5, err
    assert bool((out[:, N:] == 7.0).all())                       # rows past the device-side count stay untouched
    if Cbb:
        errb = float((outb[:, :N].double() - ref[..., Cb:]).abs().max()) / scale
        assert errb < 2e-5, errb


def test_bf16x3_gemm_matches_fp32_gemm():
    """Opt-in bf16x3 GEMM (QT_GEMM_BF16X3=1; three bf16 terms per operand, six MFMAs per product group) must agree with
    the default exact-fp32 MFMA GEMM to fp32 rounding level.  Run in a child process: the switch is read once per process."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent('''
        import sys, os, torch
        sys.path.insert(0, os.path.join(os.getcwd(), 'quadtree-mpnnlstm_amd'))
        from qtmpnn import _lib
        from qtmpnn._lib import ptr
        torch.manual_seed(0)
        dev = torch.device('cuda', 0)
        N, K, C, Co = 5000, 5, 20, 64
        Z = torch.randn(N, C, device=dev); TZ = torch.randn(K - 1, N, C, device=dev)
        S = torch.randn(N, 4, device=dev); W = torch.randn(K * C + 4, Co, device=dev); Y = torch.empty(N, Co, device=dev)
        _lib.call('qt_dense', ptr(Z), ptr(TZ), K, C, ptr(W), ptr(S), 4, ptr(W[K * C:]), 1, Co, N, None, 0, None, 0, None, ptr(Y))
        A = torch.cat([Z] + list(TZ) + [S], dim=1).double()
        ref = (A @ W.double())
        err = ((Y.double() - ref).abs().max() / ref.abs().max()).item()
        print('RELERR', err)
    ''')
    errs = {}
    for mode in ('0', '1'):
        env = dict(os.environ)
        env.pop('QT_GEMM_BF16X3', None)
        if mode == '1':
            env['QT_GEMM_BF16X3'] = '1'
        out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300,
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert out.returncode == 0, out.stderr[-2000:]
        errs[mode] = float(out.stdout.split('RELERR')[1])
    assert errs['0'] < 2e-6 and errs['1'] < 2e-6, errs


@pytest.mark.parametrize('with_c,h,K', [(True, 16, 3), (False, 16, 3), (True, 8, 3), (True, 32, 3), (True, 32, 7), (False, 32, 5)])
def test_fused_gate_cell_equals_gemm_then_cell(with_c, h, K):
    """qt_dense_lstm (the cell as the gate GEMM's epilogue, h = 8 / 16 / 32) on Z given as two row-strided column views [X | H]
    == qt_dense on the whole Z followed by qt_lstm_fwd, bit for bit, forward and every gradient; a node count that is
    not a multiple of the 128-row tile."""
    from qtmpnn import ops
    mesh, _ = _mesh_64(5, noise=0.03, B=2)
    torch.manual_seed(3)
    C, Ks = (8 if K > 3 else 4) + h, 1          # (hidden 32, K = 7, 8 + 32 channels: the 280 output columns of BASELINE configs[3] / [4])
    cx = C - h
    mk = lambda *s: torch.randn(*s, device=dev()).requires_grad_(True)
    Z, W = mk(mesh.N, C), mk(K * C + 4, 4 * h)
    Cp = mk(mesh.N, h) if with_c else None
    wc, b, ln = mk(3, h), mk(4, h), mk(4, h)
    gs = [torch.randn(mesh.N, h, device=dev()) for _ in range(3)]
    ins = [t for t in (Z, W, Cp, wc, b, ln) if t is not None]

    def run(fused):
        if fused:
            outs = ops.gate_cell(Z[:, :cx], Z[:, cx:], W, Cp, wc, b, ln, mesh, K, Ks)
        else:
            outs = ops.lstm_cell(ops.cheb_poly(Z, W, mesh, K, Ks), Cp, wc, b, ln, mesh)
        grads = torch.autograd.grad(outs, ins, gs)
        return [o.detach() for o in outs] + list(grads)

    import os
    names = ['O', 'H', 'C'] + [n for n, t in zip(('gZ', 'gW', 'gCp', 'gwc', 'gb', 'gln'), (Z, W, Cp, wc, b, ln)) if t is not None]
    os.environ['QT_NO_DGRAD_FUSION'] = '1'            # backward as qt_lstm_bwd + qt_dense2: everything bit for bit
    try:
        for a, r, n in zip(run(True), run(False), names):
            assert torch.equal(a, r), n
    finally:
        del os.environ['QT_NO_DGRAD_FUSION']
    # default backward: the cell backward and the data gradient in one launch (qt_lstm_bwd_dgrad, h = 8 / 16 / 32).  The forward and
    # the state / weight gradients stay bit-identical; the parameter partials are summed per 128-node workgroup instead of
    # per grid-stride sweep (fp32 rounding); the data gradient is the exact fp32 product (the split-bf16 form is opt-in:
    # ops.DGRAD_SPLIT_BF16, tests/test_gpu_headline.py).
    for a, r, n in zip(run(True), run(False), names):
        if n in ('gwc', 'gb', 'gln'):
            close(a, r, 1e-4, 1e-4 * float(r.abs().max()), msg=n)
        else:
            assert torch.equal(a, r), n


def test_spmm_two_row_strided_parts_equal_one_matrix():
    """qt_spmm2 on [Xa | Xb] given as row-strided column views (with strided addends) == qt_spmm on the concatenated
    matrix, bit for bit: the parts only change where a row's floats live."""
    from qtmpnn.mesh import spmm, spmm2
    mesh, _ = _mesh_64(7, noise=0.03, B=2)
    torch.manual_seed(0)
    N = mesh.N
    wide_x, wide_p = torch.randn(N, 36, device=dev()), torch.randn(N, 28, device=dev())
    xa, xb = wide_x[:, 8:12], wide_x[:, 16:32]               # widths 4 and 16, row stride 36
    pa, pb = wide_p[:, 4:8], wide_p[:, 12:28]
    qa, qb = torch.randn(N, 4, device=dev()), torch.randn(N, 16, device=dev())
    oa, ob = torch.empty(N, 4, device=dev()), torch.empty(N, 16, device=dev())
    spmm2(mesh, [xa, xb], 2.0, [pa, pb], -1.0, [qa, qb], 0.5, [oa, ob])
    x, p, q = (torch.cat(t, dim=1).contiguous() for t in ((xa, xb), (pa, pb), (qa, qb)))
    ref = torch.empty(N, 20, device=dev())
    spmm(mesh, x, 2.0, p, -1.0, q, 0.5, ref, 20)
    assert torch.equal(torch.cat([oa, ob], dim=1), ref)


@pytest.mark.parametrize('clip_width', [4, 2, 0])
@pytest.mark.parametrize('K,widths,B', [(5, (4, 16), 3), (3, (16, 4), 2), (5, (16,), 1), (2, (8,), 2), (3, (32,), 2)])
def test_clip_resident_recurrence_equals_per_hop_launches(K, widths, B, clip_width):
    """csrc/chebclip.hip (all hops of a ChebConv recurrence in one launch, a clip's rows in LDS) against one qt_spmm2 launch per
    hop: forward planes T_1 .. T_{K-1} and the Clenshaw backward, bit for bit; row-strided column views as Z; a mesh with big
    cells beside small ones (rows with more than four edges take the CSR tail); static capacities (node counts on the device,
    capacity rows poisoned with NaN) give the same valid rows.  Both slice widths of the kernel (4 and 2 channels per
    workgroup; the `width` argument of the entry points) and the automatic choice."""
    from qtmpnn import _lib, ops
    from qtmpnn._lib import ptr
    from qtmpnn.mesh import spmm2
    mesh, _ = _mesh_64(11, noise=0.0, B=B)                       # sparse mesh: cells of 1 .. 32 pixels side by side
    deg = (mesh.rowptr[1:] - mesh.rowptr[:-1])
    assert int(deg.max()) > 4 and ops._clip_resident(mesh, list(widths), max(K, ops._CLIP_MIN_K))
    torch.manual_seed(K)
    N = mesh.N
    wide = torch.randn(N, sum(widths) + 8, device=dev())
    Zs, o = [], 4
    for w in widths:
        Zs.append(wide[:, o:o + w])
        o += w
    fused = [torch.empty(K - 1, N, w, device=dev()) for w in widths]
    ops.clip_planes(mesh, Zs, fused, K, width=clip_width)
    prev, ops._CLIP_CHEB = ops._CLIP_CHEB, False          # one qt_spmm2 launch per hop
    try:
        ref, sm = ops._cheb_planes(Zs, mesh, K)
        assert sm == 0
    finally:
        ops._CLIP_CHEB = prev
    for a, r in zip(fused, ref):            # (the fused launch stores its planes slice-major)
        assert torch.equal(ops.planes_rowmajor(a, 1), r)
    # backward: Clenshaw on random gradient planes
    G = [torch.randn(K, N, w, device=dev()) for w in widths]
    Gf = [g.clone() for g in G]
    ops.clip_clenshaw(mesh, Gf, K, width=clip_width)
    Gr = [g.clone() for g in G]
    for k in range(K - 2, 0, -1):
        spmm2(mesh, [g[k + 1] for g in Gr], 2.0, [g[k] for g in Gr], 1.0, [g[k + 2] for g in Gr] if k + 2 < K else None, -1.0,
              [g[k] for g in Gr])
    spmm2(mesh, [g[1] for g in Gr], 1.0, [g[0] for g in Gr], 1.0, [g[2] for g in Gr] if K > 2 else None, -1.0, [g[0] for g in Gr])
    for a, r, g0 in zip(Gf, Gr, G):
        assert torch.equal(a[0], r[0])
        assert torch.equal(a[1:], g0[1:])                          # the fused launch leaves planes 1 .. K-1 as given
    # the same with planes 1 .. K-1 stored slice-major, (C / 4, N, 4) each -- what the data-gradient kernels write on request
    Gs = []
    for g0, w in zip(G, widths):
        t = g0.clone()
        t[1:] = g0[1:].view(K - 1, N, w // 4, 4).permute(0, 2, 1, 3).reshape(K - 1, N, w)
        Gs.append(t)
    ops.clip_clenshaw(mesh, Gs, K, sm=1, width=clip_width)
    for a, r in zip(Gs, Gr):
        assert torch.equal(a[0], r[0])


def test_clip_resident_pool_overflow_walks_the_csr():
    """csrc/chebclip.hip's pool-full path (a clip with more tail edges than QT_TAIL_CAP: the rows whose run did not fit carry
    info base 0xffff and walk the CSR arrays in every hop) cannot be reached with the shipped capacity on ordinary meshes, so the
    Makefile also builds the library with QT_TAIL_CAP = 48 (libqtmpnn_hip_smallcaps.so).  A child process loads THAT build and
    checks forward planes and Clenshaw backward against the per-hop launches bit for bit (tests/_clip_overflow_child.py)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(os.path.dirname(here), 'quadtree-mpnnlstm_amd', 'qtmpnn', 'libqtmpnn_hip_smallcaps.so')
    assert os.path.exists(lib), 'run __graft_entry__.build() (make -C quadtree-mpnnlstm_amd/csrc)'
    env = dict(os.environ, QT_LIB_PATH=lib)
    r = subprocess.run([sys.executable, os.path.join(here, '_clip_overflow_child.py')], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'overflow ok' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def _tile_mesh(kind, B, static=False):
    """Meshes of several 64 x 64 base cells: (mesh, criterion image)."""
    from qtmpnn import synthetic
    from qtmpnn.mesh import build_mesh
    from helpers import dist_from_05
    mask = tf = None
    thresh = 0.1
    if kind == 'mnist128_sparse':                      # big cells across tile borders
        img = np.stack([synthetic.make_clip(30 + i, canvas=(128, 128), n_digits=2, n_frames=1, pixel_noise=0.0)[0, ..., 0] for i in range(B)])
    elif kind == 'mnist128_noisy':                     # nearly one node per pixel: tiles of ~4096 rows, 250 halo rows each
        img = np.stack([synthetic.make_clip(40 + i, canvas=(128, 128), n_digits=2, n_frames=1, pixel_noise=0.05)[0, ..., 0] for i in range(B)])
    elif kind == 'wide64x128':
        img = np.stack([synthetic.make_clip(50 + i, canvas=(128, 64), n_digits=1, n_frames=1, pixel_noise=0.02)[0, ..., 0] for i in range(B)])
    elif kind in ('one_busy_tile', 'masked_tile'):
        # one tile at full resolution beside tiles that are ONE 64 x 64 cell each (a row with ~64 neighbours in another tile per side),
        # and the same with one whole tile under the mask (an empty tile: its workgroups only count themselves out)
        rng = np.random.default_rng(70)
        img = np.zeros((B, 128, 128), np.float32)
        img[:, :64, :64] = rng.random((B, 64, 64)).astype(np.float32)
        img[:, 64:, 64:80] = 0.5 * rng.random((B, 64, 16)).astype(np.float32)
        if kind == 'masked_tile':
            mask = np.zeros((128, 128), dtype=bool)
            mask[:64, 64:] = True
        thresh = 0.3
    else:
        shape = (96, 128) if kind == 'ice96x128' else (256, 256)
        clips = [synthetic.make_ice_like(60 + i, shape=shape, channels=1, n_frames=1) for i in range(B)]
        img = np.stack([abs(abs(c[0][0, ..., 0] - 0.5) - 0.5) for c in clips])
        mask, thresh = clips[0][1], 0.15
    mesh = build_mesh(src=torch.from_numpy(np.ascontiguousarray(img)).to(dev()), thresh=thresh, mask=mask, static=static)
    return mesh, img


@pytest.mark.parametrize('kind,B', [('mnist128_sparse', 2), ('mnist128_noisy', 1), ('wide64x128', 3), ('ice96x128', 2), ('ice256', 1),
                                    ('one_busy_tile', 2), ('masked_tile', 2)])
def test_tile_records_of_multi_tile_meshes(kind, B):
    """qt_edges_norm_tiles: per tile (a contiguous label range, Mesh.cell_off) every row with an edge that leaves the tile has
    exactly one boundary record -- first four CSR edges as local rows or halo slots, the rest in the boundary pool, weights =
    nrm --, every such edge owns one halo slot that names the neighbour's global row, every other row with more than four edges
    one interior record; counts stay inside the capacities the kernel's LDS is laid out for."""
    mesh, _ = _tile_mesh(kind, B)
    tl = mesh.tiles
    assert tl is not None and mesh.tail_rec is None
    T, BT = tl['T'], mesh.B * tl['T']
    off = mesh.cell_off.cpu().numpy()
    rp, col, nrm = mesh.rowptr.cpu().numpy(), mesh.col.cpu().numpy(), mesh.nrm.cpu().numpy()
    cnt = tl['cnt'].cpu().numpy().reshape(BT, 32)
    brec = tl['brec'].cpu().numpy().view(np.uint32)
    bpool = tl['bpool'].cpu().numpy().view(np.uint32)
    rec = tl['rec'].cpu().numpy().view(np.uint32)
    pool = tl['pool'].cpu().numpy().view(np.uint32)
    halo = tl['halo'].cpu().numpy()
    assert not cnt[:, 5].any() and off[-1] == mesh.N
    seen_b = seen_i = 0
    for ts in range(BT):
        t0, nr = int(off[ts]), int(off[ts + 1] - off[ts])
        deg = rp[t0 + 1:t0 + nr + 1] - rp[t0:t0 + nr]
        remote = [[not (t0 <= col[e] < t0 + nr) for e in range(rp[t0 + r], rp[t0 + r + 1])] for r in range(nr)]
        brows = {r for r in range(nr) if any(remote[r])}
        irows = {r for r in range(nr) if deg[r] > 4 and r not in brows}
        assert cnt[ts, 3] == len(brows) <= 256 and cnt[ts, 2] == sum(sum(x) for x in remote) <= 256
        assert cnt[ts, 1] == len(irows) and nr + len(irows) <= 4096 and cnt[ts, 4] <= 1024
        got = set()
        for j in range(int(cnt[ts, 3])):
            ent = brec[ts, j]
            r = int(ent[7])
            assert r in brows and r not in got
            got.add(r)
            e0 = int(rp[t0 + r])
            fields = [int(ent[0]) & 0xffff, int(ent[0]) >> 16, int(ent[1]) & 0xffff, int(ent[1]) >> 16]
            wts = ent[2:6].view(np.float32)
            info = int(ent[6])
            d = int(deg[r])
            assert (info >> 16) == max(d - 4, 0)
            for k in range(d):
                if k < 4:
                    f, wv = fields[k], wts[k]
                    idx, is_halo = f >> 4, bool(f & 1)
                else:
                    pe = bpool[ts, (info & 0xffff) + k - 4]
                    idx, is_halo, wv = int(pe[0]) & 0x7fffffff, bool(int(pe[0]) >> 31), pe[1:2].view(np.float32)[0]
                cj = int(col[e0 + k])
                assert is_halo == remote[r][k]
                assert (int(halo[ts, idx]) if is_halo else t0 + idx) == cj
                assert wv == nrm[e0 + k]
        assert got == brows
        goti = set()
        for j in range(int(cnt[ts, 1])):
            ent = rec[ts, j]
            r = int(ent[7])
            assert r in irows and r not in goti
            goti.add(r)
            e0 = int(rp[t0 + r])
            assert [(int(ent[0]) >> 4) & 4095, int(ent[0]) >> 20, (int(ent[1]) >> 4) & 4095, int(ent[1]) >> 20] == [int(col[e0 + k]) - t0 for k in range(4)]
            info = int(ent[6])
            if (info & 0xffff) != 0xffff:
                for k in range(4, int(deg[r])):
                    pe = pool[ts, (info & 0xffff) + k - 4]
                    assert int(pe[0]) == int(col[e0 + k]) - t0 and pe[1:2].view(np.float32)[0] == nrm[e0 + k]
        assert goti == irows
        seen_b += len(brows)
        seen_i += len(irows)
    assert seen_b > 0


@pytest.mark.parametrize('kind,B', [('mnist128_sparse', 2), ('mnist128_noisy', 2), ('wide64x128', 3), ('ice96x128', 2), ('ice256', 1),
                                    ('ice256', 2), ('one_busy_tile', 2), ('masked_tile', 2)])      # (2 x 16 tiles x 10 slices = 320 workgroups: two launches of whole groups)
@pytest.mark.parametrize('K,widths', [(3, (4, 16)), (5, (16, 16)), (7, (8, 32)), (4, (16,))])
def test_tile_resident_recurrence_equals_per_hop_launches(kind, B, K, widths):
    """csrc/chebclip.hip with TILE = true -- frames of several 64 x 64 base cells: one workgroup per (clip, tile, slice), rows on
    tile borders exchanged between the workgroups of a clip after every hop through global memory -- against one qt_spmm2 launch
    per hop: forward planes T_1 .. T_{K-1} and the Clenshaw backward (row-major and slice-major gradient planes), bit for bit;
    the error word stays 0, the launch generations count the launches and no arrival is left pending."""
    from qtmpnn import ops
    from qtmpnn.mesh import spmm2
    mesh, _ = _tile_mesh(kind, B)
    assert mesh.tiles is not None          # (ops._tile_resident gates the path by shape; the launches are called directly here)
    torch.manual_seed(K)
    N = mesh.N
    wide = torch.randn(N, sum(widths) + 8, device=dev())
    Zs, o = [], 4
    for w in widths:
        Zs.append(wide[:, o:o + w])
        o += w
    fused = [torch.empty(K - 1, N, w, device=dev()) for w in widths]
    ops.clip_planes(mesh, Zs, fused, K)
    assert int(mesh.tiles['err']) == 0, 'error word set by the forward launch'
    prev, ops._CLIP_CHEB = ops._CLIP_CHEB, False          # one qt_spmm2 launch per hop
    try:
        ref, sm = ops._cheb_planes(Zs, mesh, K)
        assert sm == 0
    finally:
        ops._CLIP_CHEB = prev
    for a, r in zip(fused, ref):
        assert torch.equal(ops.planes_rowmajor(a, 1), r)
    G = [torch.randn(K, N, w, device=dev()) for w in widths]
    Gr = [g.clone() for g in G]
    for k in range(K - 2, 0, -1):
        spmm2(mesh, [g[k + 1] for g in Gr], 2.0, [g[k] for g in Gr], 1.0, [g[k + 2] for g in Gr] if k + 2 < K else None, -1.0,
              [g[k] for g in Gr])
    spmm2(mesh, [g[1] for g in Gr], 1.0, [g[0] for g in Gr], 1.0, [g[2] for g in Gr] if K > 2 else None, -1.0, [g[0] for g in Gr])
    Gf = [g.clone() for g in G]
    ops.clip_clenshaw(mesh, Gf, K)
    Gs = []
    for g0, w in zip(G, widths):
        t = g0.clone()
        t[1:] = g0[1:].view(K - 1, N, w // 4, 4).permute(0, 2, 1, 3).reshape(K - 1, N, w)
        Gs.append(t)
    ops.clip_clenshaw(mesh, Gs, K, sm=1)
    for a, b, r, g0 in zip(Gf, Gs, Gr, G):
        assert torch.equal(a[0], r[0]) and torch.equal(b[0], r[0])
        assert torch.equal(a[1:], g0[1:])                          # planes 1 .. K-1 stay as given
    sync = mesh.tiles['sync'].cpu().numpy()
    ns = len(sync) // 2
    assert int(mesh.tiles['err']) == 0 and not sync[ns:2 * ns].any() and sync[:ns].max() == 3      # three launches per used slice, none pending


def test_tile_resident_recurrence_static_capacities_and_graph_replay():
    """The tile-resident launches on a static-capacity mesh (node counts and tile ranges read on the device), captured into a
    hipGraph together with the mesh build and replayed on another image: valid rows equal the exact-size mesh's planes."""
    from qtmpnn import ops
    from qtmpnn.mesh import build_mesh
    mesh_a, img_a = _tile_mesh('mnist128_sparse', 2)
    mesh_b, img_b = _tile_mesh('ice96x128', 2)
    K, w = 5, 16
    img = torch.zeros(2, 128, 128, device=dev())
    Z = torch.randn(2 * 128 * 128, w, device=dev())
    out = torch.zeros(K - 1, 2 * 128 * 128, w, device=dev())
    Gp = torch.randn(K, 2 * 128 * 128, w, device=dev())
    Gw = torch.zeros_like(Gp)

    def run():
        sm = build_mesh(src=img, thresh=0.1, static=True)
        ops.clip_planes(sm, [Z], [out], K)
        Gw.copy_(Gp)
        ops.clip_clenshaw(sm, [Gw], K)
        return sm
    img.copy_(torch.from_numpy(img_a).to(dev()))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        sm = run()
    for im in (img_a, np.roll(img_a, 7, axis=2), img_a):
        img.copy_(torch.from_numpy(np.ascontiguousarray(im)).to(dev()))
        graph.replay()
        torch.cuda.synchronize()
        exact = build_mesh(src=img.clone(), thresh=0.1)
        nv = exact.N
        assert sm.n_valid == nv and int(sm.tiles['err']) == 0
        prev, ops._CLIP_CHEB = ops._CLIP_CHEB, False
        try:
            ref, _ = ops._cheb_planes([Z[:nv].contiguous()], exact, K)
        finally:
            ops._CLIP_CHEB = prev
        assert torch.equal(ops.planes_rowmajor(out, 1)[:, :nv], ref[0])
        Gr = Gp[:, :nv].clone()
        from qtmpnn.mesh import spmm2
        for k in range(K - 2, 0, -1):
            spmm2(exact, [Gr[k + 1]], 2.0, [Gr[k]], 1.0, [Gr[k + 2]] if k + 2 < K else None, -1.0, [Gr[k]])
        spmm2(exact, [Gr[1]], 1.0, [Gr[0]], 1.0, [Gr[2]], -1.0, [Gr[0]])
        assert torch.equal(Gw[0, :nv], Gr[0])



def _unpublish_one_boundary_row(mesh):
    """Point the published-values address of one boundary row at a halo slot of its tile that no record owns (the last one): the
    neighbour tiles that need the row then poll for a tag nobody writes.  Returns (global row, bad address)."""
    tl = mesh.tiles
    cap = tl['brec'].shape[1]
    cnt = tl['cnt'].cpu().numpy().reshape(-1, 32)
    off = mesh.cell_off.cpu().numpy()
    ts = next(t for t in range(cnt.shape[0]) if 0 < cnt[t, 3] < cap - 1)
    row = int(off[ts]) + int(tl['brec'][ts, 0, 7])
    bad = ts * cap + cap - 1
    assert int(tl['baddr'][row]) == ts * cap
    tl['baddr'][row:row + 1].fill_(bad)
    return row, bad


def test_missed_tile_handoff_sets_the_persistent_error_word_and_raises():
    """A tile that waits for a neighbour tile's boundary row that is never published gives up after a BOUNDED spin, the launch
    ends (garbage planes) and bit 0 goes into the device's persistent error word: the word outlives the mesh (it is not part of the
    buffers a mesh build zeroes), survives mesh builds and hipGraph replays, and check_tile_errors() turns it into a RuntimeError
    and clears it.  A negative test of a bounded path, run once."""
    import time
    from qtmpnn import mesh as M, ops
    assert M.tile_error_word(reset=True) == 0
    mesh, img = _tile_mesh('mnist128_sparse', 2)
    _unpublish_one_boundary_row(mesh)
    K, N = 5, mesh.N
    Z = torch.randn(N, 16, device=dev())
    out = torch.empty(K - 1, N, 16, device=dev())
    t0 = time.time()
    ops.clip_planes(mesh, [Z], [out], K)
    torch.cuda.synchronize()
    assert time.time() - t0 < 30, 'the spin is not bounded'
    assert int(mesh.tiles['err']) == 1
    del mesh, out
    other, _ = _tile_mesh('ice96x128', 2)                    # another mesh build does not clear the word
    assert M.tile_error_word() == 1 and int(other.tiles['err']) == 1
    with pytest.raises(RuntimeError, match='waited in vain'):
        M.check_tile_errors(always=True)
    assert M.tile_error_word() == 0                           # (the check reports once and clears)
    M.check_tile_errors(always=True)

    # the same inside a captured sequence: mesh build + corrupted address + launch, replayed twice
    src = torch.from_numpy(np.ascontiguousarray(img)).to(dev())
    probe = M.build_mesh(src=src, thresh=0.1, static=True)
    row, bad = _unpublish_one_boundary_row(probe)
    Zc = torch.randn(2 * 128 * 128, 16, device=dev())
    outc = torch.zeros(K - 1, 2 * 128 * 128, 16, device=dev())

    def run():
        sm = M.build_mesh(src=src, thresh=0.1, static=True)
        sm.tiles['baddr'][row:row + 1].fill_(bad)
        ops.clip_planes(sm, [Zc], [outc], K)
        return sm
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    assert M.tile_error_word(reset=True) == 1
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        keep = run()
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        assert M.tile_error_word() == 1
    del keep, graph
    with pytest.raises(RuntimeError, match='tile-resident'):
        M.check_tile_errors(always=True)


def test_training_step_and_predict_raise_on_a_missed_tile_handoff(monkeypatch):
    """The product path reads the error word: an eager train_step() that issued tile-resident launches checks it before returning,
    predict() at its end, train() once per epoch -- a missed hand-off is a RuntimeError, not a silently wrong gradient."""
    from model.mpnnlstm import NextFramePredictorS2S
    from qtmpnn import mesh as M, synthetic
    assert M.tile_error_word(reset=True) == 0
    orig = M._finish_mesh

    def finish_and_unpublish(ms, *a, **k):
        orig(ms, *a, **k)
        if ms.tiles is not None:
            _unpublish_one_boundary_row(ms)
    x, y = synthetic.make_batch(3, 0, 1, 2, 1, n_digits=2, pixel_noise=0.0, canvas=(128, 128))
    xt, yt = torch.from_numpy(x[0]).to(dev()), torch.from_numpy(y[0]).to(dev())
    mask = np.zeros((128, 128), dtype=bool)
    torch.manual_seed(0)
    nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=2, output_timesteps=1, device=dev(),
                                model_kwargs=dict(hidden_size=8, dropout=0.0, n_layers=1, n_conv_layers=2))
    nfp.initiate_training(1e-3, 0.95)
    good = float(nfp.train_step(xt, yt, None, mask))           # the healthy path does not raise
    assert np.isfinite(good)
    monkeypatch.setattr(M, '_finish_mesh', finish_and_unpublish)
    with pytest.raises(RuntimeError, match='waited in vain'):
        nfp.train_step(xt, yt, None, mask)
    assert M.tile_error_word() == 0
    loader = TinyLoaderLocal([(xt[None].cpu(), yt[None].cpu(), torch.tensor([0]))], (128, 128))
    with pytest.raises(RuntimeError, match='waited in vain'):
        nfp.predict(loader, mask=mask)
    with pytest.raises(RuntimeError, match='waited in vain'):
        nfp.train(loader, loader, n_epochs=1, lr=1e-3, mask=mask, truncated_backprop=0)
    monkeypatch.setattr(M, '_finish_mesh', orig)
    M.tile_error_word(reset=True)
    assert np.isfinite(float(nfp.train_step(xt, yt, None, mask)))


class TinyLoaderLocal(list):
    def __init__(self, items, image_shape):
        super().__init__(items)
        self.dataset = type('DS', (), {'image_shape': tuple(image_shape)})()


def test_tile_capacity_overflow_is_reported_not_waited_for():
    """The capacity-overflow path of qt_edges_norm_tiles / qt_cheb_tile_* cannot be reached with the shipped capacities on a quadtree
    mesh, so the small-caps build of the library (QT_TILE_HALO_CAP = 16) runs it in a child process: rows whose boundary record
    does not fit get the sentinel address -1, nobody waits for them (the launch returns at once), the error word is exactly
    bit 1 and check_tile_errors() names the cause (tests/_tile_overflow_child.py)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(os.path.dirname(here), 'quadtree-mpnnlstm_amd', 'qtmpnn', 'libqtmpnn_hip_smallcaps.so')
    assert os.path.exists(lib), 'run __graft_entry__.build() (make -C quadtree-mpnnlstm_amd/csrc)'
    env = dict(os.environ, QT_LIB_PATH=lib)
    r = subprocess.run([sys.executable, os.path.join(here, '_tile_overflow_child.py')], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'tile overflow ok' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_more_tiles_than_compute_units_is_an_argument_error():
    """qt_cheb_tile_fwd / _bwd need every tile of a launch resident at once: B x T > CUs is refused by the entry point itself (the
    Python gate ops._tile_resident never gets there)."""
    from qtmpnn import _lib, ops
    ncu = _lib.value('qt_num_cus')
    B = ncu // 4 + 1
    img = np.zeros((B, 128, 128), np.float32)
    img[:, 40:90, 30:100] = 1.0
    from qtmpnn.mesh import build_mesh
    mesh = build_mesh(src=torch.from_numpy(img).to(dev()), thresh=0.1)
    assert mesh.tiles is not None and mesh.B * mesh.tiles['T'] > ncu
    assert not ops._tile_resident(mesh, [16], 5)
    Z = torch.randn(mesh.N, 16, device=dev())
    with pytest.raises(RuntimeError, match='more tiles'):
        ops.clip_planes(mesh, [Z], [torch.empty(4, mesh.N, 16, device=dev())], 5)
    with pytest.raises(RuntimeError, match='more tiles'):
        ops.clip_clenshaw(mesh, [torch.randn(5, mesh.N, 16, device=dev())], 5)


def test_masks_are_cached_by_content():
    """A host mask is uploaded once per CONTENT: an array changed in place is a new mask (round 4 cached by object identity and
    went stale), an equal array built anew is the same one; new content inside a hipGraph capture is a clear error."""
    from qtmpnn.mesh import build_mesh, build_pixel_mesh
    img = torch.rand(1, 64, 64, device=dev())
    mask = np.zeros((64, 64), dtype=bool)
    mask[:8, :8] = True
    a = build_mesh(src=img, thresh=0.5, mask=mask)
    assert (a.labels[0, :8, :8] == -1).all() and int(a.labels[0, 20, 20]) >= 0
    mask[16:24, 16:24] = True                                  # in place
    b = build_mesh(src=img, thresh=0.5, mask=mask)
    assert (b.labels[0, 16:24, 16:24] == -1).all() and b.N == a.N - 64
    c = build_mesh(src=img, thresh=0.5, mask=mask.copy())
    assert c.mask.data_ptr() == b.mask.data_ptr()              # same content: the cached upload
    p1 = build_pixel_mesh(1, 64, 64, mask, dev())
    mask[40, 40] = True
    p2 = build_pixel_mesh(1, 64, 64, mask, dev())
    assert p2.N == p1.N - 1 and int(p2.labels[0, 40, 40]) == -1
    fresh = np.zeros((64, 64), dtype=bool)
    fresh[5, 50] = True
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match='warm-up'):
        with torch.cuda.graph(graph, stream=side):
            build_mesh(src=img, thresh=0.5, mask=fresh, static=True)


def test_tile_path_is_taken_where_it_was_measured_faster():
    """ops._tile_resident: frames of several base cells take the tile-resident launch for K >= 4 when all workgroups of the launch
    fit the CUs in one round (BASELINE configs[2]: 8 clips x 4 tiles x 5 .. 8 slices), and stay on one k_spmm launch per hop
    otherwise (two hops; 16 clips x 4 tiles x 10 slices = 640 workgroups); 64 x 64 frames never take it."""
    from qtmpnn import ops
    m8, _ = _tile_mesh('mnist128_noisy', 8)
    assert ops._tile_resident(m8, [4, 16], 5) and ops._tile_resident(m8, [16, 16], 5) and ops._clip_resident(m8, [16], 5)
    assert not ops._tile_resident(m8, [4, 16], 3) and not ops._clip_resident(m8, [4, 16], 3)
    m16, _ = _tile_mesh('ice96x128', 16)
    assert not ops._tile_resident(m16, [8, 32], 7) and ops._tile_resident(m16, [8], 7)
    small, _ = _mesh_64(3, noise=0.02, B=2)
    assert small.tiles is None and not ops._tile_resident(small, [4, 16], 5) and ops._clip_resident(small, [4, 16], 5)


def test_clip_resident_recurrence_static_capacities():
    """The same launch on a static-capacity mesh (N = B n m rows, valid counts per clip in node_off on the device): valid rows
    equal the exact-size mesh's, capacity rows are never read (NaN poison) or written."""
    from qtmpnn import ops
    from qtmpnn.mesh import build_mesh
    mesh, img = _mesh_64(12, noise=0.02, B=2)
    sm = build_mesh(src=torch.from_numpy(img).to(dev()), thresh=0.1, static=True)
    nv = sm.n_valid
    assert nv == mesh.N and sm.N == 2 * 64 * 64
    torch.manual_seed(1)
    Z = torch.randn(mesh.N, 16, device=dev())
    Zs = torch.full((sm.N, 16), float('nan'), device=dev())
    Zs[:nv] = Z
    planes, layout = ops._cheb_planes([Z], mesh, 4)
    ref = ops.planes_rowmajor(planes[0], layout)
    got = torch.full((3, sm.N, 16), 7.0, device=dev())
    from qtmpnn import _lib
    from qtmpnn._lib import ptr
    _lib.call('qt_cheb_clip_fwd', ptr(sm.rowptr), ptr(sm.col), ptr(sm.nrm), ptr(sm.ell), ptr(sm.node_off), ptr(sm.tail_cnt),
              ptr(sm.tail_pool), ptr(sm.tail_rec), sm.B, sm.N, 4, 16, ptr(Zs), 16, ptr(got), 0, None, 0, None, 0)
    got = ops.planes_rowmajor(got, 1)
    assert torch.equal(got[:, :nv], ref)
    assert bool((got[:, nv:] == 7.0).all())


def test_concat_cols_kernel():
    """qt_concat: column concatenation of row-strided sources == torch.cat, and the backward hands out column views."""
    from qtmpnn import ops
    torch.manual_seed(0)
    wide = torch.randn(1000, 24, device=dev())
    a, b, c = wide[:, 4:8].requires_grad_(True), torch.randn(1000, 16, device=dev(), requires_grad=True), wide[:, 12:20].requires_grad_(True)
    out = ops.concat_cols([a, b, c])
    assert torch.equal(out, torch.cat([a, b, c], dim=1))
    g = torch.randn_like(out)
    ga, gb, gc = torch.autograd.grad(out, [a, b, c], g)
    assert torch.equal(ga, g[:, :4]) and torch.equal(gb, g[:, 4:20]) and torch.equal(gc, g[:, 20:])


@pytest.mark.parametrize('cin,in_pad,K,L', [(4, 4, 3, 2), (5, 8, 3, 2), (4, 4, 2, 2), (8, 8, 3, 3), (5, 8, 2, 4)])
def test_compose2_kernel_equals_torch_composition(cin, in_pad, K, L):
    """qt_compose_step / qt_compose2 (L-layer stacks composed in weight space, the last product laid out as the packed gate
    matrix: one launch per product) == the torch composition (ops.compose_chebconvs + GConvLSTM._assemble) for both
    variants: values and every parameter gradient,
    fp32 tolerance rtol 1e-4 / atol 1e-5 (sums of <= 9 products of 16-term dot products, different summation order)."""
    from model.model import CONVOLUTION_KWARGS, GConvLSTM
    from qtmpnn import ops
    torch.manual_seed(4)
    old = dict(CONVOLUTION_KWARGS['ChebConv'])
    CONVOLUTION_KWARGS['ChebConv']['K'] = K
    try:
        cell = GConvLSTM(cin, 16, L, 'ChebConv').to(dev())
    finally:
        CONVOLUTION_KWARGS['ChebConv'].update(old)
    for p in cell.parameters():
        p.data.normal_(std=0.5)
    variants = (False, True)
    ref = cell.pack(in_pad, None, variants)                       # torch ops only
    params = cell.plan_params()
    plan = ops.PackPlan(params, lambda T, fill: cell.plan_layout(T, fill, 'c.', in_pad, variants))
    got = cell.pack_from(plan(), 'c.', in_pad, None, variants)    # gather + compose kernel
    for r, g in zip(ref, got):
        assert (r.K, r.Ks) == (g.K, g.Ks) and r.W.shape == g.W.shape
        close(g.W, r.W, 1e-4, 1e-5)

    def probe(cells):
        gen = torch.Generator().manual_seed(0)
        return sum((c.W * torch.randn(c.W.shape, generator=gen).to(c.W.device)).sum() for c in cells)
    gr = torch.autograd.grad(probe(ref), params, allow_unused=True, retain_graph=True)
    gg = torch.autograd.grad(probe(got), params, allow_unused=True)
    for a, b in zip(gr, gg):
        close(b, a if a is not None else torch.zeros_like(b), 1e-4, 1e-5)      # (the plan hands zeros to unused parameters)
    # one variant only (the upper encoder layers): the h-branch weights still get their bias-row gradient
    one = cell.pack_from(plan(), 'c.', in_pad, None, (False,))
    close(one[0].W, ref[0].W, 1e-4, 1e-5)
    g1 = torch.autograd.grad(probe(one), params, allow_unused=True)
    r1 = torch.autograd.grad(probe(ref[:1]), params, allow_unused=True)
    for a, b in zip(r1, g1):
        if a is not None:
            close(b if b is not None else torch.zeros_like(a), a, 1e-4, 1e-5)


def test_attention_dropout_epoch_counter():
    """Attention dropout masks come from (host seed, device step counter, edge): the same seed and counter reproduce the
    forward exactly (the backward relies on that), advancing the counter on the device -- all a hipGraph replay can do --
    draws a new mask for the forward and the gradient alike; without dropout the counter plays no part."""
    from qtmpnn import ops
    mesh, _ = _mesh_64(82, noise=0.0, B=2)
    torch.manual_seed(9)
    C = 8
    proj = torch.randn(mesh.N, 4 * C, device=dev())
    We = torch.randn(C, 2, device=dev())

    def run(seed):
        p = proj.clone().requires_grad_(True)
        out = ops._Attention.apply(p, We, mesh, C, 0.5, seed, None)
        g, = torch.autograd.grad(out, p, torch.ones_like(out))
        return out.detach(), g
    ep = ops.dropout_epoch(dev())
    start = int(ep.item())
    a, ga = run(1234)
    b, gb = run(1234)
    assert torch.equal(a, b) and torch.equal(ga, gb)
    ops.advance_dropout_epoch(dev())
    assert int(ep.item()) == start + 1
    c, gc = run(1234)
    assert not torch.equal(a, c) and not torch.equal(ga, gc)
    # keep = 1: the counter plays no part
    d0 = ops._Attention.apply(proj, We, mesh, C, 1.0, 1, None)
    ops.advance_dropout_epoch(dev())
    assert torch.equal(d0, ops._Attention.apply(proj, We, mesh, C, 1.0, 1, None))
    # about half of the coefficients survive, scaled by 1 / keep: the mean over many targets stays near the undropped output
    rel = float((c - d0).abs().mean() / d0.abs().mean())
    assert 0.05 < rel < 2.0, rel


@pytest.mark.parametrize('name', ['mnist64_h16', 'mnist64_noise_h8'])
def test_fused_cell_backward_with_weight_gradient_matches_reference(name, monkeypatch):
    """qt_lstm_bwd_fused (opt-in: QT_WGRAD_FUSION=1 -- cell backward, data gradient and weight gradient of a gate-cell use in
    one persistent launch, gG never written): the rollout's loss and all parameter gradients against the reference trace,
    hidden 16 (64-row tiles, 2 + 3 column tiles) and hidden 8."""
    from helpers import golden, grad_close, load_state
    from model.mpnnlstm import masked_mse
    from model.seq2seq import Seq2Seq
    monkeypatch.setenv('QT_WGRAD_FUSION', '1')
    g = golden(f'rollout_{name}.npz')
    x, y, concat = (torch.from_numpy(g[k]).to(dev()) for k in ('x', 'y', 'concat'))
    model = Seq2Seq(hidden_size=int(g['hidden']), dropout=0.0, thresh=float(g['thresh']), input_timesteps=x.shape[0],
                    input_features=x.shape[-1] + 3, output_timesteps=y.shape[0], n_layers=int(g['n_layers']),
                    n_conv_layers=int(g['n_conv']), convolution_type='ChebConv')
    load_state(model, g, 'w/')
    model.to(dev())
    outs, meshes = model(x, y, concat, teacher_forcing_ratio=0, mask=g['mask'])
    loss = masked_mse(outs, meshes, y, g['mask'])
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    loss.backward()
    for k, p in model.named_parameters():
        ref = g['g/' + k]
        if p.grad is None:
            assert not ref.any(), k
            continue
        grad_close(p.grad, ref, msg=k)


@pytest.mark.parametrize('capturable', [False, True])
def test_flat_adam_equals_clip_grad_norm_plus_torch_adam(capturable):
    """qt_flat_adam (clip + Adam on the flat parameter vector, two launches) against torch.nn.utils.clip_grad_norm_ +
    torch.optim.Adam over a few steps, with gradients above and below the clipping norm and a learning-rate change."""
    from qtmpnn.optim import FlatAdam
    torch.manual_seed(0)
    n = 34513
    p0 = torch.randn(n, device=dev())
    pa, pb = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p0.clone())
    oa = FlatAdam(pa, lr=0.01, capturable=capturable)
    ob = torch.optim.Adam([pb], lr=0.01)
    sched = torch.optim.lr_scheduler.StepLR(oa, step_size=2, gamma=0.5)
    for it in range(5):
        g = torch.randn(n, device=dev()) * (0.2 if it % 2 else 0.01)         # norm ~37 (clipped to 10) / ~1.9 (not clipped)
        pa.grad, pb.grad = g.clone(), g.clone()
        norm = torch.nn.utils.clip_grad_norm_([pb], 10.0)
        ob.step()
        oa.step(max_norm=10.0)
        assert abs(float(oa.last_norm[0]) - float(norm)) <= 1e-5 * float(norm)
        close(pa.grad, pb.grad, rtol=1e-5, atol=1e-8, msg=f'clipped gradient, step {it}')
        close(pa, pb, rtol=1e-5, atol=1e-6, msg=f'weights, step {it}')
        sched.step()
        for gr in ob.param_groups:
            gr['lr'] = float(sched.get_last_lr()[0])
    assert int(oa.state[pa]['step']) == 5


def test_scalar_cheb3_residual_as_column_view():
    """ops.scalar_cheb3 (the decoder head's one-output-channel ChebConv on single columns, model/seq2seq.py:121,174-186) with
    the residual operand given as a column view of a wider matrix (row stride 4, one column): same Y, dU and d res as with a
    contiguous (N, 1) residual -- the backward used to size the residual's gradient buffer by the view's shape while the kernel
    writes rows of the view's stride."""
    from qtmpnn import ops
    mesh, _ = _mesh_64(40, noise=0.0, B=2)
    N = mesh.N
    torch.manual_seed(3)
    U0 = torch.randn(N, 4, device=dev())
    U0[:, 3] = 0
    X0 = torch.randn(N, 4, device=dev())
    drop = (torch.rand(N, device=dev()) > 0.2).float() / 0.8
    gy = torch.randn(N, 4, device=dev())
    gy[:, 1:] = 0
    outs = []
    for view in (True, False):
        U = U0.clone().requires_grad_(True)
        X = (X0.clone() if view else X0[:, :1].clone()).requires_grad_(True)
        Y = ops.scalar_cheb3(U, X[:, :1], drop, mesh)
        gU, gX = torch.autograd.grad(Y, [U, X], gy)
        outs.append((Y.detach(), gU, gX[:, :1], gX))
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a, b)
    assert bool((outs[0][3][:, 1:] == 0).all())


@pytest.mark.parametrize('h', [16, 8])
def test_gate_cell_second_consumers_sum_inside_the_backward_launch(h):
    """ops.gate_cell(alias_h=True, pass_x=True): H' and X handed out a second time for their second consumers (the decoder's
    layer-0 state feeds layer 1 AND the next step; the decoder input is also the head's residual, model/seq2seq.py:152-186).
    Same values as using H' / X twice, and the same gradients: the fused backward launch adds the second gradient of H' on
    load and the pass-through gradient of X into plane 0 (tolerance: the sums are associated differently)."""
    from qtmpnn import ops
    mesh, _ = _mesh_64(7, noise=0.02, B=2)
    N, K, Ks = mesh.N, 3, 1
    torch.manual_seed(5)
    X0, H0, C0 = torch.randn(N, 4, device=dev()), torch.randn(N, h, device=dev()), torch.randn(N, h, device=dev())
    W0 = 0.2 * torch.randn(K * (4 + h) + 4, 4 * h, device=dev())
    wc0, b0, ln0 = 0.1 * torch.randn(3, h, device=dev()), 0.1 * torch.randn(4, h, device=dev()), torch.randn(4, h, device=dev())
    w1, w2, w3, w4 = (torch.randn(N, c, device=dev()) for c in (h, h, 4, h))
    res = []
    for alias in (True, False):
        X, H, C, W = (t.clone().requires_grad_(True) for t in (X0, H0, C0, W0))
        out = ops.gate_cell(X, H, W, C, wc0, b0, ln0, mesh, K, Ks, alias_h=alias, pass_x=alias)
        O, Hn, Cn = out[:3]
        Hn2, Xp = (out[3], out[4]) if alias else (Hn, X)
        assert Hn2.data_ptr() == Hn.data_ptr() and Xp.data_ptr() == X.data_ptr()
        loss = (Hn * w1).sum() + (Hn2 * w2).sum() + (Xp * w3).sum() + (Cn * w4).sum() + O.sum()
        res.append(torch.autograd.grad(loss, [X, H, C, W]))
    for a, b in zip(*res):
        close(a, b, rtol=1e-5, atol=1e-5 * float(b.abs().max()))


def test_head_products_in_one_launch_each_way():
    """ops.cheb_poly(post=...): the decoder head's fc_out1 (K = 3, 20 -> 16, ReLU) and the coefficient columns of fc_out2
    (16 -> 4, model/seq2seq.py:115-121,160-186) as ONE launch forward (second product in the epilogue) and the second product's
    data gradient + the ReLU gradient as ONE launch backward -- against the two cheb_poly calls they replace: Y, U and every
    gradient bit for bit (the same fused multiply-adds in the same order)."""
    from qtmpnn import ops
    mesh, _ = _mesh_64(9, noise=0.02, B=2)
    N = mesh.N
    torch.manual_seed(2)
    Za0, Zb0 = torch.randn(N, 16, device=dev()), torch.randn(N, 4, device=dev())
    W10 = 0.2 * torch.randn(3 * 20 + 1, 16, device=dev())
    W20 = 0.3 * torch.randn(16 + 1, 4, device=dev())
    gU = torch.randn(N, 4, device=dev())
    res = []
    # (forward fused, backward products in one launch): the shipped path; the round-3 path (gU -> G by the VALU kernel, G -> planes
    # on the MFMA); two separate cheb_poly calls
    for fuse, dgrad in ((True, True), (True, False), (False, False)):
        prev, ops._HEAD_FUSE, ops._HEAD_DGRAD = (ops._HEAD_FUSE, ops._HEAD_DGRAD), fuse, dgrad
        try:
            Za, Zb, W1, W2 = (t.clone().requires_grad_(True) for t in (Za0, Zb0, W10, W20))
            acc1, acc2 = ops.GradAcc(), ops.GradAcc()
            Y, U = ops.cheb_poly((Za, Zb), W1, mesh, 3, 1, ops.ACT_RELU, acc=acc1, post=(W2, acc2))
            grads = torch.autograd.grad(U, [Za, Zb, W1, W2], gU)
        finally:
            ops._HEAD_FUSE, ops._HEAD_DGRAD = prev
        res.append((Y.detach(), U.detach(), *grads))
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)


def test_row_per_lane_head_gemm_equals_the_column_split_kernel(tmp_path):
    """k_gemm_row16 (one lane = one node row, all 16 output columns: every operand quad loaded once) against k_gemm_skinny<64>
    (a wave per 4 columns), which it replaced for 16-column products: same outputs bit for bit, forward (ReLU + the second
    product in the epilogue) and as the ReLU-backward product.  The old kernel is selected by QT_GEMM_NO_ROW16=1, which the
    library reads once: a child process computes the reference."""
    import os
    import subprocess
    import sys
    code = """
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(sys.argv[1], 'quadtree-mpnnlstm_amd'))
sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
from qtmpnn import ops, synthetic
from qtmpnn.mesh import build_mesh
dev = torch.device('cuda', 0)
img = np.stack([synthetic.make_clip(9 + i, n_frames=1, pixel_noise=0.02)[0, ..., 0] for i in range(2)])
mesh = build_mesh(src=torch.from_numpy(img).to(dev), thresh=0.1)
torch.manual_seed(2)
N = mesh.N
Za, Zb = torch.randn(N, 16, device=dev), torch.randn(N, 4, device=dev)
W1, W2 = 0.2 * torch.randn(3 * 20 + 1, 16, device=dev), 0.3 * torch.randn(16 + 1, 4, device=dev)
gU = torch.randn(N, 4, device=dev)
prev, ops._HEAD_DGRAD = ops._HEAD_DGRAD, False
Za.requires_grad_(True); Zb.requires_grad_(True)
Y, U = ops.cheb_poly((Za, Zb), W1, mesh, 3, 1, ops.ACT_RELU, acc=ops.GradAcc(), post=(W2, ops.GradAcc()))
gza, gzb = torch.autograd.grad(U, [Za, Zb], gU)
np.savez(sys.argv[2], Y=Y.detach().cpu().numpy(), U=U.detach().cpu().numpy(), gza=gza.cpu().numpy(), gzb=gzb.cpu().numpy())
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for tag, extra in (('new', {}), ('old', {'QT_GEMM_NO_ROW16': '1'})):
        f = str(tmp_path / f'{tag}.npz')
        r = subprocess.run([sys.executable, '-c', code, root, f], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(np.load(f))
    for k in ('Y', 'U', 'gza', 'gzb'):
        assert np.array_equal(outs[0][k], outs[1][k]), k
