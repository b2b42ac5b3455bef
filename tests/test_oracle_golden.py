"""The CPU oracle against golden vectors captured by executing the reference
(tests/golden/make_golden.py).  Integer results are compared bit-exactly, fp32
results at rtol 1e-4 (north_star tolerance)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import qt_oracle as O

RTOL, ATOL = 1e-4, 1e-5


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def dist_from_05(arr):
    return abs(abs(arr - 0.5) - 0.5)


def test_kat_quadtree(golden_dir):
    k = load(golden_dir, 'kat.npz')
    for i in (1, 2, 3):
        lab = O.quadtree_decompose(k[f'kat{i}_img'], thresh=.5, max_size=4)
        assert np.array_equal(lab, k[f'kat{i}_labels'])
    # SURVEY.md KAT-1 literal (first rows) as an independent anchor
    assert k['kat1_labels'][0].tolist() == [12, 10, 7, 5, 2, 2, 2, 2]
    for cond in O.CONDITIONS:
        lab = O.quadtree_decompose(k['kat6_img'], thresh=.9 if 'max' in cond else .1, max_size=8, condition=cond)
        assert np.array_equal(lab, k['kat6_' + cond]), cond


def test_kat_adjacency_and_mapping(golden_dir):
    k = load(golden_dir, 'kat.npz')
    e = O.adjacency_sorted(k['kat4_labels'])
    assert np.array_equal(e, k['kat4_edges'])
    xx, yy = torch.tensor([.5, 2, 0, 1.5]), torch.tensor([0, 0, 1.5, 1.5])
    et = torch.as_tensor(e)
    np.testing.assert_allclose(O.edge_angle(et[0], et[1], xx, yy), k['kat4_attrs'][:, 0], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(O.edge_dist(et[0], et[1], xx, yy), k['kat4_attrs'][:, 1], rtol=RTOL, atol=ATOL)
    assert np.array_equal(O.dense_mapping(k['kat4_labels']), k['kat5_mapping'])
    assert np.array_equal(O.pixel_counts(k['kat4_labels']), k['kat5_npix'])


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'graph_*.npz'))),
                         ids=lambda p: os.path.basename(p)[6:-4])
def test_graph_build(path):
    g = np.load(path, allow_pickle=False)
    x = O.add_positional_encoding(torch.from_numpy(g['x']))
    mask = g['mask'] if 'mask' in g else None
    hir = g['hir'] if 'hir' in g else None
    out = O.image_to_graph(x, thresh=float(g['thresh']), mask=mask, high_interest_region=hir,
                           transform_func=dist_from_05 if bool(g['has_transform']) else None,
                           condition=str(g['condition']), use_edge_attrs=bool(g['use_attrs']))
    assert np.array_equal(out['labels'], g['labels'])
    assert np.array_equal(out['n_pixels_per_node'].numpy(), g['npix'])
    assert np.array_equal(out['edge_index'].numpy(), g['edges'])
    np.testing.assert_allclose(out['data'].numpy(), g['data'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out['edge_attrs'].numpy(), g['attrs'], rtol=RTOL, atol=ATOL)


def test_flatten_unflatten(golden_dir):
    t = load(golden_dir, 'transfer.npz')
    img = torch.from_numpy(t['img']).requires_grad_(True)
    flat = O.flatten(img, t['labels'], t['npix'])
    np.testing.assert_allclose(flat.detach().numpy(), t['flat'], rtol=RTOL, atol=ATOL)
    (gx,) = torch.autograd.grad(flat, img, torch.from_numpy(t['flat_gy']))
    np.testing.assert_allclose(gx.numpy(), t['flat_gx'], rtol=RTOL, atol=ATOL)
    data = torch.from_numpy(t['data']).requires_grad_(True)
    im = O.unflatten(data, t['labels'], (64, 64))
    np.testing.assert_allclose(im.detach().numpy(), t['unflat'], rtol=RTOL, atol=ATOL)
    (gd,) = torch.autograd.grad(im, data, torch.from_numpy(t['unflat_gi']))
    np.testing.assert_allclose(gd.numpy(), t['unflat_gd'], rtol=RTOL, atol=1e-4)


def _load_state(module, g, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}
    module.load_state_dict(sd, strict=True)


@pytest.mark.parametrize('n_conv', [1, 2, 3])
def test_gconvlstm_cell(golden_dir, n_conv):
    g = load(golden_dir, 'cells.npz')
    tag = f'nc{n_conv}'
    cell = O.GConvLSTM(4, 8, n_conv, 'ChebConv')
    _load_state(cell, g, f'{tag}_w/')
    ei, ew = torch.from_numpy(g['edges']).long(), torch.from_numpy(g['dist'])
    X, H, C = (torch.from_numpy(g[f'{tag}_{n}']).requires_grad_(True) for n in 'XHC')
    Oo, Hn, Cn = cell(X, ei, ew, H, C)
    for got, name in ((Oo, 'O'), (Hn, 'Hn'), (Cn, 'Cn')):
        np.testing.assert_allclose(got.detach().numpy(), g[f'{tag}_{name}'], rtol=RTOL, atol=ATOL)
    names = [k for k, _ in cell.named_parameters()]
    grads = torch.autograd.grad([Oo, Hn, Cn], [X, H, C] + list(cell.parameters()),
                                [torch.from_numpy(g[f'{tag}_g{n}']) for n in 'OHC'])
    for got, name in zip(grads[:3], ('gX', 'gHin', 'gCin')):
        np.testing.assert_allclose(got.numpy(), g[f'{tag}_{name}'], rtol=1e-3, atol=1e-4)
    for got, k in zip(grads[3:], names):
        ref = g[f'{tag}_g/{k}']
        np.testing.assert_allclose(got.numpy(), ref, rtol=1e-3, atol=1e-4 * max(1.0, np.abs(ref).max()))


def test_encoder_decoder_step(golden_dir):
    g = load(golden_dir, 'cells.npz')
    ei, ew = torch.from_numpy(g['edges']).long(), torch.from_numpy(g['dist'])
    enc = O.Encoder(4, 8, 2, 'ChebConv', 2)
    _load_state(enc, g, 'enc_w/')
    hid, cel = enc(torch.from_numpy(g['enc_X'])[0], ei, ew, torch.from_numpy(g['enc_H']), torch.from_numpy(g['enc_C']))
    np.testing.assert_allclose(hid.detach().numpy(), g['enc_hidden'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(cel.detach().numpy(), g['enc_cell'], rtol=RTOL, atol=ATOL)
    dec = O.Decoder(4, 8, 0.0, 2, 1, 'ChebConv')
    _load_state(dec, g, 'dec_w/')
    out, hid, cel = dec(torch.from_numpy(g['dec_X']), ei, ew, torch.from_numpy(g['dec_concat']),
                        torch.from_numpy(g['dec_H']), torch.from_numpy(g['dec_C']))
    np.testing.assert_allclose(out.detach().numpy(), g['dec_out'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(hid.detach().numpy(), g['dec_hidden'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(cel.detach().numpy(), g['dec_cell'], rtol=RTOL, atol=ATOL)


def oracle_rollout(g):
    x, y, concat = (torch.from_numpy(g[k]) for k in ('x', 'y', 'concat'))
    model = O.Seq2Seq(int(g['hidden']), 0.0, float(g['thresh']), input_timesteps=x.shape[0], input_features=x.shape[-1] + 3,
                      output_timesteps=y.shape[0], n_layers=int(g['n_layers']), n_conv_layers=int(g['n_conv']),
                      transform_func=dist_from_05 if bool(g['has_transform']) else None)
    _load_state(model, g, 'w/')
    hir = g['hir'] if 'hir' in g.files else None
    outs, maps, trace = model(x, concat, mask=g['mask'], high_interest_region=hir)
    loss = O.clip_loss(outs, maps, y, x.shape[1:3], g['mask'])
    return model, outs, maps, trace, loss


@pytest.mark.parametrize('name', ['mnist64_h16', 'mnist64_noise_h8', 'ice64_masked_h8', 'mnist64_l4_h8', 'cfg2_mnist64',
                                  'ice96x128_masked_h8', 'ice128_h32'])
def test_rollout(golden_dir, name):
    g = load(golden_dir, f'rollout_{name}.npz')
    model, outs, maps, trace, loss = oracle_rollout(g)
    for i, lab in enumerate(trace['labels']):
        assert np.array_equal(lab, g[f'labels_{i}']), f'mesh {i}'
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().numpy(), g[f'out_{i}'], rtol=RTOL, atol=ATOL)
    assert abs(float(loss.detach()) - float(g['loss'])) <= RTOL * abs(float(g['loss']))
    loss.backward()
    for k, p in model.named_parameters():
        ref = g['g/' + k]
        np.testing.assert_allclose(p.grad.numpy(), ref, rtol=1e-3, atol=1e-4 * max(1e-3, np.abs(ref).max()), err_msg=k)


def test_cheb_conv_self_consistency(golden_dir):
    """PyG arithmetic is parity-unpinned (oracle header): cross-check the restated ChebConv
    against an independent dense float64 formulation."""
    g = load(golden_dir, 'cells.npz')
    ei, ew = torch.from_numpy(g['edges']).long(), torch.from_numpy(g['dist'])
    n = int(ei.max()) + 1
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(n, 5, generator=gen)
    ws = [torch.randn(7, 5, generator=gen) for _ in range(3)]
    b = torch.randn(7, generator=gen)
    a = O.cheb_conv(x, ei, ew, ws, b)
    d = O.cheb_conv_dense(x, ei, ew, ws, b)
    np.testing.assert_allclose(a.numpy(), d.numpy(), rtol=1e-4, atol=1e-4)


def test_rollout_transformerconv(golden_dir):
    """Oracle control flow with attention convolutions (edge attributes [angle, dist], self pairs) vs the reference trace.
    The conv arithmetic itself is the shared restatement (parity unpinned, see the oracle header)."""
    g = load(golden_dir, 'transformer_rollout.npz')
    x, y, concat = (torch.from_numpy(g[k]) for k in ('x', 'y', 'concat'))
    model = O.Seq2Seq(8, 0.0, 0.15, input_timesteps=2, input_features=6, output_timesteps=3, n_layers=1, n_conv_layers=2,
                      transform_func=dist_from_05, convolution_type='TransformerConv')
    _load_state(model, g, 'w/')
    model.eval()
    outs, maps, _ = model(x, concat, mask=g['mask'])
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().numpy(), g[f'out_{i}'], rtol=RTOL, atol=ATOL)
    loss = O.clip_loss(outs, maps, y, (64, 64), g['mask'])
    assert abs(float(loss.detach()) - float(g['loss'])) <= RTOL * abs(float(g['loss']))


def _climatology(base):
    d = np.arange(365, dtype=np.float32)[:, None, None]
    return (base[None] * (0.5 + 0.5 * np.cos(2 * np.pi * d / 365.0)) + 0.001 * d)[None].astype(np.float32)


def _doys(launch, t_out):
    import datetime
    return [datetime.datetime.fromtimestamp((int(launch) + 8.640e13 * t) / 1e9).timetuple().tm_yday - 1 for t in range(t_out)]


@pytest.mark.parametrize('name', ['ice_exp9', 'ice_exp10', 'ice_exp1'])
def test_ice_exp_preset_experiments(golden_dir, name):
    """ice_exp.py exp 9 / 10 / 1 (preset heterogeneous / homogeneous mesh x edge attributes x resolution x TransformerConv; GCNConv on
    the pixelwise mesh): the oracle's preset-mesh and pixelwise paths against the reference's trainer trace -- meshes bit-exact,
    forward + backward at the initial weights, and the two train() phases (clip_grad_norm_(10) + Adam, per-epoch losses with the
    reference's /(steps + 1), weights after each phase)."""
    g = load(golden_dir, f'variant_{name}.npz')
    preset, conv = str(g['preset']), str(g['conv'])
    t_in, t_out = g['x'].shape[1], g['y'].shape[1]
    half, full = g['mask_half'].shape, g['mask'].shape
    gs_h = gs_f = None
    if preset != 'False':
        hom = preset == 'homogeneous'
        gs_h = O.static_graph(half, 4, g['mask_half'], use_edge_attrs=True, resolution=1 / 6, homogeneous=hom)
        gs_f = O.static_graph(full, 4, g['mask'], use_edge_attrs=True, resolution=1 / 12, homogeneous=hom)
        for gs, pre in ((gs_h, 'half_'), (gs_f, 'full_')):
            assert np.array_equal(gs['labels'], g[pre + 'labels'])
            assert np.array_equal(gs['n_pixels_per_node'].numpy(), g[pre + 'npix'])
            assert np.array_equal(gs['edge_index'].numpy(), g[pre + 'edges'])
            np.testing.assert_allclose(gs['edge_attrs'].numpy(), g[pre + 'attrs'], rtol=2e-5, atol=2e-5)
    model = O.Seq2Seq(32, 0.1, -np.inf, input_timesteps=t_in, input_features=8, output_timesteps=t_out, n_layers=1,
                      n_conv_layers=3, transform_func=dist_from_05, convolution_type=conv)
    _load_state(model, g, 'w/')
    model.eval()
    clim = {half: torch.from_numpy(_climatology(g['clim_base_half'])), full: torch.from_numpy(_climatology(g['clim_base']))}

    def concat_of(shape, launch):
        return torch.moveaxis(clim[shape][:, _doys(launch, t_out)], 0, -1)
    x, y = torch.from_numpy(g['x'][0]), torch.from_numpy(g['y'][0])
    outs, maps, _ = model(x, concat_of(full, g['launch'][0]), mask=g['mask'], graph_structure=gs_f)
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().numpy(), g[f'out_{i}'], rtol=RTOL, atol=ATOL)
    loss = O.clip_loss(outs, maps, y, full, g['mask'])
    assert abs(float(loss.detach()) - float(g['loss0'])) <= RTOL * float(g['loss0'])
    loss.backward()
    for k, p in model.named_parameters():
        ref = g['g/' + k]
        got = p.grad.numpy() if p.grad is not None else np.zeros_like(ref)
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-4 * max(1e-3, np.abs(ref).max()), err_msg=k)
    model.zero_grad()

    opt = torch.optim.Adam(model.parameters(), lr=float(g['lr']))               # mpnnlstm.py:174, kept across train() calls
    train_loss, test_loss = [], []

    def phase(xs, ys, shape, mask, gs, prefix):
        run = sum(O.train_step(model, opt, torch.from_numpy(xs[c]), torch.from_numpy(ys[c]), concat_of(shape, g['launch'][c]), mask,
                               graph_structure=gs) for c in range(2))
        with torch.no_grad():
            o, m, _ = model(torch.from_numpy(xs[2]), concat_of(shape, g['launch'][0]), mask=mask, graph_structure=gs)
            tl = float(O.clip_loss(o, m, torch.from_numpy(ys[2]), shape, mask))
        train_loss.append(run / 3)                                              # running / (step + 1), mpnnlstm.py:360-361
        test_loss.append(tl / 2)
        for k, v in model.state_dict().items():
            np.testing.assert_allclose(v.numpy(), g[prefix + k], rtol=1e-4, atol=2e-5, err_msg=prefix + k)
    if preset != 'False':
        phase(g['x_half'], g['y_half'], half, g['mask_half'], gs_h, 'w1/')
    phase(g['x'], g['y'], full, g['mask'], gs_f, 'w2/')
    np.testing.assert_allclose(train_loss, g['train_loss'], rtol=1e-4)
    np.testing.assert_allclose(test_loss, g['test_loss'], rtol=1e-4)
