"""SURVEY 8(f) rows 1-3 on the configuration the shipped ice scripts really run (ice_exp.py:48,145,153-162: pixelwise mesh x
TransformerConv x hidden 32 x 3 conv layers x climatology concat), the trainer's predict() / get_climatology_array
(model/mpnnlstm.py:389-440), and teacher forcing on meshes that are not rebuilt (model/seq2seq.py:420-425) -- all against
traces captured by running the reference's own classes (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from helpers import TinyLoader, climatology_from_base, close, dev, dist_from_05, golden, grad_close, load_state

pytestmark = pytest.mark.gpu


def _ice_predictor(g):
    from model.mpnnlstm import NextFramePredictorS2S
    kw = dict(hidden_size=32, dropout=0.1, n_layers=1, transform_func=dist_from_05, dummy=False, n_conv_layers=3,
              rnn_type='LSTM', convolution_type='TransformerConv')
    nfp = NextFramePredictorS2S(thresh=-np.inf, input_features=5, input_timesteps=3, output_timesteps=3, device=dev(),
                                transform_func=dist_from_05, model_kwargs=kw)
    assert nfp.get_n_params() == int(g['n_params'])
    load_state(nfp.model, g, 'w/')
    nfp.model.eval()                   # attention / decoder dropout off, like the golden run
    return nfp


def test_get_climatology_array_matches_reference():
    """Day-of-year lookup incl. the wrap over the year end (launch 30 Dec -> indices 363, 364, 0)."""
    g = golden('variant_ice_exp.npz')
    nfp = _ice_predictor(g)
    clim = torch.from_numpy(climatology_from_base(g['clim_base'])).to(dev())
    for c in range(2):
        got = nfp.get_climatology_array(clim, torch.tensor([g['launch'][c]]))
        assert tuple(got.shape) == (3, 24, 32, 1)
        assert np.array_equal(got.cpu().numpy(), g[f'concat_{c}'])
    assert np.array_equal(g['concat_1'][2, ..., 0], climatology_from_base(g['clim_base'])[0, 0])


def test_ice_exp_configuration_golden():
    """Pixelwise mesh (no self pairs: get_adj_pixelwise) x TransformerConv x hidden 32 x n_conv_layers 3 x land mask x
    climatology concat: outputs of both clips, losses, and every gradient of clip 0."""
    from model.mpnnlstm import masked_mse
    g = golden('variant_ice_exp.npz')
    nfp = _ice_predictor(g)
    clim = torch.from_numpy(climatology_from_base(g['clim_base'])).to(dev())
    mask = g['mask']
    for c in range(2):
        x, y = torch.from_numpy(g['x'][c]).to(dev()), torch.from_numpy(g['y'][c]).to(dev())
        concat = nfp.get_climatology_array(clim, torch.tensor([g['launch'][c]]))
        nfp.model.zero_grad(set_to_none=True)
        outs, meshes = nfp.model(x, y, concat, teacher_forcing_ratio=0, mask=mask)
        assert meshes[0].pixelwise and meshes[0].attn_geometry()[1] is None          # no self pairs on pixelwise meshes
        for i, o in enumerate(outs):
            assert o.shape[0] == g[f'out_{c}_{i}'].shape[0] == int((~mask).sum())
            close(o, g[f'out_{c}_{i}'], msg=f'clip {c} step {i}')
        loss = masked_mse(outs, meshes, y, mask)
        assert abs(float(loss) - float(g[f'loss_{c}'])) <= 1e-4 * float(g[f'loss_{c}'])
        if c == 0:
            loss.backward()
            for k, p in nfp.model.named_parameters():
                ref = g['g/' + k]
                if p.grad is None:
                    assert not ref.any(), k
                    continue
                grad_close(p.grad, ref, msg=k, floor=0.05 if k.endswith('lin_key.bias') else 1e-3)


def test_predict_layout_and_values_golden():
    """predict(): (n_clips, T_out, W, H, 1) with NaN under the mask (unflatten_pixelwise), values = the reference's; the
    batched call (both clips at once) gives the same array."""
    g = golden('variant_ice_exp.npz')
    nfp = _ice_predictor(g)
    clim = torch.from_numpy(climatology_from_base(g['clim_base'])).to(dev())
    items = [(torch.from_numpy(g['x'][c])[None], torch.from_numpy(g['y'][c])[None], torch.tensor([g['launch'][c]])) for c in range(2)]
    pred = nfp.predict(TinyLoader(items, (24, 32)), clim, mask=g['mask'])
    ref = g['pred']
    assert pred.shape == ref.shape == (2, 3, 24, 32, 1)
    assert np.array_equal(np.isnan(pred), np.isnan(ref))
    assert np.isnan(pred[:, :, g['mask']]).all() and not np.isnan(pred[:, :, ~g['mask']]).any()
    np.testing.assert_allclose(np.nan_to_num(pred), np.nan_to_num(ref), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('name', ['teacher_pixelwise', 'teacher_static', 'teacher_every2'])
def test_teacher_forcing_without_remesh_golden(name):
    """teacher_forcing_ratio = 1 where the mesh is not rebuilt: the next decoder input is [flatten(teacher + positional
    encoding) | RAW n_pixels_per_node] (model/seq2seq.py:420-425)."""
    from model.graph_functions import create_static_heterogeneous_graph
    from model.mpnnlstm import masked_mse
    from model.seq2seq import Seq2Seq
    g = golden(f'variant_{name}.npz')
    model = Seq2Seq(hidden_size=8, dropout=0.0, thresh=float(g['thresh']), input_timesteps=2, input_features=6, output_timesteps=4,
                    n_layers=2, n_conv_layers=1, convolution_type='ChebConv')
    load_state(model, g, 'w/')
    model.to(dev()).train()
    x, y, concat = (torch.from_numpy(g[k]).to(dev()) for k in ('x', 'y', 'concat'))
    gs = None
    if name == 'teacher_static':
        gs = create_static_heterogeneous_graph((32, 40), int(g['max_grid_size']), g['mask'], high_interest_region=g['hir'],
                                               use_edge_attrs=False, device=dev())
    outs, meshes = model(x, y, concat, teacher_forcing_ratio=1.0, mask=g['mask'], graph_structure=gs,
                         remesh_every=int(g['remesh_every']))
    for i, o in enumerate(outs):
        assert o.shape[0] == g[f'out_{i}'].shape[0], f'mesh size of step {i}'
        close(o, g[f'out_{i}'], msg=f'step {i}')
    loss = masked_mse(outs, meshes, y, g['mask'])
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * float(g['loss'])
    loss.backward()
    for k, p in model.named_parameters():
        ref = g['g/' + k]
        if p.grad is None:
            assert not ref.any(), k
            continue
        grad_close(p.grad, ref, msg=k)


@pytest.mark.parametrize('thresh', [0.1, -np.inf])
def test_continued_unroll_equals_one_unroll(thresh):
    """unroll_output(range(0, 2)) followed by unroll_output(range(2, 4)) without process_inputs in between continues from the
    updated state, like the reference (whose unroll_output updates the graph after every step, the last included)."""
    from model.seq2seq import Seq2Seq
    g = golden('variant_teacher.npz')
    torch.manual_seed(2)
    model = Seq2Seq(hidden_size=8, dropout=0.0, thresh=thresh, input_timesteps=3, input_features=4, output_timesteps=4,
                    n_layers=1, n_conv_layers=2, convolution_type='ChebConv').to(dev())
    x, y, concat = (torch.from_numpy(g[k]).to(dev()) for k in ('x', 'y', 'concat'))
    mask = g['mask']
    with torch.no_grad():
        whole, _ = model(x, y, concat, teacher_forcing_ratio=0, mask=mask)
        model.process_inputs(x, mask=mask)
        a, _ = model.unroll_output(range(0, 2), y, concat_layers=concat, teacher_forcing_ratio=0, mask=mask)
        b, _ = model.unroll_output(range(2, 4), y, concat_layers=concat, teacher_forcing_ratio=0, mask=mask)
    for i, (o, r) in enumerate(zip(a + b, whole)):
        assert o.shape == r.shape, f'step {i}'
        close(o, r, rtol=1e-6, atol=1e-7, msg=f'step {i}')
