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


def test_checkpoint_written_by_the_reference_loads_and_round_trips(tmp_path):
    """NextFramePredictorS2S.load() reads a .pth written the way the reference writes it
    (model/mpnnlstm.py:161-168: torch.save(model.state_dict()) of the reference's own Seq2Seq -- tests/golden/ref_checkpoint_h8.pth,
    made by make_golden.py) with the weights-only loader, the loaded model reproduces the reference's eval-mode rollout, and
    save() writes a file with the same keys, key order, shapes and values that a second instance loads back bit for bit."""
    import os
    import shutil
    from helpers import GOLDEN
    from model.mpnnlstm import NextFramePredictorS2S
    g = golden('checkpoint_case.npz')
    shutil.copy(os.path.join(GOLDEN, 'ref_checkpoint_h8.pth'), tmp_path / 'refrun.pth')
    mk = lambda name: NextFramePredictorS2S(thresh=0.1, experiment_name=name, input_features=1, input_timesteps=3,
                                            output_timesteps=3, device=dev(),
                                            model_kwargs=dict(hidden_size=8, dropout=0.0, n_layers=2, n_conv_layers=2))
    nfp = mk('refrun')
    nfp.load(str(tmp_path))
    ref_sd = torch.load(tmp_path / 'refrun.pth', map_location='cpu', weights_only=True)
    assert list(ref_sd.keys()) == [str(k) for k in g['keys']] == list(nfp.model.state_dict().keys())
    nfp.model.eval()
    x, concat = torch.from_numpy(g['x']).to(dev()), torch.from_numpy(g['concat']).to(dev())
    with torch.no_grad():
        outs, meshes = nfp.model(x, None, concat, teacher_forcing_ratio=0, mask=g['mask'])
    for i, o in enumerate(outs):
        assert o.shape[0] == g[f'out_{i}'].shape[0], f'mesh size of step {i}'
        close(o, g[f'out_{i}'], msg=f'step {i}')
    # save -> a second instance loads it back; the file has the reference's layout
    nfp.experiment_name = 'resaved'
    nfp.save(str(tmp_path))
    sd2 = torch.load(tmp_path / 'resaved.pth', map_location='cpu', weights_only=True)
    assert list(sd2.keys()) == list(ref_sd.keys())
    for k in ref_sd:
        assert sd2[k].shape == ref_sd[k].shape and torch.equal(sd2[k], ref_sd[k]), k
    other = mk('resaved')
    other.load(str(tmp_path))
    for (ka, a), (kb, b) in zip(nfp.model.state_dict().items(), other.model.state_dict().items()):
        assert ka == kb and torch.equal(a, b), ka
    # a loaded model trains: the flat parameter buffer is rebuilt around the loaded values
    other.initiate_training(lr=1e-3, lr_decay=0.95)
    other.model.train()
    y = torch.from_numpy(g['y']).to(dev())
    ls = [float(other.train_step(x, y, concat, g['mask'])) for _ in range(3)]
    assert np.isfinite(ls).all() and ls[-1] < ls[0], ls


def test_test_threshold_smoke():
    """NextFramePredictorS2S.test_threshold (model/mpnnlstm.py:138-156): the mesh of a few frames at a trial threshold, the
    per-node means painted back onto the image, one panel per frame, the node count in the title (Agg backend)."""
    import matplotlib
    matplotlib.use('Agg')
    from model.mpnnlstm import NextFramePredictorS2S
    from qtmpnn import synthetic
    nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=3, output_timesteps=3, device=dev(),
                                model_kwargs=dict(hidden_size=8, dropout=0.0, n_layers=1))
    clip = synthetic.make_clip(5, n_digits=1, n_frames=2, pixel_noise=0.0)             # (2, 64, 64, 1)
    x = torch.from_numpy(clip).to(dev())
    fig, axs = nfp.test_threshold(x, 0.1, mask=np.zeros((64, 64), dtype=bool))
    assert len(axs) == 2
    title = fig._suptitle.get_text()
    n_nodes = int(title.split('Num. nodes:')[1])
    assert title.startswith('Threshold: 0.1') and 1 < n_nodes < 64 * 64
    # the panel shows flatten -> unflatten of channel 0: pixels of one cell share the cell mean
    img = axs[0].get_images()[0].get_array()
    assert img.shape == (64, 64) and abs(float(img.mean()) - float(clip[0, ..., 0].mean())) < 1e-5
    import matplotlib.pyplot as plt
    plt.close(fig)


@pytest.mark.parametrize('name', ['ice_exp9', 'ice_exp10', 'ice_exp1'])
def test_ice_exp_preset_experiments_golden(name):
    """ice_exp.py exp 9 / 10 / 1 against traces of the reference's own trainer (tests/golden/make_golden.py::ice_exp_preset_cases):
    preset heterogeneous / homogeneous mesh x max_grid_size 4 x use_edge_attrs=True x resolution 1/6 and 1/12 x TransformerConv
    x hidden 32 x 3 conv layers x land mask, and GCNConv on the pixelwise mesh.  Meshes: labels / npix / edge lists bit-exact,
    [angle, dist] at 2e-5; forward + backward at the initial weights: outputs, loss, all gradients at 1e-4; then the script's
    sequence -- train() at half resolution, train() again at full resolution with the same model and optimizer, predict() --
    per-epoch losses, the weights after each phase and the prediction."""
    from model.graph_functions import create_static_heterogeneous_graph, create_static_homogeneous_graph
    from model.mpnnlstm import NextFramePredictorS2S, masked_mse
    g = golden(f'variant_{name}.npz')
    preset, conv = str(g['preset']), str(g['conv'])
    t_in, t_out = g['x'].shape[1], g['y'].shape[1]
    half, full = g['mask_half'].shape, g['mask'].shape
    kw = dict(hidden_size=32, dropout=0.1, n_layers=1, transform_func=dist_from_05, dummy=False, n_conv_layers=3,
              rnn_type='LSTM', convolution_type=conv)
    nfp = NextFramePredictorS2S(thresh=-np.inf, experiment_name=name, input_features=5, input_timesteps=t_in,
                                output_timesteps=t_out, device=dev(), transform_func=dist_from_05, binary=False, debug=False,
                                model_kwargs=kw)
    assert nfp.get_n_params() == int(g['n_params'])
    load_state(nfp.model, g, 'w/')
    nfp.model.eval()                                   # like the golden run (dropout cannot be RNG matched)
    gs_h = gs_f = None
    if preset != 'False':
        make = create_static_heterogeneous_graph if preset == 'heterogeneous' else create_static_homogeneous_graph
        gs_h = make(half, 4, g['mask_half'], use_edge_attrs=True, resolution=1 / 6, device=dev())
        gs_f = make(full, 4, g['mask'], use_edge_attrs=True, resolution=1 / 12, device=dev())
        for gs, pre, shape in ((gs_h, 'half_', half), (gs_f, 'full_', full)):
            mesh = gs['mapping']
            assert np.array_equal(mesh.labels[0].cpu().numpy(), g[pre + 'labels']), pre + 'labels'
            assert np.array_equal(gs['n_pixels_per_node'].cpu().numpy(), g[pre + 'npix']), pre + 'npix'
            assert np.array_equal(gs['edge_index'].cpu().numpy(), g[pre + 'edges']), pre + 'edges'
            close(gs['edge_attrs'], g[pre + 'attrs'], rtol=2e-5, atol=2e-5, msg=pre + 'attrs')
    clim_h = torch.from_numpy(climatology_from_base(g['clim_base_half'])).to(dev())
    clim_f = torch.from_numpy(climatology_from_base(g['clim_base'])).to(dev())
    launch, mask, hir = g['launch'], g['mask'], g['hir']

    # forward + backward at the initial weights (full resolution, clip 0)
    x, y = torch.from_numpy(g['x'][0]).to(dev()), torch.from_numpy(g['y'][0]).to(dev())
    concat = nfp.get_climatology_array(clim_f, torch.tensor([launch[0]]))
    outs, meshes = nfp.model(x, y, concat, teacher_forcing_ratio=0, mask=mask, high_interest_region=hir, graph_structure=gs_f)
    for i, o in enumerate(outs):
        assert o.shape[0] == g[f'out_{i}'].shape[0], f'mesh size of step {i}'
        close(o, g[f'out_{i}'], msg=f'step {i}')
    loss = masked_mse(outs, meshes, y, mask)
    assert abs(float(loss) - float(g['loss0'])) <= 1e-4 * float(g['loss0'])
    loss.backward()
    for k, p in nfp.model.named_parameters():
        ref = g['g/' + k]
        if p.grad is None:
            assert not ref.any(), k
            continue
        grad_close(p.grad, ref, msg=k, floor=0.05 if k.endswith('lin_key.bias') else 1e-3)
    nfp.model.zero_grad(set_to_none=True)

    # the script's sequence through train() / predict(); loaders in fixed order like the golden run
    def loader(xs, ys, shape, lo, hi):
        return TinyLoader([(torch.from_numpy(xs[c])[None], torch.from_numpy(ys[c])[None], torch.tensor([launch[c - lo]]))
                           for c in range(lo, hi)], shape)
    lr, steps = float(g['lr']), 0

    def weights_close(prefix, steps):
        # Adam moves a weight by at most lr per step, whatever the size of its gradient: an entry of 1e-7 that carries 1e-9 of
        # summation-order noise moves its weight by lr * (1 +- 0.01), and an entry that is rounding noise of an exact zero by
        # lr * g / (|g| + eps) either way.  Hold the weights to 1e-4 relative plus 3 % of that reach (the gradients themselves
        # are held to 1e-4 above; the key projections of a softmax over 2 - 5 neighbours are where such entries live)
        for k, v in nfp.model.state_dict().items():
            close(v, g[prefix + k], rtol=1e-4, atol=0.03 * steps * lr, msg=prefix + k)
    if preset != 'False':
        nfp.train(loader(g['x_half'], g['y_half'], half, 0, 2), loader(g['x_half'], g['y_half'], half, 2, 3), clim_h, lr=lr,
                  n_epochs=1, mask=g['mask_half'], truncated_backprop=0, graph_structure=gs_h)
        steps += 2
        weights_close('w1/', steps)
    nfp.train(loader(g['x'], g['y'], full, 0, 2), loader(g['x'], g['y'], full, 2, 3), clim_f, lr=lr, n_epochs=1, mask=mask,
              high_interest_region=hir, truncated_backprop=0, graph_structure=gs_f)
    steps += 2
    weights_close('w2/', steps)
    np.testing.assert_allclose(nfp.loss['train_loss'].values, g['train_loss'], rtol=2e-4)
    np.testing.assert_allclose(nfp.loss['test_loss'].values, g['test_loss'], rtol=2e-4)
    pred = nfp.predict(loader(g['x'], g['y'], full, 2, 3), clim_f, mask=mask, graph_structure=gs_f)
    assert pred.shape == g['pred'].shape and np.array_equal(np.isnan(pred), np.isnan(g['pred']))
    np.testing.assert_allclose(np.nan_to_num(pred), np.nan_to_num(g['pred']), rtol=2e-4, atol=2e-5)


def test_edge_inputs_run_or_fail_loudly():
    """Inputs at the edges of what the reference's callers pass: float64 arrays and non-contiguous tensors give the float32 loss; a
    mask may be a numpy array, a CPU or a CUDA tensor; frames smaller than a base cell (8 x 8, 1 x 7) and constant frames (coarsest
    possible input mesh) train; NaNs in the input raise image_to_graph's ValueError (graph_functions.py:626-627) on the eager step; a mask
    that covers every pixel raises instead of dividing by zero nodes."""
    from model.mpnnlstm import NextFramePredictorS2S
    from qtmpnn import synthetic

    def fresh(conv='ChebConv'):
        torch.manual_seed(0)
        nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=3, output_timesteps=2, device=dev(),
                                    model_kwargs=dict(hidden_size=8, dropout=0.0, n_layers=1, convolution_type=conv))
        nfp.initiate_training(lr=1e-3, lr_decay=0.95)
        nfp.model.train()
        return nfp
    x, y = synthetic.make_batch(5, 0, 2, 3, 2, n_digits=1, pixel_noise=0.0)
    xt, yt = torch.from_numpy(x).to(dev()), torch.from_numpy(y).to(dev())
    mask = np.zeros((64, 64), dtype=bool)
    mask[:, 40:] = True
    base = float(fresh().train_step(xt, yt, None, mask=mask))
    assert np.isfinite(base)
    assert float(fresh().train_step(xt.double(), yt.double(), None, mask=mask)) == base
    swapped = xt.permute(0, 1, 3, 2, 4).contiguous().permute(0, 1, 3, 2, 4)
    assert not swapped.is_contiguous() and float(fresh().train_step(swapped, yt, None, mask=mask)) == base
    for m in (torch.from_numpy(mask), torch.from_numpy(mask).to(dev()), mask.astype(np.uint8), mask.tolist()):
        assert float(fresh().train_step(xt, yt, None, mask=m)) == base
    for shape in ((8, 8), (1, 7)):
        xs, ys = torch.rand(2, 3, *shape, 1, device=dev()), torch.rand(2, 2, *shape, 1, device=dev())
        for conv in ('ChebConv', 'GCNConv', 'TransformerConv'):
            assert np.isfinite(float(fresh(conv).train_step(xs, ys, None, mask=np.zeros(shape, dtype=bool)))), (shape, conv)
    for conv in ('ChebConv', 'TransformerConv'):
        assert np.isfinite(float(fresh(conv).train_step(torch.zeros_like(xt), torch.zeros_like(yt), None, mask=mask)))
    bad = xt.clone()
    bad[1, 0, 3, 3, 0] = float('nan')
    with pytest.raises(ValueError, match='Found NaNs in image data 1 / '):
        fresh().train_step(bad, yt, None, mask=mask)
    with pytest.raises(ValueError, match='covers every pixel'):
        fresh().train_step(xt, yt, None, mask=np.ones((64, 64), dtype=bool))
    # targets the loss kernels would read out of bounds, and a preset mesh of another frame size: refused by name
    for bad_y in (torch.zeros(2, 2, 32, 32, 1, device=dev()), yt[..., 0], yt[:, :1], torch.cat([yt, yt], dim=-1)):
        with pytest.raises(ValueError, match='targets of shape'):
            fresh().train_step(xt, bad_y, None, mask=mask)
    from model.graph_functions import create_static_heterogeneous_graph
    small = create_static_heterogeneous_graph((32, 32), 8, np.zeros((32, 32), dtype=bool), use_edge_attrs=False, device=dev())
    with pytest.raises(ValueError, match='graph_structure was built for 32 x 32'):
        fresh().train_step(xt, yt, None, mask=mask, graph_structure=small)


@pytest.mark.parametrize('conv,h,nl,nc', [('ChebConv', 8, 1, 2), ('ChebConv', 32, 2, 3), ('ChebConv', 64, 1, 3), ('ChebConv', 128, 1, 1),
                                          ('ChebConv', 16, 2, 4), ('GCNConv', 16, 1, 2), ('GCNConv', 64, 2, 2), ('GCNConv', 128, 1, 2),
                                          ('TransformerConv', 8, 1, 2), ('TransformerConv', 16, 2, 1), ('TransformerConv', 32, 1, 2)])
def test_every_supported_hidden_size_trains(conv, h, nl, nc):
    """The hidden sizes the kernels are built for, at the largest stacks whose composed gate matrix still fits the GEMM kernels' 512
    rows (Seq2Seq refuses the others at construction: tests/test_host_cpu.py): one training step through the trainer runs and moves
    the loss; a frame with the wrong channel count is refused by name."""
    from model.mpnnlstm import NextFramePredictorS2S
    from qtmpnn import synthetic
    torch.manual_seed(0)
    nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=3, output_timesteps=2, device=dev(),
                                model_kwargs=dict(hidden_size=h, dropout=0.0, n_layers=nl, convolution_type=conv, n_conv_layers=nc))
    nfp.initiate_training(lr=1e-3, lr_decay=0.95)
    nfp.model.train()
    x, y = synthetic.make_batch(5, 0, 2, 3, 2, n_digits=1, pixel_noise=0.0)
    xt, yt = torch.from_numpy(x).to(dev()), torch.from_numpy(y).to(dev())
    mask = np.zeros((64, 64), dtype=bool)
    l0, l1 = float(nfp.train_step(xt, yt, None, mask=mask)), float(nfp.train_step(xt, yt, None, mask=mask))
    assert np.isfinite(l0) and np.isfinite(l1) and l1 != l0
    with pytest.raises(ValueError, match='channels'):
        nfp.train_step(torch.cat([xt, xt], dim=-1), yt, None, mask=mask)


def test_modules_refuse_node_tensors_of_another_mesh():
    """The modules take the Mesh where the reference takes edge_index; their kernels walk the mesh's rows and read the node tensors
    unchecked, so a tensor with another row count (the reference would die in an index error inside PyG) is refused before a launch."""
    from model.model import ChebConv, GConvLSTM, TransformerConv
    from model.seq2seq import Decoder, Encoder
    from qtmpnn import synthetic
    from qtmpnn.mesh import build_mesh
    c = synthetic.make_clip(31, canvas=(64, 64), n_digits=1, n_frames=1, pixel_noise=0.0)
    mesh = build_mesh(src=torch.from_numpy(c[:, ..., 0]).to(dev()), thresh=0.1, mask=None)
    N = mesh.N
    good, short = torch.randn(N, 4, device=dev()), torch.randn(N - 1, 4, device=dev())
    h_ok, h_bad = torch.randn(N, 8, device=dev()), torch.randn(N + 3, 8, device=dev())
    for conv in (ChebConv(4, 8).to(dev()), TransformerConv(4, 8).to(dev())):
        assert conv(good, mesh).shape == (N, 8)
        with pytest.raises(ValueError, match='rows for a mesh'):
            conv(short, mesh)
    cell = GConvLSTM(4, 8, 1, 'ChebConv').to(dev())
    assert cell(good, mesh, None, h_ok, h_ok)[1].shape == (N, 8)
    for args in ((short, mesh, None, h_ok, h_ok), (good, mesh, None, h_bad, h_ok), (good, mesh, None, h_ok, h_bad)):
        with pytest.raises(ValueError, match='rows for a mesh'):
            cell(*args)
    enc = Encoder(4, 8, 0.0, n_layers=1, convolution_type='ChebConv', n_conv_layers=1).to(dev())
    dec = Decoder(4, 8, 0.0, n_layers=1, concat_layers_dim=1, convolution_type='ChebConv', n_conv_layers=1).to(dev())
    H, C = enc(good, mesh)
    assert H.shape == (1, N, 8)
    with pytest.raises(ValueError, match='rows for a mesh'):
        enc(short, mesh)
    with pytest.raises(ValueError, match='rows for a mesh'):
        enc(good, mesh, None, H=h_bad, C=h_ok)
    y, H2, C2 = dec(good, mesh, None, torch.randn(N, 1, device=dev()), H, C)
    assert y.shape == (N, 1)
    with pytest.raises(ValueError, match='rows for a mesh'):
        dec(good, mesh, None, torch.randn(N - 2, 1, device=dev()), H, C)
    with pytest.raises(ValueError, match='rows for a mesh'):
        dec(good, mesh, None, torch.randn(N, 1, device=dev()), H[:, :-1], C)
