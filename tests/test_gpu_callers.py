"""The drop-in boundary on the callers the reference ships: the call sequences of moving_mnist_example.ipynb
(cells 0-7) and of ice_exp.py:109-224 -- the calls into the package, in order, with the callers' imports, keyword arguments and
defaults -- with stand-ins only
for the two dataset classes (the reference's download MNIST / read ERA5 files) and smaller sample counts / epochs.

Where HEAD of the reference crashes on these very sequences the build defines a behaviour;
the tests at the bottom pin each of them: `mask=None` == an all-False mask, `concat_layers=None` == the decoder's current
input value as the one concat channel, `truncated_backprop` beyond the rollout length == one chunk of all steps."""
import os

import numpy as np
import pytest
import torch

from helpers import TinyIceDataset, TinyMovingMNISTDataset, climatology_from_base, close, dev

pytestmark = pytest.mark.gpu


def test_notebook_cells_run_as_written(tmp_path, monkeypatch):
    """moving_mnist_example.ipynb cells 0-7, the calls into the package with the notebook's arguments: the imports of cells 0 and 2
    (incl. the abstract NextFramePredictor), the predictor built with the notebook's kwargs (input_timesteps left at its default 3
    although the clips carry 4 input frames), test_threshold on a CPU clip, `model.train(loader, loader, lr=0.01, n_epochs=1)` under
    cProfile, a second predictor trained with `model.train(loader_train, loader_test, lr=0.01, n_epochs=...)` -- mask=None, no
    climatology, truncated_backprop=45 > T_out -- `model.loss.plot()` and `model.predict(loader_val)`.  Stand-ins: the dataset class
    (the notebook's downloads MNIST) and the sample / epoch counts."""
    import cProfile
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    from torch.utils.data import DataLoader
    monkeypatch.chdir(tmp_path)                      # (a SummaryWriter, when tensorboard is installed, writes runs/ here)
    # cells 0 and 2: what the notebook imports from the package
    from model.utils import normalize, add_positional_encoding  # noqa: F401
    from model.mpnnlstm import NextFramePredictorS2S, NextFramePredictor
    from model.model import MPNNLSTM, MPNNLSTMI  # noqa: F401
    from model.graph_functions import image_to_graph, flatten, Graph, unflatten  # noqa: F401
    torch.manual_seed(1)
    # cell 1: 4 input frames + 10 output frames of one 18 x 18 digit on a 32 x 32 canvas, DataLoader(batch_size=1)
    t_in, t_out = 4, 10
    kw = dict(input_timesteps=t_in, output_timesteps=t_out, n_digits=1, gap=0, canvas_size=(32, 32), digit_size=(18, 18),
              pixel_noise=0.05, velocity_noise=0.0)
    data_train, data_test, data_val = (TinyMovingMNISTDataset(n, seed=i, **kw) for i, n in enumerate((6, 3, 3)))   # (200 / 50 / 50)
    loader_train, loader_test = (DataLoader(d, batch_size=1, shuffle=True) for d in (data_train, data_test))
    loader_val = DataLoader(data_val, batch_size=1, shuffle=False)
    device = torch.device('cuda:0')

    def predictor():            # cells 2 and 5
        return NextFramePredictorS2S(
            thresh=0.1,
            experiment_name='test',
            decompose=True,
            input_features=1,
            device=device,
            output_timesteps=t_out,
            remesh_input=False,
            model_kwargs=dict(hidden_size=16, dropout=0.1, n_layers=2))
    model = predictor()
    assert model.get_n_params() == 34513                                        # SURVEY KAT-6
    assert isinstance(model, NextFramePredictor) and NextFramePredictor.__abstractmethods__ == {'train', 'predict', 'score'}
    with pytest.raises(TypeError):
        NextFramePredictor(thresh=0.1)                                         # abstract, like the reference's
    # cell 3: a clip as the loader hands it over (CPU tensor)
    x = next(iter(loader_val))[0].squeeze(0)
    for th in (1.5, 0.85, 0.5, 0.15):
        fig, axs = model.test_threshold(x, thresh=th)
        assert len(axs) == t_in
        plt.close(fig)
    # cell 4
    loader_profile = DataLoader(data_train, batch_size=1, sampler=torch.utils.data.SubsetRandomSampler(range(4)))
    cProfile.runctx('model.train(loader_profile, loader_profile, lr=0.01, n_epochs=1)', globals(), locals(), sort=1)
    assert len(model.loss) == 1 and np.isfinite(model.loss.values).all()
    # cells 5, 6
    model = predictor()
    before = {k: v.clone() for k, v in model.model.state_dict().items()}
    model.train(loader_train, loader_test, lr=0.01, n_epochs=3)                # (n_epochs=20 in the notebook)
    plt.close(model.loss.plot().figure)
    assert list(model.loss.columns) == ['train_loss', 'test_loss'] and len(model.loss) == 3
    assert np.isfinite(model.loss.values).all() and (model.loss.values < 4).all()
    assert model.loss.train_loss.iloc[-1] < model.loss.train_loss.iloc[0]
    assert any(not torch.equal(v, before[k]) for k, v in model.model.state_dict().items())
    # cells 7, 8
    y_hat = model.predict(loader_val)
    assert y_hat.shape == (3, t_out, 32, 32, 1) and np.isfinite(y_hat).all()
    assert loader_val.dataset.x[0][0, ..., 0].shape == y_hat[0][0][..., 0].shape


EXPERIMENTS = {0: ('TransformerConv', None), 1: ('GCNConv', None), 9: ('TransformerConv', 'heterogeneous'), 10: ('TransformerConv', 'homogeneous'),
               'nwt': ('TransformerConv', None)}


@pytest.mark.parametrize('exp', [9, 10, 1, 0, 'nwt'])
def test_ice_exp_sequence_runs_as_written(exp, tmp_path, monkeypatch):
    """The calls `ice_exp.py` makes into the package, in its order and with its keyword arguments (:109-112, 127-130, 153-176, 181,
    185-206, 214-224): preset heterogeneous (exp 9) / homogeneous (exp 10) meshes with `max_grid_size=4, use_edge_attrs=True,
    resolution=1/6 | 1/12`, TransformerConv x hidden 32 x 3 conv layers, `debug=True`, `binary=`, train() at half resolution (no
    climatology, as the script) and again at full resolution with ONE model, loss.to_csv, save, eval, predict with the preset mesh;
    exp 1 = GCNConv on the pixelwise mesh, exp 0 = the script's defaults (TransformerConv, pixelwise); 'nwt' = ice_exp_nwt.py:46-142,
    the same calls without a climatology in train() and predict() and lr 0.001.  Only the data (the script
    reads ERA5 / GLORYS files through xarray) and the sizes are stand-ins."""
    from torch.utils.data import DataLoader
    from model.utils import int_to_datetime
    from model.mpnnlstm import NextFramePredictorS2S
    from model.graph_functions import create_static_heterogeneous_graph, create_static_homogeneous_graph
    from qtmpnn import synthetic
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(21)
    device = torch.device('cuda:0')
    convolution_type, preset_mesh = EXPERIMENTS[exp]
    make_mesh = {'heterogeneous': create_static_heterogeneous_graph, 'homogeneous': create_static_homogeneous_graph}.get(preset_mesh)
    lr, truncated_backprop, binary, t_in, t_out = 0.0001, 0, False, 3, 4                # (the script: in 10, out 90)

    def loaders(shape, seed, n_val=0):
        sets = [TinyIceDataset(n, t_in, t_out, shape, seed=seed + i, first_day=(2010, 12, 30) if i == 2 else (2010, 3, 1))
                for i, n in enumerate((3, 2, n_val)) if n]
        return [DataLoader(d, batch_size=1, shuffle=i < 2) for i, d in enumerate(sets)]

    mask = synthetic.make_ice_like(32, shape=(48, 64), channels=1, n_frames=1)[1]
    high_interest_region = np.zeros_like(mask)
    high_interest_region[10:20, 30:44] = True
    graph_structure = graph_structure_half = None
    if preset_mesh:
        mask_half = synthetic.make_ice_like(31, shape=(24, 32), channels=1, n_frames=1)[1]
        loader_train_half, loader_test_half = loaders(mask_half.shape, 10)
        graph_structure_half = make_mesh(mask_half.shape, 4, mask_half, use_edge_attrs=True, resolution=1/6, device=device)
        graph_structure = make_mesh(mask.shape, 4, mask, use_edge_attrs=True, resolution=1/12, device=device)
    loader_train, loader_test, loader_val = loaders(mask.shape, 12, n_val=2)
    base = synthetic.make_ice_like(33, shape=mask.shape, channels=1, n_frames=1)[0][0, ..., 0]
    climatology = torch.tensor(np.nan_to_num(climatology_from_base(base))).to(device) if exp != 'nwt' else None
    if exp == 'nwt':
        lr, high_interest_region = 0.001, None

    def dist_from_05(arr):
        return abs(abs(arr - 0.5) - 0.5)

    model = NextFramePredictorS2S(
        thresh=-np.inf,
        experiment_name='ice',
        input_features=5,
        input_timesteps=t_in,
        output_timesteps=t_out,
        transform_func=dist_from_05,
        device=device,
        binary=binary,
        debug=True,
        model_kwargs=dict(hidden_size=32, dropout=0.1, n_layers=1, transform_func=dist_from_05, dummy=False, n_conv_layers=3,
                          rnn_type='LSTM', convolution_type=convolution_type))
    assert model.get_n_params() > 0 and 'Seq2Seq' in repr(model.model)
    model.model.train()
    opt = None
    if preset_mesh:             # multires_training: half resolution first, WITHOUT climatology (HEAD fails here: DESIGN section 2)
        model.train(loader_train_half, loader_test_half, lr=lr, n_epochs=2, mask=mask_half, truncated_backprop=truncated_backprop,
                    graph_structure=graph_structure_half)
        assert len(model.loss) == 2
        opt = model.optimizer
    model.train(loader_train, loader_test, climatology, lr=lr, n_epochs=2, mask=mask, high_interest_region=high_interest_region,
                truncated_backprop=truncated_backprop, graph_structure=graph_structure)
    # the second train() call keeps the optimizer and appends to the loss history (mpnnlstm.py:203-205, 373-374)
    assert (opt is None or model.optimizer is opt) and len(model.loss) == (4 if preset_mesh else 2)
    assert np.isfinite(model.loss.values).all()
    os.makedirs('results')
    model.loss.to_csv('results/loss_ice.csv')
    model.save('results')
    assert os.path.exists('results/ice.pth')
    model.model.eval()
    val_preds = model.predict(loader_val, climatology, mask=mask, graph_structure=graph_structure)
    assert len([int_to_datetime(t) for t in loader_val.dataset.launch_dates]) == 2
    y_hat = val_preds.squeeze(-1)
    assert y_hat.shape == loader_val.dataset.y.squeeze(-1).shape == (2, t_out, *mask.shape)
    assert np.isfinite(y_hat[:, :, ~mask]).all()
    if not preset_mesh:
        assert np.isnan(y_hat[:, :, mask]).all()                                # unflatten_pixelwise: NaN under the mask


def _mnist_predictor(dropout=0.0, seed=3, t_out=4):
    from model.mpnnlstm import NextFramePredictorS2S
    torch.manual_seed(seed)
    return NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=3, output_timesteps=t_out, device=dev(),
                                 model_kwargs=dict(hidden_size=8, dropout=dropout, n_layers=1))


def test_mask_none_is_an_all_false_mask():
    """`train(..., mask=None)` (the notebook; HEAD: TypeError at `~mask`, mpnnlstm.py:246) trains exactly like an all-False mask."""
    from torch.utils.data import DataLoader
    data = TinyMovingMNISTDataset(3, 3, 4, canvas_size=(32, 32), digit_size=(18, 18), pixel_noise=0.0, velocity_noise=0.0)
    losses = []
    for mask in (None, np.zeros((32, 32), dtype=bool)):
        nfp = _mnist_predictor()
        loader = DataLoader(data, batch_size=1, shuffle=False)
        nfp.train(loader, loader, lr=0.01, n_epochs=2, mask=mask, truncated_backprop=0)
        losses.append(nfp.loss.values.copy())
        pred = nfp.predict(loader, mask=mask)
        assert pred.shape == (3, 4, 32, 32, 1)
    assert np.array_equal(losses[0], losses[1])


def test_truncated_backprop_beyond_the_rollout_is_one_chunk():
    """truncated_backprop=45 (the default) with T_out=4: HEAD unrolls range(-41, 5) and fails; here the chunk is clamped to
    range(0, 4): ONE chunk, whose loss and gradients are those of the full rollout (no gradient clipping in this branch)."""
    data = TinyMovingMNISTDataset(1, 3, 4, canvas_size=(32, 32), digit_size=(18, 18), pixel_noise=0.0, velocity_noise=0.0)
    x, y = torch.from_numpy(data.x[0]).to(dev()), torch.from_numpy(data.y[0]).to(dev())
    mask = np.zeros((32, 32), dtype=bool)
    nfp = _mnist_predictor()
    nfp.initiate_training(0.01, 0.95)
    chunk_losses = nfp.truncated_backward(x, y, None, mask, truncated_backprop=45)
    assert len(chunk_losses) == 1
    g_trunc = {k: p.grad.clone() for k, p in nfp.model.named_parameters() if p.grad is not None}
    nfp.zero_grad()
    loss = nfp.forward_loss(x, y, None, mask)
    loss.backward()
    assert float(loss.detach()) == float(chunk_losses[0])
    for k, p in nfp.model.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, g_trunc[k]), k
    # tb = 3 does not divide T_out = 4 either: chunks range(0, 3) and range(2, 4) (HEAD: range(2, 5), IndexError)
    assert len(nfp.truncated_backward(x, y, None, mask, truncated_backprop=3)) == 2


def test_concat_layers_none_uses_the_decoder_input_value():
    """No climatology (the notebook, ice_exp.py's half-resolution phase; HEAD: 16 vs 17 channels at fc_out1, seq2seq.py:115,164):
    the decoder's current input value is the one concat channel -- the same numbers as passing that channel explicitly on a mesh
    that never changes (pixelwise), where the input value of step t is known up front only for t = 0, so compare step 0."""
    from model.seq2seq import Seq2Seq
    torch.manual_seed(5)
    model = Seq2Seq(hidden_size=8, dropout=0.0, thresh=-np.inf, input_timesteps=2, input_features=4, output_timesteps=1,
                    n_layers=1, n_conv_layers=1).to(dev()).eval()
    data = TinyMovingMNISTDataset(1, 2, 1, canvas_size=(32, 32), digit_size=(18, 18), pixel_noise=0.02, velocity_noise=0.0)
    x = torch.from_numpy(data.x[0]).to(dev())
    mask = np.zeros((32, 32), dtype=bool)
    with torch.no_grad():
        a, _ = model(x, None, None, teacher_forcing_ratio=0, mask=mask)
        b, _ = model(x, None, x[-1:].clone(), teacher_forcing_ratio=0, mask=mask)      # concat = the last input frame itself
    close(a[0], b[0], rtol=1e-6, atol=1e-7)
