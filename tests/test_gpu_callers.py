"""The drop-in boundary on the callers the reference ships: the call sequences of moving_mnist_example.ipynb
(cells 0-7) and of ice_exp.py:127-222 run as written -- same imports, keyword arguments and defaults -- with stand-ins only
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
    """moving_mnist_example.ipynb cells 0-7: imports (incl. the abstract NextFramePredictor of cell 2), the predictor built
    with the notebook's kwargs (input_timesteps left at its default 3 although the clips carry 4 input frames), test_threshold
    on a CPU clip, `model.train(loader, loader, lr=0.01, n_epochs=1)` under cProfile, a second predictor trained with
    `model.train(loader_train, loader_test, lr=0.01, n_epochs=...)` -- mask=None, no climatology, truncated_backprop=45 > T_out
    -- `model.loss.plot()` and `model.predict(loader_val)`."""
    import matplotlib
    matplotlib.use('Agg')
    monkeypatch.chdir(tmp_path)                      # (a SummaryWriter, when tensorboard is installed, writes runs/ here)
    # ---- cell 0
    import matplotlib.pyplot as plt
    import random
    from model.utils import normalize  # noqa: F401
    from torch.utils.data import DataLoader
    ModMovingMNISTDataset = TinyMovingMNISTDataset
    from model.mpnnlstm import NextFramePredictorS2S
    from model.model import MPNNLSTM, MPNNLSTMI  # noqa: F401
    # ---- cell 1
    np.random.seed(1)
    random.seed(1)
    torch.manual_seed(1)
    input_features = 1  # noqa: F841
    input_timesteps = 4
    output_timesteps = 10
    mnist_kwargs = dict(
        input_timesteps=input_timesteps,
        output_timesteps=output_timesteps,
        n_digits=1,
        gap=0,
        canvas_size=(32, 32),
        digit_size=(18, 18),
        pixel_noise=0.05,
        velocity_noise=0.0
    )
    data_train = ModMovingMNISTDataset(6, **mnist_kwargs)                     # (200 / 50 / 50 in the notebook)
    data_test = ModMovingMNISTDataset(3, seed=1, **mnist_kwargs)
    data_val = ModMovingMNISTDataset(3, seed=2, **mnist_kwargs)
    loader_train = DataLoader(data_train, batch_size=1, shuffle=True)
    loader_test = DataLoader(data_test, batch_size=1, shuffle=True)
    loader_val = DataLoader(data_val, batch_size=1, shuffle=False)
    # ---- cell 2
    from model.mpnnlstm import NextFramePredictor
    from torch.optim.lr_scheduler import StepLR  # noqa: F401
    from model.graph_functions import image_to_graph, flatten, Graph, unflatten  # noqa: F401
    from model.utils import add_positional_encoding  # noqa: F401
    np.random.seed(1)
    random.seed(1)
    torch.manual_seed(1)
    device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    model_kwargs = dict(
        hidden_size=16,
        dropout=0.1,
        n_layers=2
    )
    model = NextFramePredictorS2S(
        thresh=0.1,
        experiment_name='test',
        decompose=True,
        input_features=1,
        device=device,
        output_timesteps=output_timesteps,
        remesh_input=False,
        model_kwargs=model_kwargs)
    assert model.get_n_params() == 34513                                        # SURVEY KAT-6
    assert isinstance(model, NextFramePredictor) and NextFramePredictor.__abstractmethods__ == {'train', 'predict', 'score'}
    with pytest.raises(TypeError):
        NextFramePredictor(thresh=0.1)                                         # abstract, like the reference's
    # ---- cell 3
    x, _, _ = next(iter(loader_val))
    x = x.squeeze(0)
    for th in (1.5, 0.85, 0.5, 0.15):
        fig, axs = model.test_threshold(x, thresh=th)
        assert len(axs) == input_timesteps
        plt.close(fig)
    # ---- cell 4
    import cProfile
    loader_profile = DataLoader(data_train, batch_size=1, sampler=torch.utils.data.SubsetRandomSampler(range(4)))
    cProfile.runctx('model.train(loader_profile, loader_profile, lr=0.01, n_epochs=1)', globals(), locals(), sort=1)
    assert len(model.loss) == 1 and np.isfinite(model.loss.values).all()
    # ---- cell 5
    model = NextFramePredictorS2S(
        thresh=0.1,
        experiment_name='test',
        decompose=True,
        input_features=1,
        device=device,
        output_timesteps=output_timesteps,
        remesh_input=False,
        model_kwargs=model_kwargs)
    before = {k: v.clone() for k, v in model.model.state_dict().items()}
    model.train(loader_train, loader_test, lr=0.01, n_epochs=3)                # (n_epochs=20 in the notebook)
    # ---- cell 6
    ax = model.loss.plot()
    plt.close(ax.figure)
    assert list(model.loss.columns) == ['train_loss', 'test_loss'] and len(model.loss) == 3
    assert np.isfinite(model.loss.values).all() and (model.loss.values < 4).all()
    assert model.loss.train_loss.iloc[-1] < model.loss.train_loss.iloc[0]
    assert any(not torch.equal(v, before[k]) for k, v in model.model.state_dict().items())
    # ---- cell 7
    y_hat = model.predict(loader_val)
    assert y_hat.shape == (3, output_timesteps, 32, 32, 1) and np.isfinite(y_hat).all()
    # ---- cell 8 reads these
    assert loader_val.dataset.x[0][0, ..., 0].shape == y_hat[0][0][..., 0].shape


@pytest.mark.parametrize('exp', [9, 10, 1, 0])
def test_ice_exp_sequence_runs_as_written(exp, tmp_path, monkeypatch, capsys):
    """ice_exp.py:47-90 (experiment switch) and :109-224: preset heterogeneous (exp 9) / homogeneous (exp 10) meshes with
    `max_grid_size=4, use_edge_attrs=True, resolution=1/6 | 1/12`, TransformerConv x hidden 32 x 3 conv layers, `debug=True`,
    `binary=`, train() at half resolution and again at full resolution with ONE model, loss.to_csv, save, eval, predict with
    the preset mesh; exp 1 = GCNConv on the pixelwise mesh, exp 0 = the defaults (TransformerConv, pixelwise)."""
    from model.utils import normalize, int_to_datetime  # noqa: F401
    from model.mpnnlstm import NextFramePredictorS2S
    from model.seq2seq import Seq2Seq  # noqa: F401
    from torch.utils.data import Dataset, DataLoader  # noqa: F401
    from model.graph_functions import create_static_heterogeneous_graph, create_static_homogeneous_graph
    import random
    monkeypatch.chdir(tmp_path)
    np.random.seed(21)
    random.seed(21)
    torch.manual_seed(21)
    device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    month = 6
    # Defaults
    convolution_type = 'TransformerConv'
    lr = 0.0001
    multires_training = False
    truncated_backprop = 0
    training_years = range(2007, 2013)
    x_vars = ['siconc', 't2m', 'v10', 'u10', 'sshf']
    input_features = len(x_vars)
    input_timesteps = 3                                                         # (10 / 90 in the script)
    output_timesteps = 4
    preset_mesh = False
    binary = False
    if exp == 1:
        convolution_type = 'GCNConv'
    elif exp == 9:
        multires_training = True
        preset_mesh = 'heterogeneous'
    elif exp == 10:
        multires_training = True
        preset_mesh = 'homogeneous'

    from qtmpnn import synthetic
    if multires_training:
        mask_half = synthetic.make_ice_like(31, shape=(24, 32), channels=1, n_frames=1)[1]
        data_train_half = TinyIceDataset(3, input_timesteps, output_timesteps, (24, 32), seed=10)
        data_test_half = TinyIceDataset(2, input_timesteps, output_timesteps, (24, 32), seed=11)
        loader_train_half = DataLoader(data_train_half, batch_size=1, shuffle=True)
        loader_test_half = DataLoader(data_test_half, batch_size=1, shuffle=True)
        if preset_mesh == 'heterogeneous':
            graph_structure_half = create_static_heterogeneous_graph(mask_half.shape, 4, mask_half, use_edge_attrs=True, resolution=1/6, device=device)
        elif preset_mesh == 'homogeneous':
            graph_structure_half = create_static_homogeneous_graph(mask_half.shape, 4, mask_half, use_edge_attrs=True, resolution=1/6, device=device)

    mask = synthetic.make_ice_like(32, shape=(48, 64), channels=1, n_frames=1)[1]
    high_interest_region = np.zeros_like(mask)
    high_interest_region[10:20, 30:44] = True
    image_shape = mask.shape
    graph_structure = None
    if preset_mesh == 'heterogeneous':
        graph_structure = create_static_heterogeneous_graph(image_shape, 4, mask, use_edge_attrs=True, resolution=1/12, device=device)
    elif preset_mesh == 'homogeneous':
        graph_structure = create_static_homogeneous_graph(image_shape, 4, mask, use_edge_attrs=True, resolution=1/12, device=device)

    data_train = TinyIceDataset(3, input_timesteps, output_timesteps, image_shape, seed=12)
    data_test = TinyIceDataset(2, input_timesteps, output_timesteps, image_shape, seed=13)
    data_val = TinyIceDataset(2, input_timesteps, output_timesteps, image_shape, seed=14, first_day=(2010, 12, 30))
    loader_train = DataLoader(data_train, batch_size=1, shuffle=True)
    loader_test = DataLoader(data_test, batch_size=1, shuffle=True)
    loader_val = DataLoader(data_val, batch_size=1, shuffle=False)
    base = synthetic.make_ice_like(33, shape=image_shape, channels=1, n_frames=1)[0][0, ..., 0]
    climatology = torch.tensor(np.nan_to_num(climatology_from_base(base))).to(device)

    thresh = -np.inf
    print(f'Threshold is {thresh}')

    def dist_from_05(arr):
        return abs(abs(arr - 0.5) - 0.5)

    model_kwargs = dict(
        hidden_size=32,
        dropout=0.1,
        n_layers=1,
        transform_func=dist_from_05,
        dummy=False,
        n_conv_layers=3,
        rnn_type='LSTM',
        convolution_type=convolution_type,
    )
    experiment_name = f'M{str(month)}_Y{training_years[0]}_Y{training_years[-1]}_I{input_timesteps}O{output_timesteps}'
    model = NextFramePredictorS2S(
        thresh=thresh,
        experiment_name=experiment_name,
        input_features=input_features,
        input_timesteps=input_timesteps,
        output_timesteps=output_timesteps,
        transform_func=dist_from_05,
        device=device,
        binary=binary,
        debug=True,
        model_kwargs=model_kwargs)
    print('Num. parameters:', model.get_n_params())
    print('Model:\n', model.model)
    model.model.train()

    if multires_training:
        model.train(
            loader_train_half,
            loader_test_half,
            lr=lr,
            n_epochs=2,
            mask=mask_half,
            truncated_backprop=truncated_backprop,
            graph_structure=graph_structure_half)
        assert len(model.loss) == 2
    opt = model.optimizer if multires_training else None

    model.train(
        loader_train,
        loader_test,
        climatology,
        lr=lr,
        n_epochs=2,
        mask=mask,
        high_interest_region=high_interest_region,
        truncated_backprop=truncated_backprop,
        graph_structure=graph_structure,
        )
    # the second train() call keeps the optimizer and appends to the loss history (mpnnlstm.py:203-205, 373-374)
    assert (opt is None or model.optimizer is opt) and len(model.loss) == (4 if multires_training else 2)
    assert np.isfinite(model.loss.values).all()

    results_dir = f'ice_results_jun3_{exp}_multires_noclim'
    if not os.path.exists(results_dir):
        os.makedirs(results_dir)
    model.loss.to_csv(f'{results_dir}/loss_{experiment_name}.csv')
    model.save(results_dir)
    assert os.path.exists(f'{results_dir}/{experiment_name}.pth')

    model.model.eval()
    val_preds = model.predict(
        loader_val,
        climatology,
        mask=mask,
        graph_structure=graph_structure
        )
    launch_dates = [int_to_datetime(t) for t in loader_val.dataset.launch_dates]
    assert len(launch_dates) == 2
    y_hat = val_preds.squeeze(-1)
    assert y_hat.shape == loader_val.dataset.y.squeeze(-1).shape == (2, output_timesteps, *image_shape)
    if preset_mesh == 'homogeneous':
        # a homogeneous cell keeps its masked pixels (graph_functions.py:707-737): only cells entirely under the mask are empty
        assert np.isfinite(y_hat[:, :, ~mask]).all()
    else:
        assert np.isfinite(y_hat[:, :, ~mask]).all()
        if preset_mesh is False:
            assert np.isnan(y_hat[:, :, mask]).all()                            # unflatten_pixelwise: NaN under the mask
    assert 'Num. parameters:' in capsys.readouterr().out


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
    assert float(loss) == float(chunk_losses[0])
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
