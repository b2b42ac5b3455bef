"""GPU parity of the whole rollout (Seq2Seq forward + masked MSE + backward) against the golden
traces captured from the reference, plus batching / determinism / training smoke checks."""
import numpy as np
import pytest
import torch

from helpers import close, dev, dist_from_05, golden, grad_close, load_state

pytestmark = pytest.mark.gpu


def _model_from_golden(g, x, conv='ChebConv'):
    from model.seq2seq import Seq2Seq
    model = Seq2Seq(hidden_size=int(g['hidden']), dropout=0.0, thresh=float(g['thresh']), input_timesteps=x.shape[0],
                    input_features=x.shape[-1] + 3, output_timesteps=g['y'].shape[0], n_layers=int(g['n_layers']),
                    n_conv_layers=int(g['n_conv']), transform_func=dist_from_05 if bool(g['has_transform']) else None,
                    convolution_type=conv)
    load_state(model, g, 'w/')
    return model.to(dev())


def _run(g, batch=1, conv='ChebConv'):
    from model.mpnnlstm import masked_mse
    x, y, concat = (torch.from_numpy(g[k]).to(dev()) for k in ('x', 'y', 'concat'))
    model = _model_from_golden(g, g['x'], conv)
    if batch > 1:
        x, y, concat = (t.unsqueeze(0).repeat(batch, *[1] * t.dim()) for t in (x, y, concat))
    hir = g['hir'] if 'hir' in g.files else None
    outs, meshes = model(x, y, concat, teacher_forcing_ratio=0, mask=g['mask'], high_interest_region=hir)
    loss = masked_mse(outs, meshes, y, g['mask'])
    return model, outs, meshes, loss


def _check_grads(model, g):
    """A parameter the HIP path never touches (conv_h of the encoder's upper layers, whose hidden input is
    identically zero) has grad None; the reference produces an exactly-zero gradient there."""
    for k, p in model.named_parameters():
        ref = g['g/' + k]
        if p.grad is None:
            assert not ref.any(), f'{k}: no gradient on the HIP path but the reference gradient is non-zero'
            continue
        grad_close(p.grad, ref, msg=k, floor=0.05 if k.endswith('lin_key.bias') else 1e-3)


def _assert_grads_do_not_alias(model):
    """clip_grad_norm_ scales .grad in place: two parameters must never share gradient memory."""
    spans = sorted((p.grad.data_ptr(), p.grad.data_ptr() + p.grad.numel() * 4, k)
                   for k, p in model.named_parameters() if p.grad is not None)
    for (a0, a1, ka), (b0, b1, kb) in zip(spans[:-1], spans[1:]):
        assert a1 <= b0, f'gradients of {ka} and {kb} overlap in memory'


def _check_trace(g, outs, meshes, clip=0):
    """Per-step index parity: labels must be bit-exact at every step of the trace (a mismatch fails the test)."""
    thresh = float(g['thresh'])
    for i, mesh in enumerate(meshes):
        off = mesh.node_off.cpu().numpy()
        lab = mesh.labels[clip].cpu().numpy()
        lab = np.where(lab >= 0, lab - off[clip], -1)
        ref = g[f'labels_{i}']
        if not np.array_equal(lab, ref):
            img = g[f'image_{i}']
            near = np.abs(img - thresh).min()
            # every committed trace keeps its pixels away from the threshold: a differing mesh is a regression (the steps
            # before it were asserted), never silently skipped
            pytest.fail(f'mesh {i} differs from the reference trace (nearest pixel to the threshold: {near:.2e})')
        o = outs[i][off[clip]:off[clip + 1]]
        close(o, g[f'out_{i}'], msg=f'output step {i}')


@pytest.mark.parametrize('name', ['mnist64_h16', 'mnist64_noise_h8', 'ice64_masked_h8', 'mnist64_l4_h8', 'cfg2_mnist64'])
def test_rollout_golden(name):
    g = golden(f'rollout_{name}.npz')
    model, outs, meshes, loss = _run(g)
    _check_trace(g, outs, meshes)
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss'])), (float(loss), float(g['loss']))
    loss.backward()
    _check_grads(model, g)


LARGE = ['ice96x128_masked_h8', 'ice128_h32']


@pytest.mark.parametrize('name', LARGE)
def test_rollout_large_frames_golden(name):
    """Re-meshing rollouts on frames of SEVERAL 64x64 base cells against the reference's traces (model/seq2seq.py:339-398,
    434-491; base-cell stack model/graph_functions.py:199-205): 96x128 with a land mask (2x2 base cells, lower row cropped;
    hidden 8, 2 layers, stacks of 2 ChebConvs) and the BASELINE configs[3] shape (128x128, 5 channels, transform_func,
    thresh 0.15, hidden 32, 1 layer, stacks of 3; ice_exp.py:145-162).  These frames take the per-hop / halo recurrences,
    the multi-tile state transfer (Mesh.cell_off), stage 3's cross-cell scan and the hidden-32 cell, none of which a 64x64
    trace reaches.  Per-step labels bit-exact, outputs, loss, all gradients -- single clip and as a 2-clip batch."""
    g = golden(f'rollout_{name}.npz')
    model, outs, meshes, loss = _run(g)
    _check_trace(g, outs, meshes)
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss'])), (float(loss), float(g['loss']))
    loss.backward()
    _check_grads(model, g)
    _assert_grads_do_not_alias(model)
    from qtmpnn.mesh import tile_error_word
    assert tile_error_word() == 0          # (no tile-resident launch gave up waiting for a neighbour tile)
    model, outs, meshes, loss = _run(g, batch=2)
    for c in range(2):
        _check_trace(g, outs, meshes, clip=c)
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    loss.backward()
    _check_grads(model, g)


@pytest.mark.parametrize('name', LARGE)
def test_rollout_large_frames_static_capacity_graph_step(name):
    """The same two traces through what the benchmarks time: static capacities (node counts read on the device) with
    forward + loss + backward captured into ONE hipGraph and replayed on a 2-clip batch; labels bit-exact at every step,
    outputs, loss and all gradients against the reference."""
    from model.mpnnlstm import masked_mse
    g = golden(f'rollout_{name}.npz')
    x, y, concat = (torch.from_numpy(g[k]).to(dev()).unsqueeze(0).repeat(2, *[1] * g[k].ndim) for k in ('x', 'y', 'concat'))
    model = _model_from_golden(g, g['x'])
    model.static_shapes = True
    # (ONE mask / region object for every call: the device copy is cached per object, an upload inside a capture is an error)
    mask, hir = g['mask'], (g['hir'] if 'hir' in g.files else None)
    params = list(model.parameters())
    n, m = g['x'].shape[1:3]

    def fwd_bwd(a, b, c):
        for p in params:
            p.grad = None
        outs, meshes = model(a, b, c, teacher_forcing_ratio=0, mask=mask, high_interest_region=hir)
        loss = masked_mse(outs, meshes, b, mask)
        loss.backward()
        return outs, meshes, loss.detach()
    sx, sy, sc = (torch.zeros_like(t) for t in (x, y, concat))          # captured on other data than it is replayed on
    sx.copy_(x.flip(0).roll(1, 1)); sy.copy_(y); sc.copy_(concat)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd_bwd(sx, sy, sc)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        outs, meshes, loss = fwd_bwd(sx, sy, sc)
    sx.copy_(x)
    graph.replay()
    torch.cuda.synchronize()
    assert all(ms.n_dev is not None and ms.N == 2 * n * m for ms in meshes)
    for i, ms in enumerate(meshes):
        off = ms.node_off.cpu().numpy()
        for c in range(2):
            lab = ms.labels[c].cpu().numpy()
            assert np.array_equal(np.where(lab >= 0, lab - off[c], -1), g[f'labels_{i}']), f'mesh {i} clip {c}'
            close(outs[i][off[c]:off[c + 1]], g[f'out_{i}'], msg=f'output step {i} clip {c}')
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss'])), (float(loss), float(g['loss']))
    _check_grads(model, g)


def test_rollout_gcnconv_golden():
    """convolution_type='GCNConv' (model/model.py:41,50) through the whole rollout with re-meshing against the reference's
    trace (two layers, stacks of two GCNConvs composed in weight space as the Chebyshev series [0, -W^T]): per-step labels
    bit-exact, outputs, loss and all gradients; the same as a 2-clip batch."""
    g = golden('rollout_gcn_mnist64_h8.npz')
    model, outs, meshes, loss = _run(g, conv='GCNConv')
    assert set(k for k, _ in model.named_parameters()) == set(k[2:] for k in g.files if k.startswith('w/'))
    _check_trace(g, outs, meshes)
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss'])), (float(loss), float(g['loss']))
    loss.backward()
    _check_grads(model, g)
    model, outs, meshes, loss = _run(g, batch=2, conv='GCNConv')
    for c in range(2):
        _check_trace(g, outs, meshes, clip=c)
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))


def test_rollout_batched_equals_single():
    """B identical clips: per-clip outputs equal the single-clip outputs; loss equal; grads equal (mean over clips)."""
    g = golden('rollout_ice64_masked_h8.npz')
    model, outs, meshes, loss = _run(g, batch=3)
    for c in range(3):
        _check_trace(g, outs, meshes, clip=c)
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    loss.backward()
    _check_grads(model, g)


def test_rollout_deterministic():
    g = golden('rollout_mnist64_h16.npz')
    res = []
    for _ in range(2):
        model, outs, meshes, loss = _run(g)
        loss.backward()
        res.append((loss.detach().clone(), [p.grad.clone() for p in model.parameters() if p.grad is not None]))
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)


def test_train_steps_reduce_loss():
    from model.mpnnlstm import NextFramePredictorS2S
    from qtmpnn import synthetic
    torch.manual_seed(1)
    x, y = synthetic.make_batch(1, 0, 4, 5, 5, n_digits=1, pixel_noise=0.05)
    x, y = torch.from_numpy(x).to(dev()), torch.from_numpy(y).to(dev())
    mask = np.zeros((64, 64), dtype=bool)
    concat = torch.zeros(4, 5, 64, 64, 1, device=dev())
    nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=5, output_timesteps=5, device=dev(),
                                model_kwargs=dict(hidden_size=16, dropout=0.1, n_layers=2))
    assert nfp.get_n_params() == 34513
    nfp.initiate_training(lr=0.01, lr_decay=0.95)
    losses = [float(nfp.train_step(x, y, concat, mask)) for _ in range(8)]
    assert np.isfinite(losses).all()
    assert losses[-1] < losses[0], losses


def test_rollout_static_capacity_mode_equals_golden():
    """Static mode (worst-case buffers, node counts read on the device, no host sync): same results."""
    g = golden('rollout_ice64_masked_h8.npz')
    from model.mpnnlstm import masked_mse
    x, y, concat = (torch.from_numpy(g[k]).to(dev()) for k in ('x', 'y', 'concat'))
    model = _model_from_golden(g, g['x'])
    model.static_shapes = True
    outs, meshes = model(x, y, concat, teacher_forcing_ratio=0, mask=g['mask'], high_interest_region=g['hir'] if 'hir' in g.files else None)
    assert all(ms.n_dev is not None and ms.N == 64 * 64 for ms in meshes)
    for i, ms in enumerate(meshes):
        nv = ms.n_valid
        assert nv == len(g[f'out_{i}'])
        assert np.array_equal(ms.labels[0].cpu().numpy(), g[f'labels_{i}'])
        close(outs[i][:nv], g[f'out_{i}'], msg=f'output step {i}')
    loss = masked_mse(outs, meshes, y, g['mask'])
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    loss.backward()
    _check_grads(model, g)


def test_graphed_step_matches_eager_steps():
    """A hipGraph-captured training step (fwd + loss + bwd + clip + Adam in ONE graph launch) replays to the same
    losses and weights as eager steps from the same state."""
    from model.mpnnlstm import NextFramePredictorS2S
    from qtmpnn import synthetic
    x, y = synthetic.make_batch(1, 0, 3, 4, 4, n_digits=1, pixel_noise=0.05)
    x2, y2 = synthetic.make_batch(1, 50, 3, 4, 4, n_digits=1, pixel_noise=0.05)
    t = lambda a: torch.from_numpy(a).to(dev())
    mask = np.zeros((64, 64), dtype=bool)
    concat = torch.zeros(3, 4, 64, 64, 1, device=dev())

    def fresh():
        torch.manual_seed(3)
        nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=4, output_timesteps=4, device=dev(),
                                    model_kwargs=dict(hidden_size=16, dropout=0.0, n_layers=2))
        nfp.initiate_training(lr=1e-3, lr_decay=0.95, capturable=True)
        nfp.model.static_shapes = True
        return nfp
    eager, graphed = fresh(), fresh()
    for _ in range(2):
        eager.train_step(t(x), t(y), concat, mask)
    step = graphed.make_graphed_step(t(x), t(y), concat, mask, warmup=2)     # two eager warm-up steps on (x, y)
    for a, b in ((x2, y2), (x, y), (x2, y2)):
        le, lg = float(eager.train_step(t(a), t(b), concat, mask)), float(step(t(a), t(b), concat))
        assert abs(le - lg) <= 1e-6 * abs(le), (le, lg)
    for (k, p), (_, q) in zip(eager.model.named_parameters(), graphed.model.named_parameters()):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=1e-6, atol=1e-7, err_msg=k)


def test_graphed_transformerconv_step_matches_eager_steps():
    """a training step with convolution_type='TransformerConv' captures into a hipGraph (edge attributes
    built once per mesh on the device, weights packed once per pass) and replays to the eager losses; with attention
    dropout on, replays draw new masks (the losses of two replays on the same batch differ), eager or graphed."""
    from model.mpnnlstm import NextFramePredictorS2S
    from qtmpnn import synthetic
    x, y = synthetic.make_batch(1, 0, 2, 3, 3, n_digits=1, pixel_noise=0.02)
    x2, y2 = synthetic.make_batch(1, 50, 2, 3, 3, n_digits=1, pixel_noise=0.02)
    t = lambda a: torch.from_numpy(a).to(dev())
    mask = np.zeros((64, 64), dtype=bool)
    concat = torch.zeros(2, 3, 64, 64, 1, device=dev())

    def fresh(lr):
        torch.manual_seed(5)
        nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=3, output_timesteps=3, device=dev(),
                                    model_kwargs=dict(hidden_size=8, dropout=0.0, n_layers=1, n_conv_layers=2,
                                                      convolution_type='TransformerConv'))
        nfp.initiate_training(lr=lr, lr_decay=0.95, capturable=True)
        nfp.model.static_shapes = True
        return nfp
    from model.model import CONVOLUTION_KWARGS
    old = dict(CONVOLUTION_KWARGS['TransformerConv'])
    try:
        CONVOLUTION_KWARGS['TransformerConv']['dropout'] = 0.0
        eager, graphed = fresh(1e-3), fresh(1e-3)
        for _ in range(2):
            eager.train_step(t(x), t(y), concat, mask)
        step = graphed.make_graphed_step(t(x), t(y), concat, mask, warmup=2)
        for a, b in ((x2, y2), (x, y)):
            le, lg = float(eager.train_step(t(a), t(b), concat, mask)), float(step(t(a), t(b), concat))
            assert np.isfinite(le) and abs(le - lg) <= 1e-5 * abs(le), (le, lg)
        # attention dropout: a frozen model (lr = 0) replayed twice on the same batch sees two different masks
        CONVOLUTION_KWARGS['TransformerConv']['dropout'] = 0.3
        drop = fresh(0.0)
        step = drop.make_graphed_step(t(x), t(y), concat, mask, warmup=2)
        l1, l2 = float(step(t(x), t(y), concat)), float(step(t(x), t(y), concat))
        assert np.isfinite(l1) and np.isfinite(l2) and l1 != l2, (l1, l2)
    finally:
        CONVOLUTION_KWARGS['TransformerConv'].update(old)


def test_graph_replay_gradients_bit_identical_to_eager():
    """Forward + loss + backward captured in a hipGraph and replayed on NEW inputs gives bit-identical loss and
    gradients to the eager launches (same weights): the capture contains every kernel of the path."""
    from model.mpnnlstm import NextFramePredictorS2S
    from qtmpnn import synthetic
    x, y = synthetic.make_batch(1, 0, 3, 4, 4, n_digits=1, pixel_noise=0.05)
    x2, y2 = synthetic.make_batch(1, 50, 3, 4, 4, n_digits=1, pixel_noise=0.05)
    t = lambda a: torch.from_numpy(a).to(dev())
    mask = np.zeros((64, 64), dtype=bool)
    concat = torch.zeros(3, 4, 64, 64, 1, device=dev())
    torch.manual_seed(3)
    nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=4, output_timesteps=4, device=dev(),
                                model_kwargs=dict(hidden_size=16, dropout=0.0, n_layers=2))
    nfp.model.static_shapes = True
    params = list(nfp.model.parameters())

    def fwd_bwd(a, b):
        for p in params:
            p.grad = None
        loss = nfp.forward_loss(a, b, concat, mask)
        loss.backward()
        return loss.detach()
    ref_loss = fwd_bwd(t(x2), t(y2)).clone()
    ref = [None if p.grad is None else p.grad.clone() for p in params]
    sx, sy = t(x).clone(), t(y).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd_bwd(sx, sy)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    for p in params:
        p.grad = None
    with torch.cuda.graph(graph, stream=side):
        loss = fwd_bwd(sx, sy)
    for _ in range(2):
        sx.copy_(t(x2))
        sy.copy_(t(y2))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(loss, ref_loss)
        for p, r in zip(params, ref):
            assert (p.grad is None) == (r is None)
            if r is not None:
                assert torch.equal(p.grad, r)


def _run_fixed(g, gs_builder=None, batch=1):
    from model.mpnnlstm import masked_mse
    from model.seq2seq import Seq2Seq
    x, y, concat = (torch.from_numpy(g[k]).to(dev()) for k in ('x', 'y', 'concat'))
    model = Seq2Seq(hidden_size=int(g['hidden']), dropout=0.0, thresh=-np.inf, input_timesteps=g['x'].shape[0],
                    input_features=g['x'].shape[-1] + 3, output_timesteps=g['y'].shape[0], n_layers=int(g['n_layers']),
                    n_conv_layers=int(g['n_conv']), convolution_type='ChebConv')
    load_state(model, g, 'w/')
    model.to(dev())
    if batch > 1:
        x, y, concat = (t.unsqueeze(0).repeat(batch, *[1] * t.dim()) for t in (x, y, concat))
    gs = gs_builder() if gs_builder else None
    outs, meshes = model(x, y, concat, teacher_forcing_ratio=0, mask=g['mask'], graph_structure=gs)
    return model, outs, meshes, masked_mse(outs, meshes, y, g['mask']), gs


@pytest.mark.parametrize('batch', [1, 2])
def test_pixelwise_rollout_golden(batch):
    """thresh = -inf (one node per unmasked pixel, no re-mesh) against the reference trace."""
    from model.graph_functions import unflatten
    g = golden('fixed_pixelwise48x64.npz')
    model, outs, meshes, loss, _ = _run_fixed(g, batch=batch)
    n1 = meshes[0].N // batch
    for b in range(batch):
        for i, o in enumerate(outs):
            close(o[b * n1:(b + 1) * n1], g[f'out_{i}'], msg=f'clip {b} step {i}')
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    if batch == 1:
        img = unflatten(outs[0], meshes[0], (48, 64), g['mask'])             # NaN under the mask like unflatten_pixelwise
        ref = g['y_hat'][0]
        assert np.array_equal(np.isnan(img.detach().cpu().numpy()), np.isnan(ref))
        close(torch.nan_to_num(img), np.nan_to_num(ref))
    loss.backward()
    _check_grads(model, g)


def test_static_mesh_rollout_golden():
    """preset heterogeneous mesh (create_static_heterogeneous_graph) + thresh = -inf."""
    from model.graph_functions import create_static_heterogeneous_graph
    g = golden('fixed_static48x64.npz')
    build = lambda: create_static_heterogeneous_graph((48, 64), int(g['max_grid_size']), g['mask'], high_interest_region=g['hir'],
                                                      use_edge_attrs=False, device=dev())
    model, outs, meshes, loss, gs = _run_fixed(g, build)
    mesh = gs['mapping']
    assert np.array_equal(mesh.labels[0].cpu().numpy(), g['static_labels'])
    assert np.array_equal(gs['n_pixels_per_node'].cpu().numpy(), g['static_npix'])
    assert np.array_equal(gs['edge_index'].cpu().numpy(), g['static_edges'])
    close(gs['edge_attrs'], g['static_dist'], atol=2e-5)
    for i, o in enumerate(outs):
        close(o, g[f'out_{i}'], msg=f'step {i}')
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    loss.backward()
    _check_grads(model, g)
    # the same preset mesh drives a 3-clip batch
    model2, outs2, meshes2, loss2, _ = _run_fixed(g, build, batch=3)
    assert meshes2[0].B == 3 and meshes2[0].N == 3 * mesh.N
    assert abs(float(loss2) - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))


def test_homogeneous_mesh_rollout_golden():
    """uniform preset mesh with fully masked cells removed (create_static_homogeneous_graph); partly
    masked cells keep all their pixels, so the loss masks per pixel -- against the reference trace."""
    from model.graph_functions import create_static_homogeneous_graph
    g = golden('fixed_homog48x64.npz')
    build = lambda: create_static_homogeneous_graph((48, 64), int(g['max_grid_size']), g['mask'], use_edge_attrs=False, device=dev())
    model, outs, meshes, loss, gs = _run_fixed(g, build)
    mesh = gs['mapping']
    assert np.array_equal(mesh.labels[0].cpu().numpy(), g['static_labels'])
    assert np.array_equal(gs['n_pixels_per_node'].cpu().numpy(), g['static_npix'])
    assert np.array_equal(gs['edge_index'].cpu().numpy(), g['static_edges'])
    close(gs['edge_attrs'], g['static_dist'], atol=2e-5)
    for i, o in enumerate(outs):
        close(o, g[f'out_{i}'], msg=f'step {i}')
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    loss.backward()
    _check_grads(model, g)
    model2, outs2, meshes2, loss2, _ = _run_fixed(g, build, batch=2)
    assert meshes2[0].B == 2 and meshes2[0].N == 2 * mesh.N
    assert abs(float(loss2) - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))


@pytest.mark.parametrize('cfg', ['cfg3_mnist128', 'cfg4_ice128', 'cfg4_ice128_transformer', 'cfg5_ice256'])
def test_baseline_config_shapes_train(cfg):
    """BASELINE.json configs[2..4] at their full image sizes AND their full per-GPU batches -- configs[2]: 128x128, in=10/out=20,
    64 clips over 8 GPUs = 8 per GPU; configs[3]: ice-like 128x128, 5 channels, in=12/out=6, 16 clips on one GPU (ChebConv, and
    the TransformerConv stacks ice_exp.py:48 hard-codes); configs[4]: ice-like 256x256, in=12/out=12, 32 clips over 8 GPUs = 4
    per GPU, quadtree rebuilt at every step (ice_exp_nwt.py:46,80,89-96).  Eager training steps must give finite, falling
    losses, gradients for every used parameter and meshes that satisfy the size-independent invariants; the hipGraph-replayed
    step (static capacities; what tools/bench_configs.py times) must reproduce the eager losses from the same state."""
    import gc
    from model.mpnnlstm import NextFramePredictorS2S
    from qtmpnn import synthetic
    if cfg == 'cfg3_mnist128':
        B, t_in, t_out, shape, kw = 8, 10, 20, (128, 128), dict(hidden_size=16, dropout=0.0, n_layers=2)
        x, y = synthetic.make_batch(3, 0, B, t_in, t_out, n_digits=2, pixel_noise=0.05, canvas=shape)
        mask, thresh, tf, feat = np.zeros(shape, dtype=bool), 0.1, None, 1
    else:
        n = 128 if cfg.startswith('cfg4_ice128') else 256
        B, t_in, t_out, shape = (16, 12, 6, (n, n)) if n == 128 else (4, 12, 12, (n, n))
        kw = dict(hidden_size=32, dropout=0.0, n_layers=1, n_conv_layers=3)
        if cfg.endswith('transformer'):
            kw['convolution_type'] = 'TransformerConv'
        clips = [synthetic.make_ice_like(40 + i, shape=shape, channels=5, n_frames=t_in + t_out) for i in range(B)]
        x = np.stack([c[0][:t_in] for c in clips])
        y = np.stack([c[0][t_in:, ..., :1] for c in clips])
        mask, thresh, feat = clips[0][1], 0.15, 5
        tf = lambda a: abs(abs(a - 0.5) - 0.5)
        kw['transform_func'] = tf          # (the model takes it from model_kwargs, ice_exp.py:157)
    attention = cfg.endswith('transformer')
    # (attention dropout, p = 0.1, is part of the convolution's kwargs and its masks are seeded by a per-process launch counter: with it
    # on, eager and captured steps draw different masks and their losses agree only statistically -- round 4 held them to 20 %, which
    # depended on how many attention launches earlier tests had issued.  It is switched off for this comparison; replays drawing new
    # masks is test_graphed_transformerconv_step_matches_eager_steps' subject)
    from model.model import CONVOLUTION_KWARGS
    conv_kw = dict(CONVOLUTION_KWARGS['TransformerConv'])
    CONVOLUTION_KWARGS['TransformerConv']['dropout'] = 0.0
    try:
        _baseline_config_body(cfg, attention, B, t_in, t_out, shape, kw, x, y, mask, thresh, tf, feat, gc)
    finally:
        CONVOLUTION_KWARGS['TransformerConv'].update(conv_kw)


def _baseline_config_body(cfg, attention, B, t_in, t_out, shape, kw, x, y, mask, thresh, tf, feat, gc):
    from model.mpnnlstm import NextFramePredictorS2S
    xt, yt = torch.from_numpy(x).to(dev()), torch.from_numpy(y).to(dev())
    concat = torch.zeros(B, t_out, *shape, 1, device=dev())
    P_valid = int((~mask).sum())

    def fresh(capturable):
        torch.manual_seed(0)
        nfp = NextFramePredictorS2S(thresh=thresh, input_features=feat, input_timesteps=t_in, output_timesteps=t_out, device=dev(),
                                    transform_func=tf, model_kwargs=kw)
        nfp.initiate_training(lr=1e-3, lr_decay=0.95, capturable=capturable)
        nfp.model.train()
        return nfp

    # (both predictors train with static capacities and the capturable optimizer, like the captured step's own warm-up steps:
    # a rollout that re-meshes on its own output amplifies a last-bit difference between two optimizer variants within a few steps)
    nfp = fresh(True)
    outs, meshes = nfp.model(xt, yt, concat, teacher_forcing_ratio=0, mask=mask)          # exact node counts: invariants
    for ms in meshes:
        assert float(ms.npix.sum()) == B * P_valid
        lab = ms.labels
        assert int(lab.max()) == ms.N - 1 and bool(((lab < 0) == torch.from_numpy(mask).to(dev())).all())
    del outs, meshes
    nfp.model.static_shapes = True
    le = [float(nfp.train_step(xt, yt, concat, mask)) for _ in range(4)]
    assert np.isfinite(le).all() and le[-1] < le[0], le
    missing = [k for k, p in nfp.model.named_parameters() if p.grad is None and 'rnns.1.conv_h' not in k]
    assert not missing, missing[:5]
    assert all(torch.isfinite(p.grad).all() for p in nfp.model.parameters() if p.grad is not None)
    del nfp
    gc.collect()
    torch.cuda.empty_cache()
    # the same state through the captured step: 2 eager warm-up steps (static capacities), then replays = eager steps 3, 4
    graphed = fresh(True)
    step = graphed.make_graphed_step(xt, yt, concat, mask=mask, warmup=2)
    lg = [float(step(xt, yt, concat)) for _ in range(2)]
    assert np.isfinite(lg).all(), lg
    for a, b in zip(le[2:], lg):
        assert abs(a - b) <= 1e-4 * abs(a), (le, lg)
    del graphed, step
    gc.collect()
    torch.cuda.empty_cache()


def _variant_model(g, binary=False):
    from model.seq2seq import Seq2Seq
    model = Seq2Seq(hidden_size=8, dropout=0.0, thresh=0.1, input_timesteps=3, input_features=4, output_timesteps=4,
                    n_layers=1, n_conv_layers=2, convolution_type='ChebConv', binary=binary)
    load_state(model, g, 'w/')
    return model.to(dev())


def test_teacher_forcing_golden():
    """teacher_forcing_ratio = 1 -- next mesh and input come from the ground-truth frame."""
    from model.mpnnlstm import masked_mse
    g = golden('variant_teacher.npz')
    model = _variant_model(g)
    x, y, concat = (torch.from_numpy(g[k]).to(dev()) for k in ('x', 'y', 'concat'))
    outs, meshes = model(x, y, concat, teacher_forcing_ratio=1.0, mask=g['mask'])
    for i, o in enumerate(outs):
        assert o.shape[0] == g[f'out_{i}'].shape[0], f'mesh size of step {i}'
        close(o, g[f'out_{i}'], msg=f'step {i}')
    loss = masked_mse(outs, meshes, y, g['mask'])
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    loss.backward()
    _check_grads(model, g)


def test_binary_head_golden():
    """binary=True -- sigmoid on the decoder output, BCE loss."""
    from model.mpnnlstm import masked_mse
    g = golden('variant_binary.npz')
    model = _variant_model(g, binary=True)
    x, y, concat = (torch.from_numpy(g[k]).to(dev()) for k in ('x', 'y', 'concat'))
    outs, meshes = model(x, y, concat, teacher_forcing_ratio=0, mask=g['mask'])
    for i, o in enumerate(outs):
        close(o, g[f'out_{i}'], msg=f'step {i}')
    loss = masked_mse(outs, meshes, y, g['mask'], binary=True)
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    loss.backward()
    _check_grads(model, g)


def test_remesh_input_golden():
    """remesh_input=True -- every encoder step runs on the mesh of its own input frame and the state moves
    to the next frame's mesh after each step (x carries input_timesteps + 1 frames, as the reference's indexing demands)."""
    from model.mpnnlstm import masked_mse
    from model.seq2seq import Seq2Seq
    g = golden('variant_remesh_input.npz')
    model = Seq2Seq(hidden_size=8, dropout=0.0, thresh=0.1, input_timesteps=3, input_features=4, output_timesteps=4,
                    n_layers=1, n_conv_layers=2, convolution_type='ChebConv', remesh_input=True)
    load_state(model, g, 'w/')
    model.to(dev())
    x, y, concat = (torch.from_numpy(g[k]).to(dev()) for k in ('x', 'y', 'concat'))
    assert x.shape[0] == 4
    outs, meshes = model(x, y, concat, teacher_forcing_ratio=0, mask=g['mask'])
    for i, o in enumerate(outs):
        assert o.shape[0] == g[f'out_{i}'].shape[0], f'mesh size of step {i}'
        close(o, g[f'out_{i}'], msg=f'step {i}')
    loss = masked_mse(outs, meshes, y, g['mask'])
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    loss.backward()
    _check_grads(model, g)
    with pytest.raises(IndexError):          # the reference reads x[[t + 1]] after the last encoder step
        model(x[:3], y, concat, teacher_forcing_ratio=0, mask=g['mask'])


def test_truncated_bptt_golden():
    """the reference's truncated-BPTT chunk loop with truncated_backprop = 2."""
    from model.mpnnlstm import NextFramePredictorS2S
    g = golden('variant_tbptt.npz')
    nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=3, output_timesteps=4, device=dev(),
                                model_kwargs=dict(hidden_size=8, dropout=0.0, n_layers=1, n_conv_layers=2))
    load_state(nfp.model, g, 'w/')
    nfp.initiate_training(lr=1e-3, lr_decay=0.95)
    x, y, concat = (torch.from_numpy(g[k]).to(dev()) for k in ('x', 'y', 'concat'))
    losses = nfp.truncated_backward(x, y, concat, g['mask'], truncated_backprop=2)
    np.testing.assert_allclose([float(l) for l in losses], g['chunk_losses'], rtol=1e-4)
    _check_grads(nfp.model, g)          # the surviving gradient is the LAST chunk's


def test_transformer_rollout_golden():
    """full masked rollout with convolution_type='TransformerConv' (the ice scripts' setting)."""
    from model.mpnnlstm import masked_mse
    from model.seq2seq import Seq2Seq
    g = golden('transformer_rollout.npz')
    model = Seq2Seq(hidden_size=8, dropout=0.0, thresh=0.15, input_timesteps=2, input_features=6, output_timesteps=3,
                    n_layers=1, n_conv_layers=2, transform_func=dist_from_05, convolution_type='TransformerConv')
    load_state(model, g, 'w/')
    model.to(dev()).eval()
    x, y, concat = (torch.from_numpy(g[k]).to(dev()) for k in ('x', 'y', 'concat'))
    outs, meshes = model(x, y, concat, teacher_forcing_ratio=0, mask=g['mask'])
    for i, o in enumerate(outs):
        assert o.shape[0] == g[f'out_{i}'].shape[0], f'mesh size of step {i}'
        close(o, g[f'out_{i}'], msg=f'step {i}')
    loss = masked_mse(outs, meshes, y, g['mask'])
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    loss.backward()
    _check_grads(model, g)


def test_transformer_rollout_batched_equals_single_and_is_deterministic():
    """TransformerConv stacks (the layer-by-layer multi-head launches): a batch of 3 identical clips reproduces the single-clip
    reference trace clip by clip (the batched mesh is block diagonal: no head, group or clip may leak into another), with the
    same loss and gradients; two runs give bit-identical losses and gradients (no atomics anywhere in the attention backward)."""
    from model.mpnnlstm import masked_mse
    from model.seq2seq import Seq2Seq
    g = golden('transformer_rollout.npz')

    def run(batch):
        model = Seq2Seq(hidden_size=8, dropout=0.0, thresh=0.15, input_timesteps=2, input_features=6, output_timesteps=3,
                        n_layers=1, n_conv_layers=2, transform_func=dist_from_05, convolution_type='TransformerConv')
        load_state(model, g, 'w/')
        model.to(dev()).eval()
        x, y, concat = (torch.from_numpy(np.stack([g[k]] * batch)).to(dev()) for k in ('x', 'y', 'concat'))
        outs, meshes = model(x, y, concat, teacher_forcing_ratio=0, mask=g['mask'])
        loss = masked_mse(outs, meshes, y, g['mask'])
        loss.backward()
        return model, outs, meshes, loss

    model, outs, meshes, loss = run(3)
    for i, (o, ms) in enumerate(zip(outs, meshes)):
        off = ms.node_off.cpu().numpy()
        for c in range(3):
            assert off[c + 1] - off[c] == g[f'out_{i}'].shape[0], f'mesh size of step {i}, clip {c}'
            close(o[off[c]:off[c + 1]], g[f'out_{i}'], msg=f'step {i} clip {c}')
    assert abs(float(loss.detach()) - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    _check_grads(model, g)
    model2, _, _, loss2 = run(3)
    assert torch.equal(loss.detach(), loss2.detach())
    for a, b in zip(model.parameters(), model2.parameters()):
        assert (a.grad is None) == (b.grad is None) and (a.grad is None or torch.equal(a.grad, b.grad))


@pytest.mark.parametrize('tb', [0, 45])
def test_train_loop_with_graph_replay_matches_eager_loop(tb):
    """NextFramePredictorS2S.train(use_graph=True): the reference's epoch loop with the step replayed as a hipGraph gives the
    eager loop's losses (one update per batch, learning-rate schedule included) on a tiny in-memory loader.  tb = 45: the
    default truncation length, longer than the rollout -- the eager loop takes the truncated branch (one chunk, no clipping,
    mpnnlstm.py:281-315) and the graph captures exactly that."""
    from model.mpnnlstm import NextFramePredictorS2S
    from qtmpnn import synthetic

    class DS:
        image_shape = (64, 64)

    class Loader(list):
        dataset = DS()

    x, y = synthetic.make_batch(5, 0, 4, 3, 2, n_digits=1, pixel_noise=0.0)
    items = [(torch.from_numpy(x[i:i + 2]), torch.from_numpy(y[i:i + 2]), torch.zeros(1)) for i in (0, 2)]
    mask = np.zeros((64, 64), dtype=bool)

    def run(use_graph):
        torch.manual_seed(4)
        nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=3, output_timesteps=2, device=dev(),
                                    model_kwargs=dict(hidden_size=8, dropout=0.0, n_layers=1))
        nfp.train(Loader(items), Loader(items[:1]), n_epochs=4, lr=0.01, lr_decay=0.5, mask=mask, truncated_backprop=tb,
                  use_graph=use_graph)
        return nfp.train_loss, nfp.test_loss

    (tr_g, te_g), (tr_e, te_e) = run(True), run(False)
    # (capacity-sized launches sum some reductions in another order than exact-size ones: 1e-4-level drift after 8 updates)
    for a, b in zip(tr_g + te_g, tr_e + te_e):
        assert abs(a - b) <= 1e-3 * abs(b) + 1e-7, (tr_g, tr_e, te_g, te_e)
    if tb:          # a truncation shorter than the rollout re-runs the encoder per chunk: not one capturable step
        nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=3, output_timesteps=2, device=dev(),
                                    model_kwargs=dict(hidden_size=8, dropout=0.0, n_layers=1))
        with pytest.raises(ValueError, match='use_graph'):
            nfp.train(Loader(items), Loader(items[:1]), n_epochs=1, mask=mask, truncated_backprop=1, use_graph=True)


def test_graphed_step_odd_shape_with_mask():
    """Static capacities + hipGraph on a frame that is neither square nor a multiple of the 64-pixel base cell, with a land
    mask and 3 clips: the replayed step gives the eager step's loss and weights (exercises the fused mesh-build scans, the
    dense multi-part re-mesh transfer, the fused cell backward and the rollout-wide loss away from the bench shape)."""
    from model.mpnnlstm import NextFramePredictorS2S
    rng = np.random.default_rng(11)
    n, m, B, T = 40, 72, 3, 3
    mask = np.zeros((n, m), dtype=bool)
    mask[5:14, 50:66] = True
    mask[30:, :9] = True

    def clips(seed):
        r = np.random.default_rng(seed)
        f = r.random((B, 2 * T, n, m, 1)).astype(np.float32)
        f = (f > 0.93).astype(np.float32) * r.random((B, 2 * T, n, m, 1)).astype(np.float32)     # sparse blobs: mixed cell sizes
        return f[:, :T], f[:, T:]
    (x, y), (x2, y2) = clips(1), clips(2)
    t = lambda a: torch.from_numpy(a).to(dev())
    concat = torch.from_numpy(rng.random((B, T, n, m, 1)).astype(np.float32) * 0.1).to(dev())

    def fresh():
        torch.manual_seed(7)
        nfp = NextFramePredictorS2S(thresh=0.2, input_features=1, input_timesteps=T, output_timesteps=T, device=dev(),
                                    model_kwargs=dict(hidden_size=16, dropout=0.0, n_layers=2))
        nfp.initiate_training(lr=1e-3, lr_decay=0.95, capturable=True)
        nfp.model.static_shapes = True
        return nfp
    eager, graphed = fresh(), fresh()
    for _ in range(2):
        eager.train_step(t(x), t(y), concat, mask)
    step = graphed.make_graphed_step(t(x), t(y), concat, mask, warmup=2)
    for a, b in ((x2, y2), (x, y)):
        le, lg = float(eager.train_step(t(a), t(b), concat, mask)), float(step(t(a), t(b), concat))
        assert np.isfinite(le) and abs(le - lg) <= 1e-6 * abs(le), (le, lg)
    for (k, p), (_, q) in zip(eager.model.named_parameters(), graphed.model.named_parameters()):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=1e-6, atol=1e-7, err_msg=k)
