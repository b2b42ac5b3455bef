#!/usr/bin/env python3
"""Generate tests/golden/*.npz by EXECUTING the reference's own modules.

Run in the build container only (needs /root/reference; the GPU box has neither
the reference nor any need for this script):

    python tests/golden/make_golden.py

The reference's model/{graph_functions,utils,model,seq2seq}.py are imported
UNMODIFIED from /root/reference.  Third-party modules that are not installed
here get small stand-ins:
  numba.jit -> identity;  torch_geometric.data.Data -> attribute bag;
  torch_geometric.nn.{ChebConv,GCNConv} -> oracle/qt_oracle.py's restatement of
  the published PyG 2.2.0 definitions (so conv ARITHMETIC is "parity unpinned";
  everything around it -- quadtree, adjacency, flatten/unflatten, GConvLSTM,
  Encoder, Decoder, Seq2Seq control flow -- is the reference's own code);
  tensorboard / torchviz -> no-ops.
Only arrays (inputs and expected outputs) are written; no reference source.
"""
import os
import sys
import types
import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import qt_oracle as O          # noqa: E402
import importlib.util                      # noqa: E402
# the build's own package dir must NOT be on sys.path here: it holds a regular package called `model`, which would
# shadow the reference's `model/` (a namespace package: no __init__.py) whatever the path order
_spec = importlib.util.spec_from_file_location('qt_synthetic', os.path.join(ROOT, 'quadtree-mpnnlstm_amd', 'qtmpnn', 'synthetic.py'))
synthetic = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synthetic)


def install_standins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    mod('numba', jit=jit)

    class Data:
        def __init__(self, **kw):
            self.__dict__.update(kw)

        def to(self, *_a, **_k):
            return self

    class _Absent(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError('not part of the pinned path')

    class MessagePassing(nn.Module):
        def __init__(self, **k):
            super().__init__()

    class Linear(nn.Linear):
        def __init__(self, i, o, bias=True, **k):
            super().__init__(i, o, bias=bias)

    pyg = mod('torch_geometric')
    pyg.data = mod('torch_geometric.data', Data=Data)
    pyg.nn = mod('torch_geometric.nn', ChebConv=O.ChebConv, GCNConv=O.GCNConv, TransformerConv=O.TransformerConv,
                 GATConv=_Absent, GATv2Conv=_Absent, GraphConv=_Absent, MessagePassing=MessagePassing)
    pyg.nn.conv = mod('torch_geometric.nn.conv', MessagePassing=MessagePassing)
    pyg.nn.dense = mod('torch_geometric.nn.dense')
    pyg.nn.dense.linear = mod('torch_geometric.nn.dense.linear', Linear=Linear)
    pyg.nn.inits = mod('torch_geometric.nn.inits', glorot=lambda t: nn.init.xavier_uniform_(t),
                       zeros=lambda t: nn.init.zeros_(t))
    pyg.typing = mod('torch_geometric.typing', Adj=object, OptTensor=object, PairTensor=object)
    pyg.utils = mod('torch_geometric.utils', add_self_loops=None, degree=None)
    mod('torchviz', make_dot=lambda *a, **k: None)
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        mod('torch.utils.tensorboard', SummaryWriter=type('SummaryWriter', (), {
            '__init__': lambda s, *a, **k: None, 'add_scalar': lambda s, *a, **k: None, 'flush': lambda s: None}))


install_standins()
sys.path.insert(0, '/root/reference')
from model import graph_functions as RG    # noqa: E402  (the reference)
assert RG.__file__.startswith('/root/reference/'), RG.__file__
from model import utils as RU              # noqa: E402
from model import model as RM              # noqa: E402
from model import seq2seq as RS            # noqa: E402


def sort_edges(ei, attrs):
    ei = ei.numpy()
    order = np.lexsort((ei[1], ei[0]))
    return ei[:, order].astype(np.int32), attrs.detach().numpy()[order]


def dist_from_05(arr):
    return abs(abs(arr - 0.5) - 0.5)


def randomize(module, seed, scale=0.3, bscale=0.2):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (scale if p.dim() > 1 and p.shape[0] > 1 else bscale))
        for name, p in module.named_parameters():
            if 'norm' in name and name.endswith('weight'):
                p.add_(1.0)


def state_arrays(module, prefix):
    return {prefix + k: v.detach().numpy().copy() for k, v in module.state_dict().items()}      # (a copy: a later optimizer step must not reach it)


# ------------------------------------------------------------------ known-answer tests
def kats():
    out = {}
    img = np.zeros((8, 8), np.float32); img[1, 2] = 1
    out['kat1_img'], out['kat1_labels'] = img, RG.quadtree_decompose(img, thresh=.5, max_size=4)
    img = np.zeros((8, 8), np.float32); img[4, 0] = 1
    out['kat2_img'], out['kat2_labels'] = img, RG.quadtree_decompose(img, thresh=.5, max_size=4)
    img = np.zeros((6, 6), np.float32); img[5, 5] = 1
    out['kat3_img'], out['kat3_labels'] = img, RG.quadtree_decompose(img, thresh=.5, max_size=4)
    lab = np.array([[0, 0, 1], [2, 3, 3], [2, 3, 3]])
    xx, yy = torch.tensor([.5, 2, 0, 1.5]), torch.tensor([0, 0, 1.5, 1.5])
    ei, at = RG.get_adj(lab, xx, yy, use_edge_attrs=True)
    out['kat4_labels'] = lab
    out['kat4_edges_raw'] = ei.numpy()
    out['kat4_edges'], out['kat4_attrs'] = sort_edges(ei, at)
    mp, nodes, npx = RG.get_mapping(lab)
    out['kat5_mapping'], out['kat5_npix'] = mp.to_dense().numpy(), npx.numpy()
    for cond in RG.CONDITIONS:
        rng = np.random.default_rng(7)
        img = rng.random((16, 24)).astype(np.float32)
        out['kat6_img'] = img
        out['kat6_' + cond] = RG.quadtree_decompose(img, thresh=.9 if 'max' in cond else .1, max_size=8, condition=cond)
    np.savez_compressed(os.path.join(HERE, 'kat.npz'), **out)


# ------------------------------------------------------------------ graph-build cases
def graph_case(name, x, thresh, mask=None, hir=None, transform=None, condition='max_larger_than', attrs=False):
    """x: (ns, w, h, c) float32 without positional encoding."""
    xt = RU.add_positional_encoding(torch.from_numpy(x))
    g = RG.image_to_graph(xt, thresh=thresh, mask=mask, high_interest_region=hir, transform_func=transform,
                          condition=condition, use_edge_attrs=attrs)
    mapping = g['mapping'].numpy()
    labels = np.where(mapping.sum(0) > 0, mapping.argmax(0), -1).reshape(x.shape[1:3])
    ei, at = sort_edges(g['edge_index'], g['edge_attrs'])
    out = dict(x=x, thresh=np.float64(thresh), labels=labels.astype(np.int32), npix=g['n_pixels_per_node'].numpy(),
               data=g['data'].numpy(), edges=ei, attrs=at, condition=np.array(condition),
               has_transform=np.array(transform is not None), use_attrs=np.array(attrs))
    if mask is not None:
        out['mask'] = mask
    if hir is not None:
        out['hir'] = hir
    np.savez_compressed(os.path.join(HERE, f'graph_{name}.npz'), **out)
    print(name, 'N =', len(out['npix']), 'E =', ei.shape[1])


def graphs():
    c = synthetic.make_clip(11, n_digits=1, n_frames=3, pixel_noise=0.0)
    graph_case('64_1blob_clean', c, 0.1)
    c = synthetic.make_clip(12, n_digits=1, n_frames=2, pixel_noise=0.05)
    graph_case('64_1blob_noise', c, 0.1)
    c = synthetic.make_clip(13, n_digits=2, n_frames=3, pixel_noise=0.0)
    graph_case('64_2blob_clean_attrs', c, 0.1, attrs=True)
    c = synthetic.make_clip(14, canvas=(128, 128), n_digits=2, n_frames=1, pixel_noise=0.0)
    graph_case('128_2blob_clean', c, 0.1)
    c = synthetic.make_clip(15, canvas=(100, 100), n_digits=2, n_frames=2, pixel_noise=0.0)
    graph_case('100_2blob_clean', c, 0.1)
    c = synthetic.make_clip(16, canvas=(128, 64), n_digits=1, n_frames=1, pixel_noise=0.0)  # swapaxes -> (64,128) wide
    graph_case('64x128_wide', c, 0.1)
    f, m = synthetic.make_ice_like(17, shape=(96, 96), channels=5, n_frames=2)
    hir = np.zeros_like(m); hir[40:50, 60:75] = True
    graph_case('96_ice_masked', f, 0.15, mask=m, hir=hir, transform=dist_from_05)
    graph_case('96_ice_minsmaller', f[:1], 0.2, mask=m, condition='min_smaller_than')
    graph_case('96_ice_maxsmaller', f[:1], 0.6, condition='max_smaller_than')


# ------------------------------------------------------------------ flatten / unflatten
def transfers():
    c = synthetic.make_clip(21, n_digits=1, n_frames=2, pixel_noise=0.0)
    xt = RU.add_positional_encoding(torch.from_numpy(c)).requires_grad_(True)
    g = RG.image_to_graph(xt.detach(), thresh=0.1, use_edge_attrs=False)
    mapping, npx = g['mapping'], g['n_pixels_per_node']
    labels = mapping.numpy().argmax(0).reshape(64, 64).astype(np.int32)
    flat = RG.flatten(xt, mapping, npx)
    gy = torch.randn(flat.shape, generator=torch.Generator().manual_seed(1))
    (gx,) = torch.autograd.grad(flat, xt, gy)
    data = torch.randn(2, len(npx), 5, generator=torch.Generator().manual_seed(2)).requires_grad_(True)
    img = RG.unflatten(data, mapping, (64, 64))
    gi = torch.randn(img.shape, generator=torch.Generator().manual_seed(3))
    (gd,) = torch.autograd.grad(img, data, gi)
    np.savez_compressed(os.path.join(HERE, 'transfer.npz'), labels=labels, npix=npx.numpy(), img=xt.detach().numpy(),
                        flat=flat.detach().numpy(), flat_gy=gy.numpy(), flat_gx=gx.numpy(), data=data.detach().numpy(),
                        unflat=img.detach().numpy(), unflat_gi=gi.numpy(), unflat_gd=gd.numpy())


# ------------------------------------------------------------------ cells
def cells():
    c = synthetic.make_clip(31, canvas=(64, 64), n_digits=1, n_frames=1, pixel_noise=0.0)
    g = RG.image_to_graph(RU.add_positional_encoding(torch.from_numpy(c)), thresh=0.1, use_edge_attrs=False)
    ei, ew = g['edge_index'], g['edge_attrs']
    n = g['data'].shape[1]
    labels = g['mapping'].numpy().argmax(0).reshape(64, 64).astype(np.int32)
    sei, sew = sort_edges(ei, ew)
    gen = torch.Generator().manual_seed(5)
    out = dict(labels=labels, edges=sei, dist=sew)
    for tag, n_conv in (('nc1', 1), ('nc2', 2), ('nc3', 3)):
        cell = RM.GConvLSTM(4, 8, n_conv_layers=n_conv, convolution_type='ChebConv')
        randomize(cell, 40 + n_conv)
        X = torch.randn(n, 4, generator=gen).requires_grad_(True)
        H = torch.randn(n, 8, generator=gen).requires_grad_(True)
        C = torch.randn(n, 8, generator=gen).requires_grad_(True)
        Oo, Hn, Cn = cell(X, ei, ew, H, C)
        gO, gH, gC = (torch.randn(n, 8, generator=gen) for _ in range(3))
        params = list(cell.parameters())
        grads = torch.autograd.grad([Oo, Hn, Cn], [X, H, C] + params, [gO, gH, gC])
        out.update({f'{tag}_X': X.detach().numpy(), f'{tag}_H': H.detach().numpy(), f'{tag}_C': C.detach().numpy(),
                    f'{tag}_O': Oo.detach().numpy(), f'{tag}_Hn': Hn.detach().numpy(), f'{tag}_Cn': Cn.detach().numpy(),
                    f'{tag}_gO': gO.numpy(), f'{tag}_gH': gH.numpy(), f'{tag}_gC': gC.numpy(),
                    f'{tag}_gX': grads[0].numpy(), f'{tag}_gHin': grads[1].numpy(), f'{tag}_gCin': grads[2].numpy()})
        out.update(state_arrays(cell, f'{tag}_w/'))
        for (k, _), gr in zip(cell.named_parameters(), grads[3:]):
            out[f'{tag}_g/{k}'] = gr.numpy()
    # one Encoder step (2 layers, 2 conv layers) and one Decoder step with explicit concat
    enc = RS.Encoder(4, 8, 0.0, n_layers=2, convolution_type='ChebConv', rnn_type='LSTM', n_conv_layers=2)
    randomize(enc, 50)
    X = torch.randn(1, n, 4, generator=gen)
    H = torch.randn(n, 8, generator=gen)
    C = torch.randn(n, 8, generator=gen)
    hid, cel = enc(X, ei, ew, H=H, C=C)
    out.update(enc_X=X.numpy(), enc_H=H.numpy(), enc_C=C.numpy(), enc_hidden=hid.detach().numpy(), enc_cell=cel.detach().numpy())
    out.update(state_arrays(enc, 'enc_w/'))
    dec = RS.Decoder(4, 8, 0.0, n_layers=2, concat_layers_dim=1, convolution_type='ChebConv', rnn_type='LSTM')
    randomize(dec, 51)
    dec.eval()
    Xd = torch.randn(n, 4, generator=gen)
    Hd = torch.randn(2, n, 8, generator=gen)
    Cd = torch.randn(2, n, 8, generator=gen)
    cl = torch.randn(n, 1, generator=gen)
    o, hid, cel = dec(Xd, ei, ew, cl, Hd, Cd)
    out.update(dec_X=Xd.numpy(), dec_H=Hd.numpy(), dec_C=Cd.numpy(), dec_concat=cl.numpy(), dec_out=o.detach().numpy(),
               dec_hidden=hid.detach().numpy(), dec_cell=cel.detach().numpy())
    out.update(state_arrays(dec, 'dec_w/'))
    np.savez_compressed(os.path.join(HERE, 'cells.npz'), **out)


# ------------------------------------------------------------------ full rollouts
def rollout(name, x, y, concat, mask, hidden, n_layers, n_conv, thresh, t_in, t_out, transform=None, hir=None, seed=60,
            scale=0.25, bscale=0.2, conv='ChebConv'):
    in_feat = x.shape[-1] + 3
    model = RS.Seq2Seq(hidden_size=hidden, dropout=0.0, thresh=thresh, input_timesteps=t_in, input_features=in_feat,
                       output_timesteps=t_out, n_layers=n_layers, n_conv_layers=n_conv, transform_func=transform,
                       convolution_type=conv)
    randomize(model, seed, scale=scale, bscale=bscale)
    model.train()
    xt, yt, ct = torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(concat)
    step_labels = []
    orig = RG.image_to_graph

    def spy(img, *a, **k):
        g = orig(img, *a, **k)
        mp = g['mapping'].numpy()
        step_labels.append((np.where(mp.sum(0) > 0, mp.argmax(0), -1).reshape(img.shape[1:3]).astype(np.int32),
                            img[0, ..., 0].detach().numpy().copy()))
        return g
    RS.image_to_graph = spy
    try:
        outs, maps = model(xt, yt, ct, teacher_forcing_ratio=0, mask=mask, high_interest_region=hir)
    finally:
        RS.image_to_graph = orig
    shape = x.shape[1:3]
    y_hat = torch.stack([RG.unflatten(outs[i], maps[i], shape, mask) for i in range(t_out)])   # mpnnlstm.py:243-244
    mk = torch.from_numpy(mask)
    loss = torch.nn.MSELoss()(y_hat[:, ~mk], yt[:, ~mk])                                        # mpnnlstm.py:246
    loss.backward()
    out = dict(x=x, y=y, concat=concat, mask=mask, thresh=np.float64(thresh), loss=np.float64(loss.item()),
               y_hat=y_hat.detach().numpy(), hidden=np.int64(hidden), n_layers=np.int64(n_layers),
               n_conv=np.int64(n_conv), has_transform=np.array(transform is not None))
    if hir is not None:
        out['hir'] = hir
    for i, (lab, img) in enumerate(step_labels):
        out[f'labels_{i}'] = lab
        if i > 0:
            out[f'image_{i}'] = img          # the image the mesh of step i was built from
    for i, o in enumerate(outs):
        out[f'out_{i}'] = o.detach().numpy()
    out.update(state_arrays(model, 'w/'))
    for k, p in model.named_parameters():
        out['g/' + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, f'rollout_{name}.npz'), **out)
    print(name, 'loss', loss.item(), 'N per step', [len(o) for o in outs])


def rollouts():
    x, y = synthetic.make_batch(9, 0, 1, 4, 4, n_digits=1, pixel_noise=0.0)
    mask = np.zeros((64, 64), dtype=bool)
    concat = np.zeros((4, 64, 64, 1), np.float32)
    rollout('mnist64_h16', x[0], y[0], concat, mask, hidden=16, n_layers=2, n_conv=2, thresh=0.1, t_in=4, t_out=4, scale=0.06,
            bscale=0.02)
    # noisy frames, weights small enough that the model's own output keeps the meshes fine (N stays in the thousands; the
    # round-1 version of this case collapsed to a 1-node mesh by the last step)
    x, y = synthetic.make_batch(9, 5, 1, 3, 3, n_digits=1, pixel_noise=0.05, canvas=(64, 64))
    concat = (0.1 * np.random.default_rng(3).random((3, 64, 64, 1))).astype(np.float32)
    rollout('mnist64_noise_h8', x[0], y[0], concat, mask, hidden=8, n_layers=1, n_conv=1, thresh=0.1, t_in=3, t_out=3, seed=61,
            scale=0.03, bscale=0.01)
    # the reference's default depth n_layers=4: 1 + 2*4 = 9 state parts cross every re-mesh
    x, y = synthetic.make_batch(9, 7, 1, 2, 3, n_digits=1, pixel_noise=0.04, canvas=(64, 64))
    concat = (0.1 * np.random.default_rng(4).random((3, 64, 64, 1))).astype(np.float32)
    rollout('mnist64_l4_h8', x[0], y[0], concat, mask, hidden=8, n_layers=4, n_conv=2, thresh=0.1, t_in=2, t_out=3, seed=63,
            scale=0.05, bscale=0.01)
    f, m = synthetic.make_ice_like(18, shape=(64, 64), channels=3, n_frames=5)
    concat = f[2:5, ..., :1].copy() * 0.5
    rollout('ice64_masked_h8', f[:2], f[2:5, ..., :1].copy(), concat, m, hidden=8, n_layers=1, n_conv=3, thresh=0.15,
            t_in=2, t_out=3, transform=dist_from_05, seed=62, scale=0.1)


def rollouts_large():
    """Re-meshing rollouts on frames of SEVERAL 64x64 base cells (model/seq2seq.py:339-398,434-491 with the base-cell stack of
    model/graph_functions.py:199-205): (a) 96x128 with a land mask -- base cells 2x2, the lower row cropped --, hidden 8,
    2 layers, stacks of 2 ChebConvs; (b) the shape BASELINE configs[3] runs: 128x128, 5 channels, transform_func=dist_from_05,
    thresh 0.15, hidden 32, 1 layer, stacks of 3 ChebConvs (ice_exp.py:145-162).  Model seeds chosen so that every image a
    COMPARED mesh is built from keeps its pixels >= 8e-5 away from the threshold (the re-mesh after the last step is never
    used; its image may come closer)."""
    f, m = synthetic.make_ice_like(41, shape=(96, 128), channels=3, n_frames=5)
    concat = f[2:5, ..., :1].copy() * 0.5
    rollout('ice96x128_masked_h8', f[:2], f[2:5, ..., :1].copy(), concat, m, hidden=8, n_layers=2, n_conv=2, thresh=0.15,
            t_in=2, t_out=3, transform=dist_from_05, seed=79, scale=0.07)
    f, m = synthetic.make_ice_like(42, shape=(128, 128), channels=5, n_frames=5)
    concat = f[2:5, ..., :1].copy() * 0.5
    rollout('ice128_h32', f[:2], f[2:5, ..., :1].copy(), concat, m, hidden=32, n_layers=1, n_conv=3, thresh=0.15,
            t_in=2, t_out=3, transform=dist_from_05, seed=78, scale=0.05, bscale=0.02)


def rollout_gcn():
    """convolution_type='GCNConv' (model/model.py:41,50: GCNConv(add_self_loops=False)) through the reference's whole
    Seq2Seq rollout with re-meshing: two layers, stacks of two convolutions, noisy frames (meshes of 1-3 k nodes)."""
    x, y = synthetic.make_batch(9, 11, 1, 3, 4, n_digits=1, pixel_noise=0.04, canvas=(64, 64))
    mask = np.zeros((64, 64), dtype=bool)
    concat = (0.1 * np.random.default_rng(5).random((4, 64, 64, 1))).astype(np.float32)
    rollout('gcn_mnist64_h8', x[0], y[0], concat, mask, hidden=8, n_layers=2, n_conv=2, thresh=0.1, t_in=3, t_out=4, seed=65,
            scale=0.12, bscale=0.03, conv='GCNConv')


def checkpoint_case():
    """A checkpoint written the way the reference writes it (NextFramePredictorS2S.save, model/mpnnlstm.py:161-163:
    torch.save(self.model.state_dict(), '<dir>/<experiment_name>.pth')) from the reference's own Seq2Seq, plus the outputs of
    one eval-mode rollout with those weights: the build's load() must read the file (weights_only) and reproduce them."""
    x, y = synthetic.make_batch(9, 13, 1, 3, 3, n_digits=1, pixel_noise=0.0, canvas=(64, 64))
    mask = np.zeros((64, 64), dtype=bool)
    concat = (0.1 * np.random.default_rng(6).random((3, 64, 64, 1))).astype(np.float32)
    model = RS.Seq2Seq(hidden_size=8, dropout=0.0, thresh=0.1, input_timesteps=3, input_features=4, output_timesteps=3,
                       n_layers=2, n_conv_layers=2, convolution_type='ChebConv')
    randomize(model, 66, scale=0.06, bscale=0.02)
    model.eval()
    with torch.no_grad():
        outs, maps = model(torch.from_numpy(x[0]), None, torch.from_numpy(concat), teacher_forcing_ratio=0, mask=mask)
    torch.save(model.state_dict(), os.path.join(HERE, 'ref_checkpoint_h8.pth'))
    out = dict(x=x[0], y=y[0], concat=concat, mask=mask, keys=np.array(list(model.state_dict().keys())))
    for i, o in enumerate(outs):
        out[f'out_{i}'] = o.numpy()
    np.savez_compressed(os.path.join(HERE, 'checkpoint_case.npz'), **out)
    print('checkpoint', len(out['keys']), 'tensors, N per step', [len(o) for o in outs])


def rollout_headline():
    """One clip of the BENCHMARKED workload (BASELINE configs[1]: 64x64, 2 digits, noise 0.05, in=10/out=10, hidden 16,
    2 layers, 2 conv layers, dropout 0): clip 0 of bench.py's first batch (synthetic seed 2000)."""
    x, y = synthetic.make_batch(2, 0, 1, 10, 10, n_digits=2, pixel_noise=0.05, canvas=(64, 64))
    mask = np.zeros((64, 64), dtype=bool)
    concat = np.zeros((10, 64, 64, 1), np.float32)
    rollout('cfg2_mnist64', x[0], y[0], concat, mask, hidden=16, n_layers=2, n_conv=2, thresh=0.1, t_in=10, t_out=10, seed=64,
            scale=0.06, bscale=0.02)


# ------------------------------------------------------------------ pixelwise and preset static meshes
def rollout_fixed_mesh(name, x, y, concat, mask, hidden, n_layers, n_conv, t_in, t_out, static=None, seed=70, scale=0.1):
    """thresh = -inf: no quadtree, no re-mesh.  static=None -> every unmasked pixel is a node (image_to_graph_pixelwise);
    static=(max_grid_size, hir) -> preset mesh from create_static_heterogeneous_graph."""
    model = RS.Seq2Seq(hidden_size=hidden, dropout=0.0, thresh=-np.inf, input_timesteps=t_in, input_features=x.shape[-1] + 3,
                       output_timesteps=t_out, n_layers=n_layers, n_conv_layers=n_conv, convolution_type='ChebConv')
    randomize(model, seed, scale=scale, bscale=0.05)
    model.train()
    xt, yt, ct = torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(concat)
    gs, extra = None, {}
    if static is not None and isinstance(static[1], str) and static[1] == 'homogeneous':
        # uniform preset mesh, fully masked cells removed (graph_functions.py:707-737); partly masked cells keep ALL their pixels
        gs = RG.create_static_homogeneous_graph(x.shape[1:3], static[0], mask, use_edge_attrs=False)
        mp = gs['mapping'].numpy()
        extra = dict(static_labels=np.where(mp.sum(0) > 0, mp.argmax(0), -1).reshape(x.shape[1:3]).astype(np.int32),
                     static_npix=gs['n_pixels_per_node'].numpy(), max_grid_size=np.int64(static[0]))
        ei, at = sort_edges(gs['edge_index'], gs['edge_attrs'])
        extra.update(static_edges=ei, static_dist=at)
    elif static is not None:
        gs = RG.create_static_heterogeneous_graph(x.shape[1:3], static[0], mask, high_interest_region=static[1], use_edge_attrs=False)
        mp = gs['mapping'].numpy()
        extra = dict(static_labels=np.where(mp.sum(0) > 0, mp.argmax(0), -1).reshape(x.shape[1:3]).astype(np.int32),
                     static_npix=gs['n_pixels_per_node'].numpy(), max_grid_size=np.int64(static[0]))
        ei, at = sort_edges(gs['edge_index'], gs['edge_attrs'])
        extra.update(static_edges=ei, static_dist=at)
        if static[1] is not None:
            extra['hir'] = static[1]
    outs, maps = model(xt, yt, ct, teacher_forcing_ratio=0, mask=mask, graph_structure=gs)
    shape = x.shape[1:3]
    y_hat = torch.stack([RG.unflatten(outs[i], maps[i], shape, mask) for i in range(t_out)])
    mk = torch.from_numpy(mask)
    loss = torch.nn.MSELoss()(y_hat[:, ~mk], yt[:, ~mk])
    loss.backward()
    out = dict(x=x, y=y, concat=concat, mask=mask, loss=np.float64(loss.item()), y_hat=y_hat.detach().numpy(),
               hidden=np.int64(hidden), n_layers=np.int64(n_layers), n_conv=np.int64(n_conv), **extra)
    for i, o in enumerate(outs):
        out[f'out_{i}'] = o.detach().numpy()
    out.update(state_arrays(model, 'w/'))
    for k, p in model.named_parameters():
        out['g/' + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, f'fixed_{name}.npz'), **out)
    print(name, 'loss', loss.item(), 'N', len(outs[0]))


def fixed_meshes():
    f, m = synthetic.make_ice_like(19, shape=(48, 64), channels=3, n_frames=5)
    concat = f[2:5, ..., :1].copy() * 0.5
    rollout_fixed_mesh('pixelwise48x64', f[:2], f[2:5, ..., :1].copy(), concat, m, hidden=8, n_layers=1, n_conv=3, t_in=2, t_out=3)
    hir = np.zeros_like(m); hir[10:20, 30:44] = True
    rollout_fixed_mesh('static48x64', f[:2], f[2:5, ..., :1].copy(), concat, m, hidden=8, n_layers=2, n_conv=1, t_in=2, t_out=3,
                       static=(8, hir), seed=71)


def homogeneous_mesh():
    f, m = synthetic.make_ice_like(19, shape=(48, 64), channels=3, n_frames=5)
    m = m.copy()
    m[:8, :16] = True                       # two whole 8x8 cells under the mask: they must disappear from the mesh
    concat = f[2:5, ..., :1].copy() * 0.5
    rollout_fixed_mesh('homog48x64', f[:2], f[2:5, ..., :1].copy(), concat, m, hidden=8, n_layers=1, n_conv=2, t_in=2, t_out=3,
                       static=(8, 'homogeneous'), seed=72)


# ------------------------------------------------------------------ trainer parity features
def rollout_variants():
    """teacher forcing (ratio 1 -> deterministic), binary head (sigmoid + BCE), truncated BPTT chunk loop."""
    x, y = synthetic.make_batch(9, 20, 1, 3, 4, n_digits=1, pixel_noise=0.0)
    x, y = x[0], np.clip(y[0], 0.0, 1.0)
    mask = np.zeros((64, 64), dtype=bool)
    concat = (0.05 * np.random.default_rng(5).random((4, 64, 64, 1))).astype(np.float32)
    xt, yt, ct, mk = torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(concat), torch.from_numpy(mask)

    def build(binary=False, seed=80):
        model = RS.Seq2Seq(hidden_size=8, dropout=0.0, thresh=0.1, input_timesteps=3, input_features=4, output_timesteps=4,
                           n_layers=1, n_conv_layers=2, convolution_type='ChebConv', binary=binary)
        randomize(model, seed, scale=0.08, bscale=0.02)
        model.train()
        return model

    def dump(name, model, outs, loss, extra=None):
        out = dict(x=x, y=y, concat=concat, mask=mask, loss=np.float64(loss.item()))
        for i, o in enumerate(outs):
            out[f'out_{i}'] = o.detach().numpy()
        out.update(state_arrays(model, 'w/'))
        for k, p in model.named_parameters():
            out['g/' + k] = p.grad.numpy() if p.grad is not None else np.zeros(p.shape, np.float32)
        out.update(extra or {})
        np.savez_compressed(os.path.join(HERE, f'variant_{name}.npz'), **out)
        print(name, 'loss', loss.item(), 'N', [len(o) for o in outs])

    # teacher forcing on every step: the next mesh / input come from the ground truth (seq2seq.py:448-458)
    model = build()
    outs, maps = model(xt, yt, ct, teacher_forcing_ratio=1.0, mask=mask)
    y_hat = torch.stack([RG.unflatten(outs[i], maps[i], (64, 64), mask) for i in range(4)])
    loss = torch.nn.MSELoss()(y_hat[:, ~mk], yt[:, ~mk])
    loss.backward()
    dump('teacher', model, outs, loss)

    # binary head: sigmoid output (seq2seq.py:177-178) + BCELoss (mpnnlstm.py:171)
    model = build(binary=True, seed=81)
    outs, maps = model(xt, yt, ct, teacher_forcing_ratio=0, mask=mask)
    y_hat = torch.stack([RG.unflatten(outs[i], maps[i], (64, 64), mask) for i in range(4)])
    loss = torch.nn.BCELoss()(y_hat[:, ~mk], yt[:, ~mk])
    loss.backward()
    dump('binary', model, outs, loss)

    # remesh_input=True (seq2seq.py:266-276, 323-324, 493-527): every encoder step on the mesh of ITS frame; x carries
    # input_timesteps + 1 frames because the reference re-meshes to frame t + 1 after the last step too
    model = RS.Seq2Seq(hidden_size=8, dropout=0.0, thresh=0.1, input_timesteps=3, input_features=4, output_timesteps=4,
                       n_layers=1, n_conv_layers=2, convolution_type='ChebConv', remesh_input=True)
    randomize(model, 83, scale=0.08, bscale=0.02)
    model.train()
    x4 = np.concatenate([x, y[:1]], axis=0)
    outs, maps = model(torch.from_numpy(x4), yt, ct, teacher_forcing_ratio=0, mask=mask)
    y_hat = torch.stack([RG.unflatten(outs[i], maps[i], (64, 64), mask) for i in range(4)])
    loss = torch.nn.MSELoss()(y_hat[:, ~mk], yt[:, ~mk])
    loss.backward()
    dump('remesh_input', model, outs, loss, dict(x=x4))

    # truncated BPTT with truncated_backprop = 2 (mpnnlstm.py:281-315): every chunk re-encodes, zeroes the gradients
    # and unrolls its own steps from the encoder state; only the last chunk's gradient reaches optimizer.step()
    model = build(seed=82)
    tb, t_out, step, losses, last_outs = 2, 4, 0, [], None
    while step < t_out:
        step = min(step + tb, t_out + 1)
        steps = range(step - tb, step)
        model.zero_grad()
        model.process_inputs(xt, mask=mask)
        outs, maps = model.unroll_output(steps, yt, concat_layers=ct, teacher_forcing_ratio=0, mask=mask, remesh_every=1)
        y_hat = torch.stack([RG.unflatten(outs[i], maps[i], (64, 64), mask) for i in range(len(outs))])
        loss = torch.nn.MSELoss()(y_hat[:, ~mk], yt[steps][:, ~mk])
        loss.backward(retain_graph=True)
        losses.append(loss.item())
        last_outs = outs
    dump('tbptt', model, last_outs, loss, dict(chunk_losses=np.array(losses)))


# ------------------------------------------------------------------ TransformerConv
def transformer_cases():
    """GConvLSTM cell with TransformerConv stacks (edge attrs [angle, dist], self pairs of multi-pixel cells are real
    attention keys) and a masked rollout.  The conv arithmetic is the oracle's restatement (parity unpinned); the cell,
    encoder / decoder wiring and re-mesh control flow are the reference's own code."""
    c = synthetic.make_clip(33, canvas=(64, 64), n_digits=1, n_frames=1, pixel_noise=0.0)
    g = RG.image_to_graph(RU.add_positional_encoding(torch.from_numpy(c)), thresh=0.1, use_edge_attrs=True)
    ei, ea = g['edge_index'], g['edge_attrs']
    n = g['data'].shape[1]
    labels = g['mapping'].numpy().argmax(0).reshape(64, 64).astype(np.int32)
    sei, sea = sort_edges(ei, ea)
    gen = torch.Generator().manual_seed(6)
    out = dict(labels=labels, edges=sei, attrs=sea)
    cell = RM.GConvLSTM(4, 8, n_conv_layers=2, convolution_type='TransformerConv')
    randomize(cell, 90)
    cell.eval()                                   # attention dropout off (cannot be RNG matched)
    X = torch.randn(n, 4, generator=gen).requires_grad_(True)
    H = torch.randn(n, 8, generator=gen).requires_grad_(True)
    C = torch.randn(n, 8, generator=gen).requires_grad_(True)
    Oo, Hn, Cn = cell(X, ei, ea, H, C)
    gO, gH, gC = (torch.randn(n, 8, generator=gen) for _ in range(3))
    grads = torch.autograd.grad([Oo, Hn, Cn], [X, H, C] + list(cell.parameters()), [gO, gH, gC])
    out.update(X=X.detach().numpy(), H=H.detach().numpy(), C=C.detach().numpy(), O=Oo.detach().numpy(),
               Hn=Hn.detach().numpy(), Cn=Cn.detach().numpy(), gO=gO.numpy(), gH=gH.numpy(), gC=gC.numpy(),
               gX=grads[0].numpy(), gHin=grads[1].numpy(), gCin=grads[2].numpy())
    out.update(state_arrays(cell, 'w/'))
    for (k, _), gr in zip(cell.named_parameters(), grads[3:]):
        out['g/' + k] = gr.numpy()
    np.savez_compressed(os.path.join(HERE, 'transformer_cell.npz'), **out)
    print('transformer cell N =', n, 'E =', sei.shape[1])

    f, m = synthetic.make_ice_like(23, shape=(64, 64), channels=3, n_frames=5)
    x, y = f[:2], f[2:5, ..., :1].copy()
    concat = y * 0.5
    model = RS.Seq2Seq(hidden_size=8, dropout=0.0, thresh=0.15, input_timesteps=2, input_features=6, output_timesteps=3,
                       n_layers=1, n_conv_layers=2, transform_func=dist_from_05, convolution_type='TransformerConv')
    randomize(model, 91, scale=0.1, bscale=0.05)
    model.eval()
    xt, yt, ct, mk = torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(concat), torch.from_numpy(m)
    outs, maps = model(xt, yt, ct, teacher_forcing_ratio=0, mask=m)
    y_hat = torch.stack([RG.unflatten(outs[i], maps[i], (64, 64), m) for i in range(3)])
    loss = torch.nn.MSELoss()(y_hat[:, ~mk], yt[:, ~mk])
    loss.backward()
    out = dict(x=x, y=y, concat=concat, mask=m, loss=np.float64(loss.item()))
    for i, o in enumerate(outs):
        out[f'out_{i}'] = o.detach().numpy()
    out.update(state_arrays(model, 'w/'))
    for k, p in model.named_parameters():
        out['g/' + k] = p.grad.numpy() if p.grad is not None else np.zeros(p.shape, np.float32)
    np.savez_compressed(os.path.join(HERE, 'transformer_rollout.npz'), **out)
    print('transformer rollout loss', loss.item(), 'N', [len(o) for o in outs])


# ------------------------------------------------------------------ what ice_exp.py really runs (+ predict, climatology)
def climatology_from_base(base):
    """(1, 365, w, h) daily normals from a (w, h) base field: the fixture stores `base` only (tests/helpers.py repeats this)."""
    d = np.arange(365, dtype=np.float32)[:, None, None]
    return (base[None] * (0.5 + 0.5 * np.cos(2 * np.pi * d / 365.0)) + 0.001 * d)[None].astype(np.float32)


class _DS:
    def __init__(self, shape):
        self.image_shape = shape


class _Loader(list):
    pass


def ice_exp_case():
    """ice_exp.py:48,145,153-162: thresh = -inf (pixelwise mesh) x TransformerConv x hidden 32 x n_conv_layers 3 x n_layers 1,
    5 input variables, land mask, climatology concat through NextFramePredictorS2S.get_climatology_array, and predict().
    Run through the reference's own trainer class (model/mpnnlstm.py); eval mode (attention / decoder dropout cannot be RNG
    matched).  Launch dates sit at 12:00 so that datetime.fromtimestamp gives the same day in any time zone; the second clip
    crosses the year end (day-of-year index 363, 364, 0)."""
    from model import mpnnlstm as RP
    shape = (24, 32)
    f0, m = synthetic.make_ice_like(25, shape=shape, channels=5, n_frames=6)
    f1, _ = synthetic.make_ice_like(26, shape=shape, channels=5, n_frames=6)
    x = np.stack([f0[:3], f1[:3]])
    y = np.stack([f0[3:, ..., :1], f1[3:, ..., :1]])
    base = synthetic.make_ice_like(27, shape=shape, channels=1, n_frames=1)[0][0, ..., 0]
    clim = torch.from_numpy(climatology_from_base(base))
    import datetime as _dt
    launch = np.array([int(_dt.datetime(2010, 3, 1, 12, tzinfo=_dt.timezone.utc).timestamp()) * 10 ** 9,
                       int(_dt.datetime(2010, 12, 30, 12, tzinfo=_dt.timezone.utc).timestamp()) * 10 ** 9], dtype=np.int64)
    kw = dict(hidden_size=32, dropout=0.1, n_layers=1, transform_func=dist_from_05, dummy=False, n_conv_layers=3,
              rnn_type='LSTM', convolution_type='TransformerConv')
    nfp = RP.NextFramePredictorS2S(thresh=-np.inf, input_features=5, input_timesteps=3, output_timesteps=3,
                                   device=torch.device('cpu'), transform_func=dist_from_05, model_kwargs=kw)
    randomize(nfp.model, 95, scale=0.1, bscale=0.05)
    nfp.model.eval()
    mk = torch.from_numpy(m)
    out = dict(x=x, y=y, mask=m, clim_base=base, launch=launch, n_params=np.int64(nfp.get_n_params()))
    for c in range(2):
        ld = torch.tensor([launch[c]])
        concat = nfp.get_climatology_array(clim, ld)
        out[f'concat_{c}'] = concat.numpy()
        nfp.model.zero_grad()
        outs, maps = nfp.model(torch.from_numpy(x[c]), torch.from_numpy(y[c]), concat, teacher_forcing_ratio=0, mask=m)
        y_hat = torch.stack([RG.unflatten(outs[i], maps[i], shape, m) for i in range(3)])
        loss = torch.nn.MSELoss()(y_hat[:, ~mk], torch.from_numpy(y[c])[:, ~mk])
        loss.backward()
        out[f'loss_{c}'] = np.float64(loss.item())
        for i, o in enumerate(outs):
            out[f'out_{c}_{i}'] = o.detach().numpy()
        if c == 0:
            for k, p in nfp.model.named_parameters():
                out['g/' + k] = p.grad.numpy().copy() if p.grad is not None else np.zeros(p.shape, np.float32)
    out.update(state_arrays(nfp.model, 'w/'))
    loader = _Loader([(torch.from_numpy(x[c])[None], torch.from_numpy(y[c])[None], torch.tensor([launch[c]])) for c in range(2)])
    loader.dataset = _DS(shape)
    pred = nfp.predict(loader, clim, mask=m)
    out['pred'] = pred
    np.savez_compressed(os.path.join(HERE, 'variant_ice_exp.npz'), **out)
    print('ice_exp case: losses', out['loss_0'], out['loss_1'], 'N', len(outs[0]), 'params', int(out['n_params']), 'pred', pred.shape)


def _mesh_arrays(gs, shape, prefix):
    """labels / npix / sorted edges / [angle, dist] of a preset mesh dict (the dense mapping never leaves this script)."""
    mp = gs['mapping'].cpu().numpy()
    ei, at = sort_edges(gs['edge_index'].cpu(), gs['edge_attrs'].cpu())
    return {prefix + 'labels': np.where(mp.sum(0) > 0, mp.argmax(0), -1).reshape(shape).astype(np.int32),
            prefix + 'npix': gs['n_pixels_per_node'].cpu().numpy(), prefix + 'edges': ei, prefix + 'attrs': at}


def ice_exp_preset_cases():
    """ice_exp.py exp 9 / 10 (:82-87, 109-112, 127-130, 184-206): preset heterogeneous / homogeneous mesh with max_grid_size=4,
    use_edge_attrs=True, resolution=1/6 (half) and 1/12 (full), TransformerConv x hidden 32 x 3 conv layers x land mask, ONE
    model trained by the reference's own NextFramePredictorS2S.train() at half resolution and then at full resolution
    (truncated_backprop=0), followed by predict().  Two deviations from the script, both forced by HEAD: the half-resolution
    phase gets a climatology too (without one HEAD fails at fc_out1, SURVEY 3.5 -- the script's own commented-out lines
    :106-107, 188), and the model stays in eval mode (train() never switches modes itself; attention / decoder dropout
    cannot be RNG matched).  Also a forward + backward trace at the initial weights on the full-resolution mesh.
    exp 1 (:64-65, 145): GCNConv on the pixelwise mesh (edge_attrs=None -> unit weights), same recordings."""
    from model import mpnnlstm as RP
    import datetime as _dt
    half, full = (24, 32), (48, 64)
    m_half = synthetic.make_ice_like(31, shape=half, channels=1, n_frames=1)[1]
    m_full = synthetic.make_ice_like(32, shape=full, channels=1, n_frames=1)[1]
    hir = np.zeros_like(m_full); hir[10:20, 30:44] = True
    t_in, t_out = 2, 3

    def clips(seed0, n, shape):
        fs = [synthetic.make_ice_like(seed0 + i, shape=shape, channels=5, n_frames=t_in + t_out)[0] for i in range(n)]
        return np.stack([f[:t_in] for f in fs]), np.stack([f[t_in:, ..., :1] for f in fs])
    launch = np.array([int(_dt.datetime(2010, 6, 1 + i, 12, tzinfo=_dt.timezone.utc).timestamp()) * 10 ** 9 for i in range(3)],
                      dtype=np.int64)

    def loader(x, y, shape):
        ld = _Loader([(torch.from_numpy(x[c])[None], torch.from_numpy(y[c])[None], torch.tensor([launch[c]])) for c in range(len(x))])
        ld.dataset = _DS(shape)
        return ld
    xh, yh = clips(300, 3, half)
    xf, yf = clips(310, 3, full)
    base_h = synthetic.make_ice_like(34, shape=half, channels=1, n_frames=1)[0][0, ..., 0]
    base_f = synthetic.make_ice_like(33, shape=full, channels=1, n_frames=1)[0][0, ..., 0]
    clim_h, clim_f = torch.from_numpy(climatology_from_base(base_h)), torch.from_numpy(climatology_from_base(base_f))
    lr = 1e-3

    for name, conv, preset in (('ice_exp9', 'TransformerConv', 'heterogeneous'), ('ice_exp10', 'TransformerConv', 'homogeneous'),
                               ('ice_exp1', 'GCNConv', False)):
        kw = dict(hidden_size=32, dropout=0.1, n_layers=1, transform_func=dist_from_05, dummy=False, n_conv_layers=3,
                  rnn_type='LSTM', convolution_type=conv)
        nfp = RP.NextFramePredictorS2S(thresh=-np.inf, experiment_name=name, input_features=5, input_timesteps=t_in,
                                       output_timesteps=t_out, device=torch.device('cpu'), transform_func=dist_from_05,
                                       binary=False, debug=False, model_kwargs=kw)
        randomize(nfp.model, 120 + len(name), scale=0.08, bscale=0.04)
        nfp.model.eval()
        out = dict(x_half=xh, y_half=yh, x=xf, y=yf, mask_half=m_half, mask=m_full, hir=hir, clim_base_half=base_h,
                   clim_base=base_f, launch=launch, lr=np.float64(lr), n_params=np.int64(nfp.get_n_params()),
                   preset=np.array(str(preset)), conv=np.array(conv))
        gs_h = gs_f = None
        if preset == 'heterogeneous':
            gs_h = RG.create_static_heterogeneous_graph(half, 4, m_half, use_edge_attrs=True, resolution=1/6, device=torch.device('cpu'))
            gs_f = RG.create_static_heterogeneous_graph(full, 4, m_full, use_edge_attrs=True, resolution=1/12, device=torch.device('cpu'))
        elif preset == 'homogeneous':
            gs_h = RG.create_static_homogeneous_graph(half, 4, m_half, use_edge_attrs=True, resolution=1/6, device=torch.device('cpu'))
            gs_f = RG.create_static_homogeneous_graph(full, 4, m_full, use_edge_attrs=True, resolution=1/12, device=torch.device('cpu'))
        if preset:
            out.update(_mesh_arrays(gs_h, half, 'half_'))
            out.update(_mesh_arrays(gs_f, full, 'full_'))
        out.update(state_arrays(nfp.model, 'w/'))
        # forward + backward at the initial weights, full resolution, clip 0 (mpnnlstm.py:233-249)
        mk = torch.from_numpy(m_full)
        concat = nfp.get_climatology_array(clim_f, torch.tensor([launch[0]]))
        nfp.model.zero_grad()
        outs, maps = nfp.model(torch.from_numpy(xf[0]), torch.from_numpy(yf[0]), concat, teacher_forcing_ratio=0, mask=m_full,
                               high_interest_region=hir, graph_structure=gs_f)
        y_hat = torch.stack([RG.unflatten(outs[i], maps[i], full, m_full) for i in range(t_out)])
        loss = torch.nn.MSELoss()(y_hat[:, ~mk], torch.from_numpy(yf[0])[:, ~mk])
        loss.backward()
        out['loss0'] = np.float64(loss.item())
        for i, o in enumerate(outs):
            out[f'out_{i}'] = o.detach().numpy()
        for k, p in nfp.model.named_parameters():
            out['g/' + k] = p.grad.numpy().copy() if p.grad is not None else np.zeros(p.shape, np.float32)
        nfp.model.zero_grad()
        # the script's training sequence: two clips train / one clip test per phase, one epoch each
        if preset:
            nfp.train(loader(xh[:2], yh[:2], half), loader(xh[2:], yh[2:], half), clim_h, lr=lr, n_epochs=1, mask=m_half,
                      truncated_backprop=0, graph_structure=gs_h)
            out.update(state_arrays(nfp.model, 'w1/'))
        nfp.train(loader(xf[:2], yf[:2], full), loader(xf[2:], yf[2:], full), clim_f, lr=lr, n_epochs=1, mask=m_full,
                  high_interest_region=hir, truncated_backprop=0, graph_structure=gs_f)
        out['train_loss'] = nfp.loss['train_loss'].values.astype(np.float64)
        out['test_loss'] = nfp.loss['test_loss'].values.astype(np.float64)
        out.update(state_arrays(nfp.model, 'w2/'))
        out['pred'] = nfp.predict(loader(xf[2:], yf[2:], full), clim_f, mask=m_full, graph_structure=gs_f)
        np.savez_compressed(os.path.join(HERE, f'variant_{name}.npz'), **out)
        print(name, 'N', len(outs[0]), 'loss0', out['loss0'], 'train', out['train_loss'], 'test', out['test_loss'],
              'params', int(out['n_params']), 'pred', out['pred'].shape)


def teacher_fixed_cases():
    """Teacher forcing where no re-mesh happens (model/seq2seq.py:420-425): the decoder's next input is rebuilt as
    [flatten(teacher + positional encoding) | RAW n_pixels_per_node] -- on a pixelwise mesh (thresh = -inf), on a preset
    heterogeneous mesh, and on quadtree meshes with remesh_every = 2 (every other step takes this branch)."""
    f, m = synthetic.make_ice_like(29, shape=(32, 40), channels=3, n_frames=6)
    x, y = f[:2], f[2:6, ..., :1].copy()
    concat = y * 0.5
    mk = torch.from_numpy(m)
    xt, yt, ct = torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(concat)

    def run(name, thresh, gs=None, remesh_every=1, mask=m, seed=100, extra=None):
        model = RS.Seq2Seq(hidden_size=8, dropout=0.0, thresh=thresh, input_timesteps=2, input_features=6, output_timesteps=4,
                           n_layers=2, n_conv_layers=1, convolution_type='ChebConv')
        randomize(model, seed, scale=0.1, bscale=0.05)
        model.train()
        outs, maps = model(xt, yt, ct, teacher_forcing_ratio=1.0, mask=mask, graph_structure=gs, remesh_every=remesh_every)
        y_hat = torch.stack([RG.unflatten(outs[i], maps[i], x.shape[1:3], mask) for i in range(4)])
        mk_ = torch.from_numpy(mask)
        loss = torch.nn.MSELoss()(y_hat[:, ~mk_], yt[:, ~mk_])
        loss.backward()
        out = dict(x=x, y=y, concat=concat, mask=mask, loss=np.float64(loss.item()), thresh=np.float64(thresh),
                   remesh_every=np.int64(remesh_every))
        for i, o in enumerate(outs):
            out[f'out_{i}'] = o.detach().numpy()
        out.update(state_arrays(model, 'w/'))
        for k, p in model.named_parameters():
            out['g/' + k] = p.grad.numpy() if p.grad is not None else np.zeros(p.shape, np.float32)
        out.update(extra or {})
        np.savez_compressed(os.path.join(HERE, f'variant_{name}.npz'), **out)
        print(name, 'loss', loss.item(), 'N', [len(o) for o in outs])

    run('teacher_pixelwise', -np.inf)
    hir = np.zeros_like(m); hir[8:14, 20:30] = True
    gs = RG.create_static_heterogeneous_graph(x.shape[1:3], 8, m, high_interest_region=hir, use_edge_attrs=False)
    run('teacher_static', -np.inf, gs=gs, seed=101, extra=dict(hir=hir, max_grid_size=np.int64(8)))
    run('teacher_every2', 0.15, remesh_every=2, mask=np.zeros_like(m), seed=102)


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(4)
    only = os.environ.get('GOLDEN_ONLY', '')
    if not only:
        kats()
        graphs()
        transfers()
        cells()
    if only in ('', 'fixed'):
        fixed_meshes()
    if only in ('', 'homog'):
        homogeneous_mesh()
    if only in ('', 'variants'):
        rollout_variants()
    if only in ('', 'transformer'):
        transformer_cases()
    if only in ('', 'rollouts'):
        rollouts()
    if only in ('', 'large'):
        rollouts_large()
    if only in ('', 'gcn'):
        rollout_gcn()
    if only in ('', 'checkpoint'):
        checkpoint_case()
    if only in ('', 'headline'):
        rollout_headline()
    if only in ('', 'ice_exp'):
        ice_exp_case()
    if only in ('', 'teacher_fixed'):
        teacher_fixed_cases()
    if only in ('', 'ice_presets'):
        ice_exp_preset_cases()
    print('golden vectors written to', HERE)
