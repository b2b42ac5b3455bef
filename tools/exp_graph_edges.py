"""train(use_graph=True) / predict() across the options the reference's callers combine (ragged last batch, masks, high-interest
region, preset meshes, pixelwise meshes, the three convolutions, the default truncation length) -- diagnostics."""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from torch.utils.data import DataLoader
from model.mpnnlstm import NextFramePredictorS2S
from model.graph_functions import create_static_heterogeneous_graph, create_static_homogeneous_graph
from helpers import TinyMovingMNISTDataset
dev = torch.device('cuda', 0)
torch.set_num_threads(8)


def quiet(fn):
    so = sys.stdout; sys.stdout = open(os.devnull, 'w')
    try:
        return fn()
    finally:
        sys.stdout = so


def case(name, fn):
    try:
        print(f'{name}: ok {fn()}', flush=True)
    except Exception as e:
        print(f'{name}: {type(e).__name__}: {str(e)[:300]}', flush=True)


def run(conv='ChebConv', bs=2, n=5, graph=True, mask='zeros', hir=False, preset=None, tb=0, thresh=0.1, h=8):
    torch.manual_seed(0)
    ds = TinyMovingMNISTDataset(n, 3, 2, n_digits=1, canvas_size=(32, 32), digit_size=(12, 12))
    nfp = NextFramePredictorS2S(thresh=thresh, input_features=1, input_timesteps=3, output_timesteps=2, device=dev,
                                model_kwargs=dict(hidden_size=h, dropout=0.1, n_layers=1, convolution_type=conv))
    nfp.model.train()
    m = None if mask == 'none' else np.zeros((32, 32), dtype=bool)
    if mask == 'land':
        m[:, 20:] = True
    hi = None
    if hir:
        hi = np.zeros((32, 32), dtype=bool); hi[8:16, 8:16] = True
    gs = None
    mm = m if m is not None else np.zeros((32, 32), dtype=bool)
    if preset == 'het':
        gs = create_static_heterogeneous_graph((32, 32), 4, mm, high_interest_region=hi, use_edge_attrs=conv == 'TransformerConv', device=dev)
    if preset == 'hom':
        gs = create_static_homogeneous_graph((32, 32), 4, mm, use_edge_attrs=conv == 'TransformerConv', device=dev)
    tr, te = DataLoader(ds, batch_size=bs), DataLoader(torch.utils.data.Subset(ds, range(1)), batch_size=1)
    te.dataset.image_shape = ds.image_shape
    quiet(lambda: nfp.train(tr, te, None, lr=1e-3, n_epochs=3, mask=m, high_interest_region=hi, truncated_backprop=tb, graph_structure=gs, use_graph=graph))
    pred = nfp.predict(te, None, mask=m, high_interest_region=hi, graph_structure=gs)
    return [round(v, 4) for v in nfp.train_loss], pred.shape, bool(np.isfinite(np.nan_to_num(pred)).all())


for kw in (dict(), dict(graph=False), dict(mask='none'), dict(mask='land', hir=True), dict(preset='het', mask='land'), dict(preset='hom', mask='land'),
           dict(conv='TransformerConv'), dict(conv='TransformerConv', preset='het', mask='land', hir=True), dict(conv='GCNConv', thresh=-np.inf, mask='land'),
           dict(tb=45), dict(bs=1, n=3), dict(conv='TransformerConv', thresh=-np.inf, mask='land', h=32)):
    case(str(kw), lambda: run(**kw))
print('done')
