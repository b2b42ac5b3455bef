"""CPU-side checks (no GPU): the C-ABI library loads and exports every declared symbol, the host
logic (weight composition, packing, state-dict layout, data generator, sharding) is correct."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from qtmpnn import _lib
    header = open(os.path.join(ROOT, 'include', 'qtmpnn.h')).read()
    declared = set(re.findall(r'\b(qt_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} is declared in include/qtmpnn.h but not exported'
    assert declared == set(_lib.exported_names()), declared ^ set(_lib.exported_names())
    assert lib.qt_abi_version() == 1


def test_ctypes_signatures_match_the_header():
    """Every binding in qtmpnn._lib lists exactly as many arguments as the header declares, pointer / int / float in the
    same places (a short list makes ctypes push garbage for the rest: a host-side crash, not an error code)."""
    from qtmpnn import _lib
    header = re.sub(r'/\*.*?\*/', ' ', open(os.path.join(ROOT, 'include', 'qtmpnn.h')).read(), flags=re.S)
    decls = dict(re.findall(r'\b(qt_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', header, flags=re.S))
    kinds = {ctypes.c_void_p: 'p', ctypes.c_int: 'i', ctypes.c_float: 'f', ctypes.c_uint32: 'u', ctypes.c_int64: 'l'}
    checked = 0
    for name, sig in _lib._SIGNATURES.items():
        params = [p.strip() for p in decls[name].split(',')] if decls[name].strip() not in ('', 'void') else []
        want = ''
        for p_ in params:
            if '*' in p_:
                want += 'p'
            elif p_.startswith('float'):
                want += 'f'
            elif p_.startswith('uint32_t'):
                want += 'u'
            elif p_.startswith('int64_t'):
                want += 'l'
            else:
                want += 'i'
        got = ''.join(kinds[t] for t in sig)
        assert got == want, f'{name}: binding {got} vs header {want}'
        checked += 1
    assert checked >= 35


def test_bad_arguments_fail_loudly_without_gpu():
    from qtmpnn import _lib
    lib = _lib.load()
    rc = lib.qt_spmm(None, None, None, 4, None, 4, None, 1.0, None, 0.0, None, 0.0, None, None)
    assert rc == -1 and b'qt_spmm' in lib.qt_last_error()
    assert lib.qt_lstm_fwd(None, None, 0, None, 0, None, None, None, 1, None, 16, None, None, None, None, None) == -1
    # every entry point refuses NULL / inconsistent arguments with an error code and a message naming itself -- no launch
    null_calls = {
        'qt_lstm_bwd_dgrad': (None, 0, None, 0, None, 0, None, None, 0, None, None, 4, None, 16, None, None, None, 0, None, None, None, 3, 16, 0, None, None, 0, None, 0, None, None),
        'qt_lstm_bwd_fused': (None, 0, None, 0, None, 0, None, None, 0, None, None, 4, None, 16, None, None, 0, None, 3, 16, 0, None, None,
                              None, 0, None, None, 0, None, 3, 16, 0, None, 0, None, 0, None),
        'qt_flat_adam': (None, None, None, None, 0, None, None, 0.0, 0.9, 0.999, 1e-8, 10.0, None, None),
        'qt_compose2_fwd': (None,) * 8 + (3, 1, 3, 4, 4, 16, None, None, None, None, None),
        'qt_compose_step_fwd': (None,) * 4 + (3, 1, 3, 4, 16, None, None, None),
        'qt_remesh': (None, None, None, 0, None, None, 0, None, None, None, 1, 1, 64, 64, 4, None, None, None, None, 0, None, None),
        'qt_pool_clip': (None, 1, 0, 1, None, None, None, 1, 1, 64, 64, 4, None, 1, 0, None),
        'qt_remesh_clip': (None, None, None, 0, None, None, 0, None, None, None, None, 1, 1, 64, 64, None, None, 0, None, 0, None),
        'qt_attn_fwd': (None,) * 6 + (0, None, 8, 8, 4, None, 1.0, 0, None, None, None, 1, 0, 0, 0, 0, None),
        'qt_proj_group': (None, 0, 0, 1, 4, None, None, None, 0, 1, 1, 4, None, 0, 0, 0, 4, None, None),
        'qt_wgrad_groups': (1, None, None, None, None, None, None, 4, 4, 4, 4, 0, 1, 0, 0, 0, None, None),
        'qt_dense_sb': (None, 0, 16, None, None, 1, 4, 0, 4, None, None, None, None),
        'qt_cheb_clip_fwd': (None,) * 8 + (1, 4, 3, 4, None, 0, None, 0, None, 0, None, 0, None),
        'qt_cheb_clip_bwd': (None,) * 8 + (1, 4, 3, 4, None, 0, None, 0, 0, None),
        'qt_cheb_tile_fwd': (None,) * 15 + (1, 4, 2, 4, 3, 4, None, 0, None, 0, None, 0, None, None),
        'qt_cheb_tile_bwd': (None,) * 15 + (1, 4, 2, 4, 3, 4, None, 0, None, 0, None),
        'qt_edges_norm_tiles': (None,) * 4 + (4, None, None, None, None, None, 4, 2, None, None, None, None, None, None, None, None, None),
        'qt_dense2': (None, 0, None, None, 0, None, 1, 4, 0, None, None, None, 0, None, 1, 4, 0, 4, None, 0, None, 0, None, None, None, 0, None, None, None),
    }
    for name, args in null_calls.items():
        assert getattr(lib, name)(*args) == -1, name
        assert name.encode() in lib.qt_last_error(), (name, lib.qt_last_error())


def test_product_path_refuses_cpu_tensors():
    from qtmpnn.mesh import build_mesh
    with pytest.raises(RuntimeError, match='GPU'):
        build_mesh(src=torch.zeros(1, 8, 8), thresh=0.1)


def test_state_dict_layout_matches_reference():
    """KAT-6: 34 513 parameters in 238 tensors with the reference's key names (golden 'w/' keys)."""
    from model.seq2seq import Seq2Seq
    from oracle import qt_oracle as O
    m = Seq2Seq(hidden_size=16, dropout=0.1, thresh=0.1, input_timesteps=10, input_features=4, output_timesteps=10, n_layers=2)
    sd = m.state_dict()
    assert sum(p.numel() for p in m.parameters()) == 34513
    assert len(sd) == 238
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'rollout_mnist64_h16.npz'))
    ref_keys = [k[2:] for k in g.files if k.startswith('w/')]
    assert list(sd.keys()) == ref_keys                      # same names AND same order as the reference
    for k in ref_keys:
        assert tuple(sd[k].shape) == g['w/' + k].shape, k
    o = O.Seq2Seq(16, 0.1, 0.1, input_timesteps=10, input_features=4, output_timesteps=10, n_layers=2)
    assert set(o.state_dict().keys()) == set(ref_keys)
    assert sd['encoder.rnns.0.conv_x_i.convolutions.0.lins.0.weight'].shape == (16, 4)
    assert sd['encoder.rnns.0.w_c_i'].shape == (1, 16)


@pytest.mark.parametrize('n_conv', [1, 2, 3])
def test_chebconv_composition_algebra(n_conv):
    """compose_chebconvs (host, torch): one Chebyshev series == the oracle's sequential ChebConv stack,
    evaluated here with a dense L^ on the CPU."""
    from oracle import qt_oracle as O
    from qtmpnn.ops import compose_chebconvs
    torch.manual_seed(n_conv)
    labels = O.quadtree_decompose(np.random.default_rng(0).random((16, 16)).astype(np.float32), thresh=0.8, max_size=8)
    ei = torch.as_tensor(O.adjacency_sorted(labels))
    n = int(labels.max()) + 1
    ew = torch.rand(ei.shape[1]) + 0.5
    ew = (ew + ew[torch.argsort(torch.argsort(ei[1] * n + ei[0]))]) / 2        # symmetric weights
    key = {(int(a), int(b)): i for i, (a, b) in enumerate(ei.T)}
    ew = torch.stack([(ew[i] + ew[key[(int(b), int(a))]]) / 2 for i, (a, b) in enumerate(ei.T)])
    stack = O.GraphConv('ChebConv', 5, 6, n_conv)
    for p in stack.parameters():
        p.data.normal_(0, 0.4)
    x = torch.randn(n, 5)
    ref = stack(x, ei, ew)
    keep = ei[0] != ei[1]
    W = torch.zeros(n, n, dtype=torch.float64)
    W[ei[1][keep], ei[0][keep]] = ew[keep].double()
    deg = W.sum(0)
    dis = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))
    L = -(dis[:, None] * W * dis[None, :])
    Ws = [torch.stack([torch.stack([lin.weight.t() for lin in c.lins])]) for c in stack.convolutions]
    bs = [c.bias.unsqueeze(0) for c in stack.convolutions]
    P, beta = compose_chebconvs(Ws, bs)
    assert P.shape[1] == 2 * n_conv + 1 and beta.shape[1] == 2 * (n_conv - 1) + 1
    T = [x.double(), L @ x.double()]
    ones = [torch.ones(n, 1, dtype=torch.float64), L @ torch.ones(n, 1, dtype=torch.float64)]
    for _ in range(2, P.shape[1]):
        T.append(2 * L @ T[-1] - T[-2])
        ones.append(2 * L @ ones[-1] - ones[-2])
    out = sum(T[k] @ P[0, k].double() for k in range(P.shape[1])) + \
        sum(ones[k] @ beta[0, k].double().unsqueeze(0) for k in range(beta.shape[1]))
    np.testing.assert_allclose(out.detach().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-4)


def test_gconvlstm_packing_shapes():
    from model.model import GConvLSTM
    cell = GConvLSTM(4, 16, n_conv_layers=2, convolution_type='ChebConv')
    no_h, with_h = cell.pack(4, None, (False, True))
    assert (with_h.K, with_h.Ks) == (5, 3) and with_h.W.shape == (5 * 20 + 4, 64)          # 3 bias rows padded to 4
    assert no_h.W.shape == (5 * 4 + 4, 64) and no_h.wc is with_h.wc and no_h.acc_p is with_h.acc_p
    assert with_h.wc.shape == (3, 16) and with_h.b.shape == (4, 16)
    assert not with_h.W[-1].any()                                                           # the pad row is zero
    assert cell.pack(8, None, (False,))[0].W.shape == (5 * 8 + 4, 64)
    cell1 = GConvLSTM(16, 16, n_conv_layers=1, convolution_type='ChebConv')
    pk = cell1.pack(None, None, (True,))[0]
    assert (pk.K, pk.Ks) == (3, 1) and pk.W.shape == (3 * 32 + 4, 64)
    # gate order i, f, c, o along the output axis; rows = [k][x channels | h channels]
    lin = cell1.conv_x_c.convolutions[0].lins[1].weight
    assert torch.equal(pk.W[32:48, 32:48], lin.t())
    assert torch.equal(pk.W[96, :16], cell1.conv_x_i.convolutions[0].bias + cell1.conv_h_i.convolutions[0].bias)


def test_synthetic_clips_are_deterministic_and_shaped():
    from qtmpnn import synthetic
    a = synthetic.make_clip(5, n_digits=2, n_frames=20)
    b = synthetic.make_clip(5, n_digits=2, n_frames=20)
    assert a.shape == (20, 64, 64, 1) and a.dtype == np.float32 and np.array_equal(a, b)
    x, y = synthetic.make_batch(2, 0, 3, 10, 10, n_digits=2)
    assert x.shape == (3, 10, 64, 64, 1) and y.shape == (3, 10, 64, 64, 1)
    clean = synthetic.make_clip(5, n_frames=4, pixel_noise=0.0)
    assert 0.02 < (clean > 0.1).mean() < 0.3 and clean.max() <= 1.0 and clean.min() == 0.0


def test_positional_encoding_matches_oracle():
    from model.utils import add_positional_encoding
    from oracle import qt_oracle as O
    x = torch.randn(3, 12, 20, 2)
    assert torch.equal(add_positional_encoding(x), O.add_positional_encoding(x))
    xn = add_positional_encoding(x.numpy())
    np.testing.assert_array_equal(xn, O.add_positional_encoding(x).numpy())


def test_unsupported_variants_raise_clearly():
    from model.model import GConvGRU
    from model.seq2seq import Seq2Seq
    with pytest.raises(NotImplementedError):
        GConvGRU(4, 4)
    with pytest.raises(NotImplementedError):
        Seq2Seq(16, 0.1, 0.1, convolution_type='GATConv')
    with pytest.raises(AssertionError):
        Seq2Seq(16, 0.1, 0.1, convolution_type='NoSuchConv')
    for h, conv in ((12, 'ChebConv'), (4, 'GCNConv'), (64, 'TransformerConv'), (20, 'TransformerConv')):
        with pytest.raises(ValueError, match='hidden_size'):            # (said at construction, not by the first launch)
            Seq2Seq(h, 0.1, 0.1, convolution_type=conv)
    for h, conv in ((64, 'ChebConv'), (128, 'GCNConv'), (8, 'TransformerConv')):
        Seq2Seq(h, 0.1, 0.1, convolution_type=conv, n_layers=1, n_conv_layers=1)
    with pytest.raises(ValueError, match='gate'):           # 5 hops x (4 + 128) channels: more rows than the gate GEMM takes
        Seq2Seq(128, 0.1, 0.1, convolution_type='ChebConv', n_layers=1, n_conv_layers=2)
    with pytest.raises(ValueError, match='gate'):           # upper layers see [H below | H]: 5 x 128
        Seq2Seq(64, 0.1, 0.1, convolution_type='ChebConv', n_layers=2, n_conv_layers=2)
    Seq2Seq(32, 0.1, 0.1, convolution_type='ChebConv', n_layers=4, n_conv_layers=3)        # (7 x 64 + 8 = 456)


@pytest.mark.parametrize('n_conv', [1, 2])
def test_pack_plan_equals_per_tensor_packing(n_conv):
    """ops.PackPlan (one gather of all parameters through recorded index maps) == the per-tensor stack / cat packing,
    values and every parameter gradient; also the ChebConv head matrix with padding."""
    import torch
    from model.model import ChebConv, GConvLSTM
    from qtmpnn import ops
    torch.manual_seed(0)
    cell = GConvLSTM(4, 16, n_conv, 'ChebConv')
    for p in cell.parameters():
        p.data.normal_()
    ref = cell.pack(4, None, (False, True))
    params = cell.plan_params()
    plan = ops.PackPlan(params, lambda T, fill: cell.plan_layout(T, fill, 'c.', 4, (False, True)))
    got = cell.pack_from(plan(), 'c.', 4, None, (False, True))
    for r, g in zip(ref, got):
        assert (r.K, r.Ks) == (g.K, g.Ks) and torch.allclose(r.W, g.W, atol=1e-6)
        assert torch.equal(r.wc, g.wc) and torch.equal(r.b, g.b)

    def probe(cells):
        return sum((c.W * torch.linspace(-1, 1, c.W.numel()).view_as(c.W)).sum() for c in cells) + 2 * cells[0].wc.sum() + 3 * cells[0].b.sum()
    gr = torch.autograd.grad(probe(ref), params, allow_unused=True)
    gg = torch.autograd.grad(probe(got), params, allow_unused=True)
    for a, b in zip(gr, gg):
        assert (a is None and b is None) or torch.allclose(a, b, atol=1e-5)
    # gradients handed out by the plan are disjoint views (clip_grad_norm_ scales them in place)
    ptrs = sorted((g.data_ptr(), g.numel() * 4) for g in gg if g is not None)
    assert all(p0 + n0 <= p1 for (p0, n0), (p1, _) in zip(ptrs, ptrs[1:]))
    conv = ChebConv(17, 16)
    for p in conv.parameters():
        p.data.normal_()
    plan = ops.PackPlan(conv.plan_params(), lambda T, fill: {'w': conv.plan_layout(T, fill, 20, 16)})
    assert torch.equal(plan()['w'], conv.packed(20, 16))


def test_pack_plan_of_attention_models_equals_per_tensor_packing():
    """TransformerConv cells and the attention decoder head through ops.PackPlan: the layer-by-layer matrices of
    GConvLSTM._pack_multi / TransformerConv.pack_many bit for bit, and every parameter gradient (hidden 8 -> padded planes,
    6 input channels -> padded rows)."""
    import torch
    from model.model import GConvLSTM, TransformerConv
    from model.seq2seq import Decoder, Encoder
    from qtmpnn import ops
    torch.manual_seed(0)
    cell = GConvLSTM(6, 8, n_conv_layers=3, convolution_type='TransformerConv')
    for p in cell.parameters():
        p.data.normal_()
    assert cell.plannable
    names = [f'{br}_{g}' for br in ('conv_x', 'conv_h') for g in cell.GATES]
    ref = cell._pack_multi(names)
    params = cell.plan_params()
    assert len(params) == len(list(cell.parameters()))
    plan = ops.PackPlan(params, lambda T, fill: cell.plan_layout(T, fill, 'r.'))
    first, cont = cell.pack_from(plan(), 'r.', None, None, (False, True))
    assert first.multi is cont.multi and first.W is None
    mats = lambda multi: [w for Ws, We, _ in multi for w in Ws + [We]]
    for a, b in zip(mats(ref), mats(first.multi)):
        assert a.shape == b.shape and torch.equal(a, b)
    assert torch.equal(first.wc, torch.cat([cell.w_c_i, cell.w_c_f, cell.w_c_o])) and torch.equal(first.b, torch.cat([cell.b_i, cell.b_f, cell.b_c, cell.b_o]))
    probe = lambda ms: sum((m * torch.linspace(-1, 1, m.numel()).view_as(m)).sum() for m in ms)
    gr = torch.autograd.grad(probe(mats(ref)), params, allow_unused=True)
    gg = torch.autograd.grad(probe(mats(first.multi)), params, allow_unused=True)
    for a, b, prm in zip(gr, gg, params):
        assert (a is None and not b.any()) or torch.allclose(a, b, atol=1e-6), prm.shape
    dec = Decoder(4, 8, 0.1, n_layers=2, concat_layers_dim=1, convolution_type='TransformerConv', n_conv_layers=2)
    for p in dec.parameters():
        p.data.normal_()
    assert dec.plannable and Encoder(6, 8, 0.1, n_layers=2, convolution_type='TransformerConv', n_conv_layers=2).plannable
    pk = dec.pack(4)
    assert pk['fc1'] is None and pk['fc2'] is None
    for a, b in zip(TransformerConv.pack_many([dec.fc_out1, dec.fc_out2]), pk['heads']):
        assert torch.equal(a.W, b.W) and torch.equal(a.We, b.We)


def test_model_with_cached_plans_pickles_and_copies():
    """The per-module caches (packing plans with their closures, flat-buffer bookkeeping, remembered parameter list) stay out of
    pickle / deepcopy: torch.save(model) and copy.deepcopy(model) work after a forward pass has built them, and the copy builds its own."""
    import copy, pickle, torch
    from model.seq2seq import Seq2Seq
    from qtmpnn.flat import param_list
    m = Seq2Seq(16, 0.1, 0.1, input_timesteps=3, input_features=1, output_timesteps=2, n_layers=1)
    m._packs(4)
    assert '_plans' in m.__dict__ and '_param_slots' in m.__dict__
    m2 = pickle.loads(pickle.dumps(m))
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    assert not {'_plans', '_flat_params', '_param_slots'} & set(m2.__dict__)
    m3 = copy.deepcopy(m)
    m3._packs(4)
    assert all(a is not b for a, b in zip(param_list(m), param_list(m3)))
    # a re-assigned Parameter is seen by the remembered list
    old = m.encoder.rnns[0].b_i
    m.encoder.rnns[0].b_i = torch.nn.Parameter(torch.zeros(1, 16))
    ps = param_list(m)
    assert any(p is m.encoder.rnns[0].b_i for p in ps) and not any(p is old for p in ps)
    assert all(a is b for a, b in zip(ps, m.parameters()))


def test_flat_params_views_and_single_tensor_update_equal_per_tensor_update():
    """qtmpnn.flat.FlatParams (CPU, plain torch): every parameter becomes a view of one buffer without changing values, keys
    or shapes; load_state_dict keeps the views; clip_grad_norm_ + Adam on the ONE flat tensor give the same weights as the
    reference's per-tensor clip + Adam (mpnnlstm.py:251-257)."""
    import copy
    import torch
    from qtmpnn.flat import FlatParams, flat_params
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.LayerNorm(7), torch.nn.Linear(7, 3))
    ref = copy.deepcopy(net)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    fp = flat_params(net)
    assert isinstance(fp, FlatParams) and fp.intact(net) and flat_params(net) is fp
    assert fp.n == sum(p.numel() for p in net.parameters()) and float(fp.buffer[fp.n:].abs().sum()) == 0.0
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k])
    net.load_state_dict({k: v + 1.0 for k, v in before.items()})
    assert fp.intact(net) and torch.equal(fp.param.detach()[:35], (before['0.weight'] + 1.0).reshape(-1))
    net.load_state_dict(before)
    x = torch.randn(11, 5)
    opt_f = torch.optim.Adam([fp.param], lr=0.05)
    opt_r = torch.optim.Adam(ref.parameters(), lr=0.05)
    for _ in range(3):
        fp.zero_grad()
        (net(x) ** 2).sum().backward()
        assert fp.grad_vector() is None                # plain autograd gradients are separate tensors: gathered by copy
        fp.param.grad = fp.gather_grads()
        torch.nn.utils.clip_grad_norm_([fp.param], 1.0)
        opt_f.step()
        opt_r.zero_grad()
        (ref(x) ** 2).sum().backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        opt_r.step()
    for (k, a), b in zip(net.state_dict().items(), ref.state_dict().values()):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), k
    moved = copy.deepcopy(net)                          # a copy owns separate tensors: its own FlatParams is made on demand
    assert not fp.intact(moved) and flat_params(moved) is not fp


def test_library_issues_only_kernels_on_the_callers_stream():
    """Every device write of the C-ABI library is a kernel launched on the stream the caller passes: no hipMemset / hipMemcpy
    (sync or async), no stream or event of its own, no allocation.  That is what makes every entry capturable into the caller's
    hipGraph; round 1 lost a run to a memset issued inside a captured step by code that was never committed (DESIGN.md
    section 10 records what a controlled experiment showed about memset nodes themselves)."""
    import glob
    src = ''
    for f in glob.glob(os.path.join(ROOT, 'quadtree-mpnnlstm_amd', 'csrc', '*.h*')):
        src += re.sub(r'//[^\n]*', '', open(f).read())
    for banned in ('hipMemset', 'hipMemcpy', 'hipMalloc', 'hipFree', 'hipStreamCreate', 'hipEventCreate', 'hipDeviceSynchronize',
                   'hipStreamSynchronize'):
        assert banned not in src, f'{banned} found in csrc/: device work must be kernels on the caller stream'


def test_library_keeps_no_mutable_global_state():
    """SURVEY 8(b) / INTEGRATION.md section 2: "no global state, thread-safe per stream" -- the only file-scope variable of
    the library is the thread-local error string; tuning knobs are arguments (round 3 had a process-wide slice-width switch,
    `qt_cheb_clip_width`; it is the `width` argument of qt_cheb_clip_fwd / _bwd now).  Diagnostics builds (`#ifdef
    QT_*_TIMING` blocks: in-kernel time stamps, never part of the shipped library) are exempt."""
    import glob
    decl = re.compile(r'^(?:static\s+)?(?:thread_local\s+)?(?:unsigned\s+|long\s+)*[A-Za-z_][\w:<>]*[\s\*&]+\**\s*(g_\w+|\w+)\s*(?:\[[^\]]*\])?\s*(?:=[^;(]*)?;\s*$')
    found = []
    for f in sorted(glob.glob(os.path.join(ROOT, 'quadtree-mpnnlstm_amd', 'csrc', '*.h*'))):
        text = re.sub(r'/\*.*?\*/', '', open(f).read(), flags=re.S)
        depth, timing, stack = 0, 0, []
        for line in text.split('\n'):
            code = re.sub(r'//.*', '', line).rstrip()
            st = code.strip()
            if st.startswith('#if'):
                stack.append(bool(re.search(r'QT_\w*TIMING', st)))
                timing += stack[-1]
            elif st.startswith('#endif') and stack:
                timing -= stack.pop()
            if st.startswith('#'):
                continue
            if depth == 0 and not timing and st and not st.startswith(('constexpr', 'static constexpr', 'const ', 'static const ',
                                                                       'using ', 'typedef', 'extern', 'template', 'namespace',
                                                                       'struct', '}', 'return')):
                m = decl.match(st)
                if m and '(' not in st.split('=')[0]:
                    found.append((os.path.basename(f), st))
            depth += code.count('{') - code.count('}')
    assert [s for _, s in found] == ['static thread_local char g_qt_err[512] = "";'], found
