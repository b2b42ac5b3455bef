"""Encoder / Decoder / Seq2Seq rollout of the reference's model/seq2seq.py, batched over B clips
and running on the GPU mesh pipeline (qtmpnn).  Constructor kwargs, forward() arguments, returned
(outputs, output_mappings) and state-dict keys follow the reference; `x` may carry a leading clip
axis (B, T_in, W, H, C) -- the reference processes one clip per call (ice_exp.py:137-139).
"""
import random

import numpy as np
import torch
import torch.nn as nn

from model.graph_functions import Graph, _criterion
from model.model import CONVOLUTION_KWARGS, GConvLSTM, _conv_class, _need_mesh
from qtmpnn import ops
from qtmpnn._lib import on_device
from qtmpnn.flat import flat_params, param_list
from qtmpnn.mesh import build_mesh, build_pixel_mesh, host_mask


_PROJECT_FC2 = __import__('os').environ.get('QT_NO_PROJECT_FC2') != '1'      # (A/B switch, diagnostics)


def _ln_params(*norms):
    return torch.stack([p for n in norms for p in (n.weight, n.bias)])


def _raise_on_nan(t, what):
    """image_to_graph's ValueError (graph_functions.py:626-627, 654-655) on the hot path's own tensors (one device sync)."""
    bad = torch.isnan(t)
    if bool(bad.any()):
        raise ValueError(f'Found NaNs in {what} data {int(bad.sum())} / {t.numel()}')


class _NoCachesInPickle:
    """copy / pickle / torch.save(model) see the module as the reference's: the per-module caches built on first use (packing plans
    with their closures, the flat parameter buffer's bookkeeping, the remembered parameter list) are dropped and rebuilt."""

    def __getstate__(self):
        d = self.__dict__.copy()
        for k in ('_plans', '_flat_params', '_param_slots'):
            d.pop(k, None)
        return d


class Encoder(_NoCachesInPickle, nn.Module):
    """model/seq2seq.py:21-82.  Layer 0 continues from (H, C); upper layers restart from zero state on
    every call (:71) and one LayerNorm pair is shared by all layers (:49-50) -- reproduced as is."""

    def __init__(self, input_features, hidden_size, dropout, n_layers=1, convolution_type='GCNConv', rnn_type='LSTM',
                 n_conv_layers=3, dummy=False):
        super().__init__()
        assert rnn_type in ['GRU', 'LSTM', 'SplitLSTM']
        if rnn_type != 'LSTM' or dummy:
            raise NotImplementedError('only rnn_type="LSTM", dummy=False is on the HIP path')
        self.rnn_type, self.hidden_size, self.n_layers, self.dummy = rnn_type, hidden_size, n_layers, dummy
        dims = [input_features] + [hidden_size] * n_layers
        self.rnns = nn.ModuleList([GConvLSTM(d, hidden_size, n_conv_layers, convolution_type, name='encoder')
                                   for d in dims[:-1]])
        self.dropout = nn.Dropout(dropout)        # constructed but never applied, like the reference (:47)
        self.norm_h = nn.LayerNorm(hidden_size)
        self.norm_c = nn.LayerNorm(hidden_size)

    @property
    def plannable(self):
        return all(r.plannable for r in self.rnns)

    def pack(self, in_pad):
        """Per-forward weight packing: layer 0 with and without a hidden state, upper layers without."""
        if self.plannable:
            return self._pack_planned(in_pad)
        ln = _ln_params(self.norm_h, self.norm_c)
        first, cont = self.rnns[0].pack(in_pad, ln, (False, True))
        return dict(first=first, cont=cont, upper=[r.pack(None, ln, (False,))[0] for r in self.rnns[1:]])

    def plan_spec(self, in_pad):
        """(params, layout, finish) of the encoder's weight packing as ONE parameter gather (ops.PackPlan): `layout` maps
        stand-ins of the parameters to the packed matrices, `finish(outs)` turns the gathered matrices into the pack."""
        params = [p for r in self.rnns for p in r.plan_params()] + [self.norm_h.weight, self.norm_h.bias,
                                                                    self.norm_c.weight, self.norm_c.bias]
        counts = [len(r.plan_params()) for r in self.rnns]

        def layout(T, fill):
            out, o = {}, 0
            for i, (r, n) in enumerate(zip(self.rnns, counts)):
                out.update(r.plan_layout(T[o:o + n], fill, f'r{i}.', in_pad if i == 0 else None,
                                         (False, True) if i == 0 else (False,)))
                o += n
            out['ln'] = torch.stack(T[o:o + 4])
            return out

        def finish(outs):
            ln = outs['ln']
            first, cont = self.rnns[0].pack_from(outs, 'r0.', in_pad, ln, (False, True))
            return dict(first=first, cont=cont,
                        upper=[r.pack_from(outs, f'r{i + 1}.', None, ln, (False,))[0] for i, r in enumerate(self.rnns[1:])])
        return params, layout, finish

    def _pack_planned(self, in_pad):
        """The same through ONE parameter gather for the whole encoder (ops.PackPlan, cached per input width)."""
        plans = self.__dict__.setdefault('_plans', {})
        key = (in_pad, self.norm_h.weight.device)
        params, layout, finish = self.plan_spec(in_pad)
        if key not in plans or not plans[key].same_params(params):       # (a re-assigned Parameter invalidates the plan)
            plans[key] = ops.PackPlan(params, layout)
        return finish(plans[key]())

    def run(self, X, mesh, H, C, pk):
        """One encoder step on packed weights; returns per-layer lists (no stacking on the hot path)."""
        _, h, c = self.rnns[0].step(X, mesh, H, C, pk['cont'] if H is not None else pk['first'])
        hs, cs = [h], [c]
        for rnn, w in zip(self.rnns[1:], pk['upper']):
            _, h, c = rnn.step(hs[-1], mesh, None, None, w)
            hs.append(h)
            cs.append(c)
        return hs, cs

    def forward(self, X, edge_index, edge_weight=None, H=None, C=None, packed=None):
        X = X.squeeze(0) if X.dim() == 3 else X
        pad = (-X.shape[1]) % 4
        if pad:
            X = nn.functional.pad(X, (0, pad))
        _need_mesh(edge_index, X, H, C)          # (H, C: the first layer's state, model/seq2seq.py:64-66)
        hs, cs = self.run(X, edge_index, H, C, packed if packed is not None else self.pack(X.shape[1]))
        return torch.stack(hs), torch.stack(cs)


class Decoder(_NoCachesInPickle, nn.Module):
    """model/seq2seq.py:84-187: n_layers GConvLSTM (always one conv layer, :106) carrying (H[i], C[i]),
    then norm_o + relu on the top layer's OUTPUT GATE, concat, fc_out1 -> relu -> fc_out2 -> dropout ->
    tanh -> + X[:, [0]]."""

    def __init__(self, input_features, hidden_size, dropout, n_layers=1, concat_layers_dim=3, convolution_type='GCNConv',
                 rnn_type='LSTM', n_conv_layers=3, binary=False, dummy=False):
        super().__init__()
        assert rnn_type in ['GRU', 'LSTM', 'SplitLSTM']
        if rnn_type != 'LSTM' or dummy:
            raise NotImplementedError('only rnn_type="LSTM", dummy=False is on the HIP path')
        self.rnn_type, self.input_features, self.hidden_size = rnn_type, input_features, hidden_size
        self.n_layers, self.binary, self.dummy, self.concat_layers_dim = n_layers, binary, dummy, concat_layers_dim
        dims = [input_features] + [hidden_size] * n_layers
        self.rnns = nn.ModuleList([GConvLSTM(d, hidden_size, 1, convolution_type, name='decoder') for d in dims[:-1]])
        cls, kw = _conv_class(convolution_type), CONVOLUTION_KWARGS[convolution_type]
        self.fc_out1 = cls(in_channels=hidden_size + concat_layers_dim, out_channels=hidden_size, **kw)
        self.fc_out2 = cls(in_channels=hidden_size, out_channels=1, **kw)
        self.norm_o = nn.LayerNorm(hidden_size)
        self.norm_h = nn.LayerNorm(hidden_size)
        self.norm_c = nn.LayerNorm(hidden_size)
        self.dropout = nn.Dropout(dropout)

    @property
    def head_width(self):
        return self.hidden_size + 4          # [relu(norm_o(O)) | concat | 0 0 0] keeps rows 16-byte aligned

    def pack(self, in_pad):
        if self.plannable:
            return self._pack_planned(in_pad)
        ln = _ln_params(self.norm_h, self.norm_c)
        series = hasattr(self.fc_out1, 'packed')
        heads = None
        if hasattr(type(self.fc_out1), 'pack_many'):        # attention head: both convolutions packed once per pass, not per step
            heads = type(self.fc_out1).pack_many([self.fc_out1, self.fc_out2])
        return dict(ln_o=_ln_params(self.norm_o), acc_o=ops.GradAcc(), heads=heads,
                    rnns=[r.pack(in_pad if i == 0 else None, ln, (True,))[0] for i, r in enumerate(self.rnns)],
                    fc1=self.fc_out1.packed(self.head_width, self.hidden_size) if series else None, acc1=ops.GradAcc(),
                    fc2=self.fc_out2.packed(self.hidden_size, 4) if series else None, acc2=ops.GradAcc())

    @property
    def plannable(self):
        from model.model import ChebConv, TransformerConv
        return all(r.plannable for r in self.rnns) and any(type(self.fc_out1) is cls and type(self.fc_out2) is cls
                                                           for cls in (ChebConv, TransformerConv))

    def plan_spec(self, in_pad):
        """(params, layout, finish) of the decoder's weight packing as one parameter gather (see Encoder.plan_spec)."""
        params = ([p for r in self.rnns for p in r.plan_params()] + self.fc_out1.plan_params() + self.fc_out2.plan_params()
                  + [self.norm_h.weight, self.norm_h.bias, self.norm_c.weight, self.norm_c.bias,
                     self.norm_o.weight, self.norm_o.bias])
        counts = [len(r.plan_params()) for r in self.rnns]
        n1, n2 = len(self.fc_out1.plan_params()), len(self.fc_out2.plan_params())

        def layout(T, fill):
            out, o = {}, 0
            for i, (r, n) in enumerate(zip(self.rnns, counts)):
                out.update(r.plan_layout(T[o:o + n], fill, f'r{i}.', in_pad if i == 0 else None, (True,)))
                o += n
            if hasattr(type(self.fc_out1), 'proj_layout'):          # attention head: the two convolutions' [q | k | v | skip] matrices
                for nm, conv, n in (('h1', self.fc_out1, n1), ('h2', self.fc_out2, n2)):
                    W, We = type(conv).proj_layout([T[o:o + n]], conv.in_channels, conv.out_channels, fill)
                    out[nm + 'W'], out[nm + 'E'] = W[0], We[0]
                    o += n
                out['ln'] = torch.stack(T[o:o + 4])
                out['ln_o'] = torch.stack(T[o + 4:o + 6])
                return out
            out['fc1'] = self.fc_out1.plan_layout(T[o:o + n1], fill, self.head_width, self.hidden_size)
            o += n1
            if _PROJECT_FC2 and self.fc_out2.K == 3:
                out['fc2c'] = self.fc_out2.plan_layout_projected(T[o:o + n2], fill)
            else:
                out['fc2'] = self.fc_out2.plan_layout(T[o:o + n2], fill, self.hidden_size, 4)
            o += n2
            out['ln'] = torch.stack(T[o:o + 4])
            out['ln_o'] = torch.stack(T[o + 4:o + 6])
            return out

        def finish(outs):
            ln = outs['ln']
            heads = None
            if 'h1W' in outs:
                from model.model import PackedConv
                heads = [PackedConv(outs[nm + 'W'], outs[nm + 'E'], ops.GradAcc(), ops.GradAcc()) for nm in ('h1', 'h2')]
            return dict(ln_o=outs['ln_o'], acc_o=ops.GradAcc(), heads=heads,
                        rnns=[r.pack_from(outs, f'r{i}.', in_pad if i == 0 else None, ln, (True,))[0]
                              for i, r in enumerate(self.rnns)],
                        fc1=outs.get('fc1'), acc1=ops.GradAcc(), fc2=outs.get('fc2'), fc2c=outs.get('fc2c'), acc2=ops.GradAcc())
        return params, layout, finish

    def _pack_planned(self, in_pad):
        """The same through ONE parameter gather for the whole decoder (ops.PackPlan, cached per input width)."""
        plans = self.__dict__.setdefault('_plans', {})
        key = (in_pad, self.norm_h.weight.device)
        params, layout, finish = self.plan_spec(in_pad)
        if key not in plans or not plans[key].same_params(params):
            plans[key] = ops.PackPlan(params, layout)
        return finish(plans[key]())

    def dropout_masks(self, steps, rows, device):
        """Inverted-dropout multipliers for `steps` decoder steps at once (one RNG launch instead of one per step);
        None in eval mode.  rows = an upper bound of the node count (B*W*H); a step uses the first N of its row."""
        if not (self.training and self.dropout.p > 0):
            return None
        keep = 1.0 - self.dropout.p
        return (torch.rand(steps, rows, device=device) < keep).float() / keep

    def run(self, X, mesh, concat_layers, H, C, pk, drop=None):
        """One decoder step on packed weights; X (N, 4k) padded input, H / C per-layer lists.  Returns (y, hs, cs, state_y):
        state_y is the output once more (same values, same storage, a second autograd alias) for the state update of the
        rollout -- the head's backward launch then sums the two gradients -- or None when there is no such alias (the caller
        uses y itself)."""
        assert self.concat_layers_dim == 1
        state_y = None
        hs, cs, inp, Xres = [], [], X, X
        last = len(self.rnns) - 1
        for i, rnn in enumerate(self.rnns):
            # H' of a lower layer has two consumers (the next layer now, this layer at the next step) and so has X (layer 0 and
            # the head's residual): the cell hands both out a second time and sums their gradients inside its backward launch
            res = rnn.step(inp, mesh, H[i], C[i], pk['rnns'][i], alias_h=i < last, pass_x=i == 0)
            out, h, c = res[:3]
            hs.append(h)
            cs.append(c)
            inp = res[3] if i < last else h
            if i == 0:
                Xres = res[-1]
        if concat_layers is None:
            # beyond-reference: HEAD crashes here (fc_out1 expects hidden+1 channels, seq2seq.py:115,164);
            # the decoder's current input value is used as the 1-channel concat (what :471,484 intended)
            concat_layers = X[:, :1]
        z = ops.head_input(out, pk['ln_o'], concat_layers, self.head_width, mesh, pk['acc_o'])    # (N, h), (N, 4)
        if drop is None and self.training and self.dropout.p > 0:
            drop = self.dropout_masks(1, z[0].shape[0], z[0].device)[0]
        if pk['fc1'] is None:           # attention head (TransformerConv): activations as plain tensor ops
            p1, p2 = pk.get('heads') or (None, None)
            y = self.fc_out2(torch.relu(self.fc_out1(torch.cat(z, dim=1), mesh, packed=p1)), mesh, packed=p2)
            y = torch.tanh(y if drop is None else y * drop.unsqueeze(1)) + X[:, :1]
            return (torch.sigmoid(y) if self.binary else y), hs, cs, None
        if pk.get('fc2c') is not None:
            # fc_out2 has ONE output channel: its three coefficient columns are applied first (a 16 -> 4 product, in the epilogue
            # of fc_out1's launch) and the Chebyshev recurrence then runs on single columns -- 4 bytes per row and neighbour
            # instead of z's 64-byte rows;  U (N, 4) = z [w_0 w_1 w_2 0] + [b 0 0 0]
            z, U = ops.cheb_poly(z, pk['fc1'], mesh, self.fc_out1.K, 1, ops.ACT_RELU, acc=pk['acc1'], post=(pk['fc2c'], pk['acc2']))
            # (the output has two consumers -- the loss, and the next step's input through the re-mesh: the second one takes the
            # alias `state_y`, so that the two gradients are summed inside the head's backward launch)
            Y, Y2 = ops.scalar_cheb3(U, Xres, drop, mesh, alias=True)
            y = Y[:, :1]
            if not self.binary:
                state_y = Y2[:, :1]
        else:
            z = ops.cheb_poly(z, pk['fc1'], mesh, self.fc_out1.K, 1, ops.ACT_RELU, acc=pk['acc1'])
            # the head GEMM writes float4 rows; column 0 is the prediction (the rest is tanh(0) + X[:, 0], never read as data)
            y = ops.cheb_poly(z, pk['fc2'], mesh, self.fc_out2.K, 1, ops.ACT_TANH_RES, res=Xres, drop=drop, acc=pk['acc2'])[:, :1]
        if self.binary:
            y = torch.sigmoid(y)
        return y, hs, cs, state_y

    def forward(self, X, edge_index, edge_weight, concat_layers, H, C, packed=None):
        pad = (-X.shape[1]) % 4
        Xp = nn.functional.pad(X, (0, pad)) if pad else X
        _need_mesh(edge_index, Xp, concat_layers, *H, *C)
        y, hs, cs, _ = self.run(Xp, edge_index, concat_layers, H, C, packed if packed is not None else self.pack(Xp.shape[1]))
        return y, torch.stack(hs), torch.stack(cs)


class Seq2Seq(_NoCachesInPickle, nn.Module):
    """model/seq2seq.py:190-527 for the quadtree path (finite `thresh`)."""

    def __init__(self, hidden_size, dropout, thresh, input_timesteps=3, input_features=4, output_timesteps=5, n_layers=4,
                 n_conv_layers=2, transform_func=None, condition='max_larger_than', remesh_input=False,
                 convolution_type='ChebConv', rnn_type='LSTM', binary=False, dummy=False, device=None, debug=False):
        super().__init__()
        # a node's row is spread over hidden / 4 lanes of a 64-lane wave (float4 each): the cell kernels are built for the powers of
        # two 8 .. 128, the attention kernels for 8, 16 and 32; said here, not by the first launch
        sizes = (8, 16, 32) if convolution_type == 'TransformerConv' else (8, 16, 32, 64, 128)
        if hidden_size not in sizes:
            raise ValueError(f'hidden_size={hidden_size}: the HIP kernels for convolution_type={convolution_type!r} are built for '
                             f'hidden sizes {sizes} (the reference scripts use 16 and 32)')
        if convolution_type in ('ChebConv', 'GCNConv'):
            # the stacked convolutions of a cell are composed into ONE Chebyshev series over [X | H]: its gate GEMM reduces over
            # (hops + 1) x (input + hidden channels) + bias rows, and the GEMM kernels take at most 512 (csrc/cheb.hip: MAXQ)
            kc = n_conv_layers * (2 if convolution_type == 'ChebConv' else 1) + 1
            pad4 = lambda v: v + (-v) % 4
            widths = [pad4(input_features) + hidden_size, 4 + hidden_size] + ([2 * hidden_size] if n_layers > 1 else [])
            red = kc * max(widths) + pad4(kc)
            if red > 512:
                raise ValueError(f'hidden_size={hidden_size} with n_conv_layers={n_conv_layers}, n_layers={n_layers}: the composed gate '
                                 f'matrix would have {red} rows, the GEMM kernels take 512 (fewer conv layers or a smaller hidden size)')
        self.encoder = Encoder(input_features, hidden_size, dropout, n_layers=n_layers, convolution_type=convolution_type,
                               rnn_type=rnn_type, n_conv_layers=n_conv_layers, dummy=dummy)
        self.decoder = Decoder(1 + 3, hidden_size, dropout, n_layers=n_layers, concat_layers_dim=1,
                               convolution_type=convolution_type, rnn_type=rnn_type, n_conv_layers=n_conv_layers,
                               binary=binary, dummy=dummy)
        self.input_timesteps, self.output_timesteps, self.n_layers = input_timesteps, output_timesteps, n_layers
        self.hidden_size, self.condition, self.remesh_input, self.debug = hidden_size, condition, remesh_input, debug
        self.convolution_type = convolution_type
        self.use_edge_attrs = convolution_type in ['MHTransformerConv', 'TransformerConv', 'GATConv']
        self.thresh, self.transform_func, self.graph, self.device = thresh, transform_func, None, device
        self.max_grid_size = 64                      # image_to_graph default, never overridden (graph_functions.py:590)
        self.static_shapes = False                   # True: worst-case capacities + device-side node counts (hipGraph)

    # -- weight packing -----------------------------------------------------------------
    def _packs(self, enc_in_pad):
        """(encoder pack, decoder pack) of this forward pass.  Plain ChebConv models pack BOTH through one parameter gather
        (ops.PackPlan over all parameters in module order): its backward hands every parameter a view of ONE gradient
        vector, so the trainer's all-reduce, clip and Adam run on a single flat tensor (qtmpnn.flat)."""
        if not (self.encoder.plannable and self.decoder.plannable):
            return self.encoder.pack(enc_in_pad), None          # the decoder packs itself when the rollout starts
        params = param_list(self)
        fp = flat_params(self) if params[0].is_cuda else None
        plans = self.__dict__.setdefault('_plans', {})
        key = (enc_in_pad, params[0].device)
        if key not in plans or not plans[key].same_params(params) or plans[key].flat is not fp:
            ep, el, ef = self.encoder.plan_spec(enc_in_pad)
            dp, dl, df = self.decoder.plan_spec(4)
            pos = {id(p): i for i, p in enumerate(params)}
            ei, di = [pos[id(p)] for p in ep], [pos[id(p)] for p in dp]

            def layout(T, fill):
                out = {'e.' + k: v for k, v in el([T[i] for i in ei], fill).items()}
                out.update({'d.' + k: v for k, v in dl([T[i] for i in di], fill).items()})
                return out
            plans[key] = ops.PackPlan(params, layout, flat=fp)
            plans[key].finish = (ef, df)        # (the closures depend on the modules and the input width only: kept with the plan)
        ef, df = plans[key].finish
        outs = plans[key]()
        return (ef({k[2:]: v for k, v in outs.items() if k.startswith('e.')}),
                df({k[2:]: v for k, v in outs.items() if k.startswith('d.')}))

    # -- mesh helpers -----------------------------------------------------------------
    def _mesh_from_image(self, img0, mask, hir, tiles=None):
        B, n, m = img0.shape
        if tiles is None:
            # the per-tile structures of the tile-resident recurrences (frames of several base cells) pay on the INPUT mesh
            # when the encoder's composed stacks have at least three hops (K = 2 n_conv + 1 >= 4); the decoder's cells are
            # single ChebConvs (K = 3, model/seq2seq.py:106): its re-meshes skip them
            conv = getattr(self.encoder.rnns[0], 'convolution_type', 'ChebConv')
            tiles = conv == 'ChebConv' and 2 * getattr(self.encoder.rnns[0], 'n_conv_layers', 1) + 1 >= ops._TILE_MIN_K
        return build_mesh(src=_criterion(img0, n, m, self.max_grid_size, self.transform_func), n=n, m=m,
                          thresh=self.thresh, condition=self.condition, mask=mask, high_interest_region=hir,
                          max_size=self.max_grid_size, static=self.static_shapes, tiles=tiles)

    def _mesh_from_nodes(self, out, mesh, mask, hir):
        if self.transform_func is not None:
            img0 = ops.gather_pixels(out.detach(), mesh).view(mesh.B, mesh.n, mesh.m)
            return self._mesh_from_image(img0, mask, hir, tiles=False)
        return build_mesh(prev=(out.detach()[:, 0], mesh), thresh=self.thresh, condition=self.condition, mask=mask,
                          high_interest_region=hir, max_size=self.max_grid_size, static=self.static_shapes, tiles=False)

    # -- encoder ------------------------------------------------------------------------
    @on_device(lambda self, *a, **k: self.encoder.norm_h.weight)
    def process_inputs(self, x, mask=None, high_interest_region=None, graph_structure=None):
        """model/seq2seq.py:254-336.  x: (T_in, W, H, C) or (B, T_in, W, H, C)."""
        self._single = x.dim() == 4
        self._deferred = None
        if self._single:
            x = x.unsqueeze(0)
        x = x.float()
        B, T, n, m, c = x.shape
        if c + 3 != self.encoder.rnns[0].in_channels:
            raise ValueError(f'the frames have {c} channels, the model was built for input_features = '
                             f'{self.encoder.rnns[0].in_channels - 3} (+ 2 position channels + the node size)')
        self.mask = mask
        if mask is not None and not (torch.is_tensor(mask) and mask.is_cuda) and host_mask(mask).all():
            raise ValueError('the mask covers every pixel: no node is left to predict')
        if not self.static_shapes:
            _raise_on_nan(x, 'image')       # image_to_graph's check (graph_functions.py:626-627); not inside a captured step, where
                                            # nothing may be read back: there a NaN ends as the reference's 'NaN loss :('
        if self.training:
            ops.advance_dropout_epoch(x.device)        # (a no-op unless an attention convolution with dropout ran before)
        if self.remesh_input:
            if graph_structure is not None or self.thresh == -np.inf:
                raise NotImplementedError('remesh_input=True is built for data-driven quadtree meshes only')
            return self._process_inputs_remesh(x, mask, high_interest_region)
        if graph_structure is not None:
            # preset static mesh (:288-294): node size feature = n_pixels_per_node / 4 ("Don't assume 4 !!" in the reference)
            mesh = graph_structure['mapping'].for_batch(B)
            if (mesh.n, mesh.m) != (n, m):
                raise ValueError(f'graph_structure was built for {mesh.n} x {mesh.m} frames, the input frames are {n} x {m}')
            size = (mesh.npix / 4.0).unsqueeze(1)
        elif self.thresh == -np.inf:
            mesh = build_pixel_mesh(B, n, m, mask, x.device)            # every unmasked pixel a node (:629-630)
            size = mesh.posfeat[:, 2:3]
        else:
            mesh = self._mesh_from_image(x[..., 0].amax(dim=1), mask, high_interest_region)
            size = mesh.posfeat[:, 2:3]
        fpad = (-(c + 3)) % 4
        if x.requires_grad or not x.is_cuda:
            means = ops.pool_image(x.reshape(B, T, n * m, c), mesh, True)                 # (T, N, c)
            nodefeat = torch.cat([mesh.posfeat[:, :2], size], dim=1)
            feats = torch.cat([means, nodefeat.unsqueeze(0).expand(T, -1, -1)], dim=-1)
            feats_in = nn.functional.pad(feats, (0, fpad)) if fpad else feats
        else:
            # rows [frame means | position | size | 0-pad] assembled in place: the means land in their columns, one
            # broadcast copy fills the rest (two concatenations and a pad of the (T, N, c+3) matrix otherwise)
            feats_in = x.new_empty(T, mesh.N, c + 3 + fpad, dtype=torch.float32)
            ops.pool_image_into(x.reshape(B, T, n * m, c), mesh, feats_in, 0, True)
            if size.data_ptr() == mesh.posfeat[:, 2:3].data_ptr():
                feats_in[:, :, c:c + 3] = mesh.posfeat
            else:
                feats_in[:, :, c:c + 2] = mesh.posfeat[:, :2]
                feats_in[:, :, c + 2:c + 3] = size
            if fpad:
                feats_in[:, :, c + 3:] = 0
            feats = feats_in[:, :, :c + 3]
        self.graph = Graph(None, None)
        self.graph.mapping, self.graph.n_pixels_per_node, self.graph.image_shape = mesh, mesh.npix, (n, m)
        enc_pack, self._dec_pack = self._packs(c + 3 + fpad)     # (the decoder pack waits for unroll_output)
        hidden = cell = None
        for t in range(self.input_timesteps):
            hidden, cell = self.encoder.run(feats_in[t], mesh, None if hidden is None else hidden[-1],
                                            None if cell is None else cell[-1], enc_pack)
        self.graph.hidden, self.graph.cell = hidden, cell            # per-layer lists while the rollout runs
        last = feats[-1]                                                                # x[-1, :, [0,-3,-2,-1]] (:336)
        self.graph.pyg.x = last if (c == 1 and last.is_contiguous()) else torch.cat([last[:, :1], last[:, -3:]], dim=1)

    def _process_inputs_remesh(self, x, mask, hir):
        """remesh_input=True (model/seq2seq.py:266-276, 312, 323-324, do_remesh_input :493-527): the mesh of encoder step t
        comes from input frame t alone, and after every step the state moves to the mesh of frame t + 1 -- also after the
        last one, so x must carry input_timesteps + 1 frames (with fewer the reference fails at x[[t + 1]]; so does this)."""
        B, T, n, m, c = x.shape
        if T <= self.input_timesteps:
            raise IndexError(f'index {self.input_timesteps} is out of bounds for dimension 0 with size {T} '
                             '(remesh_input=True reads frame t + 1 after every encoder step, model/seq2seq.py:324)')
        fpad = (-(c + 3)) % 4
        L, h = self.n_layers, self.hidden_size
        enc_pack, self._dec_pack = self._packs(c + 3 + fpad)

        def frame_rows(t, mesh):                 # [frame means | position | size | 0-pad] of frame t on `mesh`
            f = x.new_empty(1, mesh.N, c + 3 + fpad, dtype=torch.float32)
            ops.pool_image_into(x[:, t:t + 1].reshape(B, 1, n * m, c), mesh, f, 0, True)
            f[0, :, c:c + 3] = mesh.posfeat
            if fpad:
                f[0, :, c + 3:] = 0
            return f[0]
        mesh = self._mesh_from_image(x[:, 0, ..., 0], mask, hir)
        rows = frame_rows(0, mesh)
        hidden = cell = None
        for t in range(self.input_timesteps):
            hidden, cell = self.encoder.run(rows, mesh, None if hidden is None else hidden[-1],
                                            None if cell is None else cell[-1], enc_pack)
            new = self._mesh_from_image(x[:, t + 1, ..., 0], mask, hir)
            parts = ops.remesh_transfer([*hidden, *cell], mesh, new, [h] * (2 * L))
            hidden, cell = list(parts[:L]), list(parts[L:])
            mesh, rows = new, frame_rows(t + 1, new)
        self.graph = Graph(None, None)
        self.graph.mapping, self.graph.n_pixels_per_node, self.graph.image_shape = mesh, mesh.npix, (n, m)
        self.graph.hidden, self.graph.cell = hidden, cell
        self.graph.pyg.x = rows if (c == 1 and fpad == 0) else torch.cat([rows[:, :1], rows[:, c:c + 3]], dim=1)

    # -- decoder + remesh ----------------------------------------------------------------
    @on_device(lambda self, *a, **k: self.encoder.norm_h.weight)
    def unroll_output(self, unroll_steps, y, concat_layers=None, teacher_forcing_ratio=0.5, mask=None,
                      high_interest_region=None, remesh_every=1):
        """model/seq2seq.py:339-398.  concat_layers: (T_out, W, H, 1) or (B, T_out, W, H, 1)."""
        g = self.graph
        self._apply_deferred_update()
        mesh = g.mapping
        if concat_layers is not None:
            concat_layers = concat_layers.to(g.pyg.x.device).float()
            if concat_layers.dim() == 4:
                concat_layers = concat_layers.unsqueeze(0)
        if y is not None and y.dim() == 4:
            y = y.unsqueeze(0)
        # packed together with the encoder by this forward pass's process_inputs; taken once: nothing on `self` may keep
        # the autograd graph alive beyond the pass (a continued unroll packs the decoder on its own)
        dec_pack, self._dec_pack = getattr(self, '_dec_pack', None), None
        if dec_pack is None:
            dec_pack = self.decoder.pack(4)
        outputs, output_mappings = [], []
        steps = list(unroll_steps)
        drops = self.decoder.dropout_masks(len(steps), mesh.B * mesh.P, g.pyg.x.device)
        for si, t in enumerate(steps):
            concat_t = None
            if concat_layers is not None:
                cl = concat_layers[:, t].reshape(mesh.B, 1, mesh.P, 1)
                concat_t = ops.pool_image(cl, mesh, True)[0]
                g.concat_layers = concat_t
            output, hidden, cell, state_y = self.decoder.run(g.pyg.x, mesh, concat_t, g.hidden, g.cell, dec_pack,
                                                             None if drops is None else drops[si, :mesh.N])
            outputs.append(output)
            if state_y is not None:
                output = state_y                                                           # (the alias for the state update)
            output_mappings.append(mesh)
            teacher_force = random.random() < teacher_forcing_ratio
            if t == steps[-1]:
                # the reference updates the state once more here (:393-396: re-mesh or input update); nothing in a training
                # step reads it, so it is DEFERRED: a later unroll_output call without process_inputs in between (a
                # continued rollout) applies it first.  The kept state is detached: a live reference into the autograd
                # graph would pin its AccumulateGrad nodes (and their stream) across iterations, which breaks hipGraph
                # capture of the next step -- so a continued rollout does not backpropagate into the earlier call.
                g.hidden = [h.detach() for h in hidden]
                g.cell = [c.detach() for c in cell]
                g.pyg.x = g.pyg.x.detach()
                if concat_layers is not None:
                    g.concat_layers = g.concat_layers.detach()
                self._deferred = (output.detach(), (t + 1) % remesh_every == 0, mask, high_interest_region, teacher_force,
                                  y[:, t].detach() if teacher_force else None)
                break
            if self.thresh != -np.inf and (t + 1) % remesh_every == 0:
                mesh = self.do_remesh(output, hidden, cell, mask, high_interest_region, teacher_force,
                                      y[:, t] if teacher_force else None)
            else:
                self.update_without_remesh(output, hidden, cell, teacher_force, y[:, t] if teacher_force else None)
        return outputs, output_mappings

    def _apply_deferred_update(self):
        """The state update the previous unroll_output call left out after its last step (see there)."""
        d, self._deferred = getattr(self, '_deferred', None), None
        if d is None:
            return
        output, remesh, mask, hir, teacher_force, teacher_input = d
        g = self.graph
        if self.thresh != -np.inf and remesh:
            self.do_remesh(output, g.hidden, g.cell, mask, hir, teacher_force, teacher_input)
        else:
            self.update_without_remesh(output, g.hidden, g.cell, teacher_force, teacher_input)

    @on_device(lambda self, *a, **k: self.encoder.norm_h.weight)
    def forward(self, x, y=None, concat_layers=None, teacher_forcing_ratio=0.5, mask=None, high_interest_region=None,
                graph_structure=None, remesh_every=1):
        self.process_inputs(x, mask=mask, high_interest_region=high_interest_region, graph_structure=graph_structure)
        return self.unroll_output(range(self.output_timesteps), y, concat_layers=concat_layers,
                                  teacher_forcing_ratio=teacher_forcing_ratio, mask=mask,
                                  high_interest_region=high_interest_region, remesh_every=remesh_every)

    def update_without_remesh(self, data, hidden, cell, teacher_force=False, teacher_input=None):
        """model/seq2seq.py:420-431.  With teacher forcing the reference rebuilds the whole input row as
        [flatten(teacher + positional encoding) | RAW n_pixels_per_node] (:422-425): the size column is then the pixel count
        itself, not the normalised size the mesh build wrote (npix / 1024, resolution^2 on pixelwise meshes, npix / 4 on
        preset meshes) -- reproduced as is."""
        g = self.graph
        if teacher_force:
            mesh = g.mapping
            val = ops.pool_image(teacher_input.reshape(mesh.B, 1, mesh.P, -1)[..., :1].float(), mesh, True)[0]
            g.pyg.x = torch.cat([val, mesh.posfeat[:, :2], mesh.npix.unsqueeze(1)], dim=-1)
        else:
            g.pyg.x = torch.cat([data, g.pyg.x[:, 1:]], dim=-1)
        g.hidden, g.cell = hidden, cell

    def do_remesh(self, data, hidden, cell, mask=None, high_interest_region=None, teacher_force=False, teacher_input=None):
        """model/seq2seq.py:434-491: the new mesh is decided by the model's own output; output, hidden and cell
        move to it as per-cell means of their un-flattened images (one fused kernel, no image in memory)."""
        g = self.graph
        old = g.mapping
        L, h = len(hidden), hidden[0].shape[1]
        if teacher_force:
            img0 = teacher_input[..., 0].reshape(old.B, old.n, old.m).float()
            new = self._mesh_from_image(img0, mask, high_interest_region)
            val = ops.pool_image(img0.reshape(old.B, 1, old.P, 1), new, True)[0]
            parts = ops.remesh_transfer([*hidden, *cell], old, new, [h] * (2 * L))
        else:
            if not self.static_shapes:
                _raise_on_nan(data, 'image')        # (the reference un-flattens the output and image_to_graph checks the image)
            new = self._mesh_from_nodes(data, old, mask, high_interest_region)
            # rows stay float4-sized: the head's own 4-wide output when `data` is its column 0, else 4 copies
            b4 = data._base
            wide = (b4 is not None and b4.dim() == 2 and b4.shape == (data.shape[0], 4) and b4.is_contiguous()
                    and data.storage_offset() == b4.storage_offset() and data.stride(0) == 4)
            # the state goes across as its parts and comes back as column views of one matrix: nothing is concatenated
            # (on the tile-resident transfer the next decoder input [value | position, size] is assembled by the same launch)
            fold = ops._DEC_FOLD and ops.clip_remesh_ok(old, new)
            val4, *parts = ops.remesh_transfer([b4 if wide else data.expand(-1, 4).contiguous(), *hidden, *cell], old, new,
                                               [4] + [h] * (2 * L), dec_input=fold)
            val = None
        g.hidden, g.cell = list(parts[:L]), list(parts[L:])
        if val is None:
            g.pyg.x = val4 if fold else ops.decoder_input(val4, new)
        else:
            g.pyg.x = torch.cat([val, new.posfeat], dim=-1)
        g.mapping, g.n_pixels_per_node = new, new.npix
        return new
