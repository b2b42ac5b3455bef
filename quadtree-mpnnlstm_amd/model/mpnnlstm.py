"""Trainer facade of the reference's model/mpnnlstm.py (NextFramePredictorS2S) around the HIP rollout.

Same constructor / train / predict / save / load surface; the train step (mpnnlstm.py:229-257) is
`train_step` below and additionally accepts batches of clips and a torch.distributed process group
(one flat gradient all-reduce per step).  tensorboard is optional (absent -> no-op writer).
"""
import datetime
import os
import time
from abc import ABC, abstractmethod

import numpy as np
import pandas as pd
import torch
from torch.optim.lr_scheduler import StepLR

from model.graph_functions import image_to_graph, unflatten, plot_contours
from model.seq2seq import Seq2Seq
from model.utils import add_positional_encoding, get_n_params, int_to_datetime
from qtmpnn import ops
from qtmpnn.dist import all_reduce_sum, allreduce_gradients
from qtmpnn.flat import flat_params
from qtmpnn._lib import on_device
from qtmpnn.mesh import check_tile_errors, host_mask

try:                                        # pragma: no cover - optional dependency
    from torch.utils.tensorboard import SummaryWriter
except Exception:                           # tensorboard is not installed in the build image
    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        def flush(self):
            pass


def masked_mse(outputs, meshes, y, mask=None, binary=False):
    """MSELoss(y_hat[:, ~mask], y[:, ~mask]) of mpnnlstm.py:243-246 without building y_hat:
    sum over steps of the per-mesh squared error, divided by (clips x steps x unmasked pixels).
    y: (T_out, W, H, 1) or (B, T_out, W, H, 1)."""
    if y.dim() == 4:
        y = y.unsqueeze(0)
    mesh0 = meshes[0]
    want = (mesh0.B, len(outputs), mesh0.n, mesh0.m, 1)
    if tuple(y.shape) != want:      # (the loss kernels read B x P targets per step straight from this buffer)
        raise ValueError(f'targets of shape {tuple(y.shape)} for {mesh0.B} clip(s) x {len(outputs)} output steps of {mesh0.n} x {mesh0.m} '
                         f'frames: expected (T_out, W, H, 1) or (B, T_out, W, H, 1) = {want}')
    mask = host_mask(mask)
    n_valid = mesh0.P if mask is None else int((~mask).sum())
    if binary:
        y_hat = torch.stack([unflatten(o, ms, (ms.n, ms.m)).reshape(ms.B, ms.n, ms.m, 1) for o, ms in zip(outputs, meshes)], 1)
        keep = torch.ones(mesh0.n, mesh0.m, dtype=torch.bool) if mask is None else ~torch.as_tensor(mask)
        return torch.nn.functional.binary_cross_entropy(y_hat[:, :, keep], y.to(y_hat.device)[:, :, keep])
    y = y.to(outputs[0].device)
    part = ops.rollout_sse_partials(outputs, y, meshes) if y.shape[1] == len(outputs) else None     # all steps in one launch
    if part is None:
        part = torch.cat([ops.step_sse_partials(out, y[:, t], mesh) for t, (out, mesh) in enumerate(zip(outputs, meshes))])
    return part.sum() / float(mesh0.B * len(outputs) * n_valid)        # one reduction for all steps


class NextFramePredictor(ABC):
    """The abstract trainer facade of the reference (model/mpnnlstm.py:34-79; moving_mnist_example.ipynb cell 2 imports it):
    it holds the decomposition settings and names the three methods a predictor offers.  `thresh` is kept as given here;
    the concrete class turns `decompose=False` into thresh = -inf (:113)."""

    def __init__(self, thresh, experiment_name='experiment', decompose=True, input_features=1, transform_func=None,
                 condition='max_larger_than', device=None):
        self.experiment_name, self.device, self.model = experiment_name, device, None
        self.thresh, self.decompose = thresh, decompose
        self.transform_func, self.condition, self.input_features = transform_func, condition, input_features

    @abstractmethod
    def train(self, loader_train, loader_test, n_epochs=200, lr=0.01, lr_decay=0.95, mask=None):
        ...

    @abstractmethod
    def predict(self, x, mask=None, rollout=None):
        ...

    @abstractmethod
    def score(self, x, y, rollout=None):
        ...


class NextFramePredictorS2S(NextFramePredictor):
    def __init__(self, thresh, experiment_name='experiment', decompose=True, input_features=1, input_timesteps=3,
                 output_timesteps=3, device=None, transform_func=None, condition='max_larger_than', remesh_input=False,
                 binary=False, debug=False, model_kwargs={}):
        super().__init__(thresh=thresh, experiment_name=experiment_name, decompose=decompose, input_features=input_features,
                         transform_func=transform_func, condition=condition, device=device)
        self.input_timesteps, self.output_timesteps = input_timesteps, output_timesteps
        self.binary, self.debug = binary, debug
        self.thresh = thresh if decompose else -np.inf
        # As in the reference (:123-133) the Seq2Seq gets the RAW `thresh` (decompose=False changes self.thresh only) and
        # its transform_func / condition from model_kwargs alone (ice_exp.py:153-176 passes transform_func twice for that).
        self.model = Seq2Seq(input_features=input_features + 3,      # + positional encoding (x, y) + node size
                             input_timesteps=input_timesteps, output_timesteps=output_timesteps, thresh=thresh,
                             device=device, remesh_input=remesh_input, binary=binary, debug=debug,
                             **model_kwargs).to(device)
        self.training_initiated = False
        self.process_group = None

    # -- small helpers of the reference ---------------------------------------------
    def get_n_params(self):
        return get_n_params(self.model)

    def save(self, directory):
        torch.save(self.model.state_dict(), os.path.join(directory, f'{self.experiment_name}.pth'))

    def load(self, directory):
        path = os.path.join(directory, f'{self.experiment_name}.pth')
        self.model.load_state_dict(torch.load(path, map_location=self.device or 'cpu', weights_only=True))

    def test_threshold(self, x, thresh, mask=None, high_interest_region=None, contours=True):
        import matplotlib.pyplot as plt
        n_sample, w, h, _ = x.shape
        graph = image_to_graph(add_positional_encoding(x), thresh=thresh, mask=mask, high_interest_region=high_interest_region,
                               transform_func=self.transform_func)
        mesh = graph['mapping']
        rec = unflatten(graph['data'][..., [0]], mesh, (w, h)).cpu()
        fig, axs = plt.subplots(1, n_sample, figsize=(5 * n_sample, 4), squeeze=False)
        for i in range(n_sample):
            axs[0, i].imshow(rec[i, ..., 0])
            if contours:
                plot_contours(axs[0, i], mesh.labels[0].cpu().numpy())
        plt.suptitle(f'Threshold: {thresh} | Num. nodes: {mesh.N}')
        return fig, axs[0]

    @on_device(lambda self, *a, **k: self.device)
    def initiate_training(self, lr, lr_decay, capturable=False):
        self.loss_func_name = 'MSE' if not self.binary else 'BCE'
        if capturable:      # optimizer.step() inside a hipGraph needs device-side step counters and lr
            lr = torch.tensor(float(lr), device=self.device)
        # fused: one multi-tensor kernel for all 238 parameter tensors (the default per-tensor path costs ~1000 tiny
        # launches per step once the step counters live on the device)
        fused = self.device is not None and torch.device(self.device).type == 'cuda'
        # Plain ChebConv models on the GPU: every parameter is a view of one flat buffer and the backward pass returns one
        # flat gradient vector (qtmpnn.flat), so the optimizer sees ONE tensor -- Adam is elementwise, and the clipping norm is
        # the norm of all gradients either way, so the update is the reference's; only the launch count differs (two launches
        # for clip + Adam instead of ~25, the all-reduce without a gather copy).
        self.flat = None
        if fused and self.model.encoder.plannable and self.model.decoder.plannable:
            self.flat = flat_params(self.model)
        if self.flat is not None:
            from qtmpnn.optim import FlatAdam
            self.optimizer = FlatAdam(self.flat.param, lr=lr, capturable=capturable)      # clip + Adam in two launches
        else:
            self.optimizer = torch.optim.Adam(self.model.parameters(), lr=lr, capturable=capturable, fused=fused or None)
        self.scheduler = StepLR(self.optimizer, step_size=3, gamma=lr_decay)
        self.writer = SummaryWriter('runs/' + self.experiment_name + '_' + datetime.datetime.now().strftime('%Y%m%d_%H_%M_%S'))
        self.test_loss, self.train_loss = [], []
        self.training_initiated = True

    # -- the measured unit -----------------------------------------------------------
    def forward_loss(self, x, y, concat_layers=None, mask=None, high_interest_region=None, graph_structure=None):
        y_hat, meshes = self.model(x, y, concat_layers, teacher_forcing_ratio=0, mask=mask,
                                   high_interest_region=high_interest_region, graph_structure=graph_structure)
        return masked_mse(y_hat, meshes, y, mask, self.binary)

    def zero_grad(self):
        """Drop all gradients (set to None, like optimizer.zero_grad(set_to_none=True) on the reference's per-tensor optimizer)."""
        if getattr(self, 'flat', None) is not None:
            self.flat.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=True)

    def _grads_ready(self, world=1, group=None, force=False):
        """After backward: average the gradients over the ranks (ONE all-reduce of one flat tensor) and hand them to the
        optimizer.  Returns the tensors clip_grad_norm_ has to see.  force: issue the collective even in a group of one."""
        if self.flat is None:
            params = list(self.model.parameters())
            if world > 1 or force:
                allreduce_gradients(params, group, force=force)
            return params
        if not self.flat.intact(self.model):
            raise RuntimeError('the model parameters were moved or replaced after initiate_training(): call it again')
        g = self.flat.grad_vector()
        if g is None:                          # gradients that did not come from the model-wide packing gather: by copy
            g = self.flat.gather_grads()
        if world > 1 or force:
            all_reduce_sum(g, group)
            g.mul_(1.0 / world)
        self.flat.param.grad = g
        return [self.flat.param]

    def _clip_and_step(self, clip_params, max_norm):
        """clip_grad_norm_(max_norm) + optimizer.step() (mpnnlstm.py:251, 257); max_norm None: no clipping (:311)."""
        if self.flat is not None:
            self.optimizer.step(max_norm=max_norm)
            return
        if max_norm is not None:
            torch.nn.utils.clip_grad_norm_(clip_params, max_norm=max_norm)
        self.optimizer.step()

    def _world(self):
        d = torch.distributed
        if self.process_group is not None:
            return d.get_world_size(self.process_group)
        return d.get_world_size() if d.is_available() and d.is_initialized() else 1

    @on_device(lambda self, *a, **k: self.device)
    def train_step(self, x, y, concat_layers=None, mask=None, high_interest_region=None, graph_structure=None,
                   max_norm=10.0):
        """zero_grad -> forward -> masked MSE -> backward -> [all-reduce] -> clip_grad_norm_(10) -> Adam
        (mpnnlstm.py:229-257).  x: (T_in, W, H, C) or (B, T_in, W, H, C).  Returns the loss tensor."""
        self.zero_grad()
        loss = self.forward_loss(x, y, concat_layers, mask, high_interest_region, graph_structure)
        loss.backward()
        self._clip_and_step(self._grads_ready(self._world(), self.process_group), max_norm)
        check_tile_errors()          # (reads the device only when this step issued tile-resident launches; raises on a failed one)
        return loss.detach()

    @on_device(lambda self, *a, **k: self.device)
    def truncated_backward(self, x, y, concat_layers, mask, high_interest_region=None, graph_structure=None,
                           truncated_backprop=45):
        """The reference's truncated-BPTT loop (mpnnlstm.py:281-315), quirks included: every chunk re-runs the encoder
        and unrolls ITS steps from the encoder state, `zero_grad` runs per chunk (so only the last chunk's gradient
        survives to optimizer.step()), and the chunk bound is min(start + tb, T_out + 1).  Returns the chunk losses.

        Beyond the reference: a chunk is clamped to the steps that exist, range(max(step - tb, 0), min(step, T_out)).  HEAD
        unrolls range(step - tb, step) as it stands, which leaves [0, T_out) whenever tb does not divide T_out -- the default
        tb = 45 with the notebook's T_out = 10 gives range(-34, 11) and ends in an IndexError at y[unroll_steps] (:308).
        Where HEAD runs (tb divides T_out: ice_exp.py exp 5 / 6) the clamp changes nothing."""
        losses, step = [], 0
        if y.dim() == 4:
            y = y.unsqueeze(0)
        while step < self.output_timesteps:
            step = min(step + truncated_backprop, self.output_timesteps + 1)
            steps = range(max(step - truncated_backprop, 0), min(step, self.output_timesteps))
            self.zero_grad()
            self.model.process_inputs(x, mask=mask, high_interest_region=high_interest_region, graph_structure=graph_structure)
            y_hat, meshes = self.model.unroll_output(steps, y, concat_layers=concat_layers, teacher_forcing_ratio=0, mask=mask,
                                                     high_interest_region=high_interest_region, remesh_every=1)
            loss = masked_mse(y_hat, meshes, y[:, steps.start:steps.stop], mask, self.binary)
            loss.backward()
            losses.append(loss.detach())
        return losses

    @on_device(lambda self, *a, **k: self.device)
    def make_graphed_step(self, x, y, concat_layers=None, mask=None, high_interest_region=None, max_norm=10.0,
                          warmup=2, graph_structure=None, force_multi=False):
        """Capture one whole training step in hipGraphs and return `step(x, y, concat) -> loss`.

        The rollout is data dependent (every decoder step re-meshes on its own output), so the capture runs in
        static mode: all node buffers have the worst-case capacity B*W*H and every kernel reads the actual node
        count from device memory -- no host sync, no shape change, ~1.8k launches replayed by one hipGraphLaunch.
        Single process: forward + loss + backward + clip + fused Adam are ONE graph.  Under torch.distributed the
        first graph ends with the gradients packed into one flat buffer; the step is then
        `graph1.replay(); all_reduce(flat); graph2.replay()` with graph2 = average + clip + fused Adam on views of
        that buffer: one collective and three host calls per step.  The `warmup` eager steps are real training steps.
        force_multi: take the multi-rank structure (graph1, all-reduce, graph2) also in a process group of ONE rank -- the
        collective then averages over one rank, i.e. changes nothing, but RCCL, its stream ordering against the two graph
        replays and the capture beside its watchdog thread all run for real (tests/test_gpu_dist.py; bench.py --force-multi).
        """
        import torch.distributed as dist
        if force_multi and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('force_multi=True needs an initialised torch.distributed process group')
        multi = force_multi or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        world = dist.get_world_size(self.process_group) if multi else 1
        self.model.static_shapes = True
        if not self.optimizer.defaults.get('capturable', False):
            lr = self.optimizer.param_groups[0]['lr']
            assert not self.optimizer.state, 'make_graphed_step must be called before the first optimizer step'
            self.initiate_training(float(lr), self.scheduler.gamma, capturable=True)
        sx, sy = x.clone(), y.clone()
        sc = concat_layers.clone() if concat_layers is not None else None

        def fwd_bwd():
            self.zero_grad()
            loss = self.forward_loss(sx, sy, sc, mask, high_interest_region, graph_structure)
            loss.backward()
            return loss.detach()

        def update(clip_params):
            self._clip_and_step(clip_params, max_norm)

        side = torch.cuda.Stream()
        if multi:
            # The warm-up steps all-reduce, and torch issues a synchronous collective -- and records its completion event -- on the
            # CURRENT stream.  The process group's watchdog thread polls that event (hipEventQuery) until the work is reaped, and
            # HIP refuses the query while the event's stream is capturing (hipErrorCapturedEvent: the watchdog dies and takes the
            # process with it; seen in round 5 as soon as RCCL was executed at all).  So under torch.distributed the warm-up runs
            # on the caller's stream and only the capture on the side stream: no collective ever touches a stream that captures.
            for _ in range(warmup):
                self.last_warmup_loss = fwd_bwd()        # (a real training step on this batch)
                update(self._grads_ready(world, self.process_group, force=multi))
            torch.cuda.synchronize()
            side.wait_stream(torch.cuda.current_stream())
        else:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self.last_warmup_loss = fwd_bwd()        # (a real training step on this batch)
                    update(self._grads_ready(world, self.process_group, force=multi))
            torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        self.zero_grad()
        # thread_local: other threads (the RCCL watchdog under torch.distributed) may issue HIP calls meanwhile
        with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
            static_loss = fwd_bwd()
            if multi:
                # the graph ends with the gradients in ONE flat buffer: the packing gather's own output on the flat path
                # (no copy), a concatenation otherwise
                if self.flat is not None:
                    flat = self.flat.grad_vector()
                    flat = flat if flat is not None else self.flat.gather_grads()
                else:
                    params = list(self.model.parameters())
                    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
            else:
                update(self._grads_ready())
        self._graph = graph
        graph2 = None
        if multi:
            if self.flat is not None:
                self.flat.param.grad = flat
                clip_params = [self.flat.param]
            else:
                off = 0
                for p in params:                          # gradients become views of the flat buffer: no unpack copies
                    p.grad = flat[off:off + p.numel()].view_as(p)
                    off += p.numel()
                clip_params = params
            graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph2, stream=side, capture_error_mode='thread_local'):
                flat.mul_(1.0 / world)
                update(clip_params)

        # the captured launches report failures (a tile-resident launch that gave up waiting) only through the device's persistent
        # error word: it is read here after the capture and then every 64 replays -- one 4-byte copy, only for graphs that
        # contain such launches at all
        from qtmpnn import mesh as _mesh
        uses_tiles = bool(_mesh._TILE_USED)
        check_tile_errors(always=uses_tiles)
        replays = [0]

        @on_device(lambda *a, **k: self.device)
        def step(x, y, concat_layers=None):
            sx.copy_(x)
            sy.copy_(y)
            if sc is not None:
                sc.copy_(concat_layers)
            graph.replay()
            if multi:
                all_reduce_sum(flat, self.process_group)
                graph2.replay()
            replays[0] += 1
            if uses_tiles and replays[0] % 64 == 0:
                check_tile_errors(always=True)
            return static_loss
        step.check = lambda: check_tile_errors(always=uses_tiles)
        return step

    @on_device(lambda self, *a, **k: self.device)
    def train(self, loader_train, loader_test, climatology=None, n_epochs=200, lr=0.01, lr_decay=0.95, mask=None,
              high_interest_region=None, truncated_backprop=45, graph_structure=None, use_graph=False):
        """The reference's training loop (mpnnlstm.py:186-387).  use_graph=True (beyond the reference) replays the whole training
        step as a hipGraph: one graph is captured per distinct batch shape on first sight (that batch's own update runs eagerly just
        before the capture) and the learning-rate schedule keeps working because the capturable optimizer holds lr in a device
        tensor that StepLR updates in place.  It needs the step to be ONE rollout: truncated_backprop in (0, None), or a truncation
        length that covers all output steps (the default 45 with the notebook's 10: the truncated loop is then a single chunk over
        the whole rollout, without gradient clipping -- mpnnlstm.py:311 is commented out -- and that is what is captured)."""
        image_shape = loader_train.dataset.image_shape
        truncate_ = truncated_backprop not in (0, None)
        single_chunk = truncate_ and truncated_backprop >= self.output_timesteps
        if use_graph and truncate_ and not single_chunk:
            raise ValueError('use_graph=True needs truncated_backprop=0 or >= the output steps (the truncated loop re-runs the encoder '
                             'per chunk)')
        if not self.training_initiated:
            self.initiate_training(lr, lr_decay, capturable=use_graph)
        graphed = {}
        if mask is not None:
            mshape = tuple(host_mask(mask).shape) if not hasattr(mask, 'shape') else tuple(mask.shape)
            assert mshape == tuple(image_shape), f'Mask and image shapes do not match. Got {mshape} and {image_shape}'
        truncate = truncate_ and not (use_graph and single_chunk)
        st = time.time()
        batch_step = 0
        for epoch in range(n_epochs):
            running, steps = 0.0, 0
            # (the mode is the caller's, as in the reference: train() never calls model.train() / .eval(); ice_exp.py:181, 218 do)
            for x, y, launch_date in loader_train:
                x, y = self._clip(x), self._clip(y)
                concat = self.get_climatology_array(climatology, launch_date) if climatology is not None else None
                if truncate:
                    loss = self.truncated_backward(x, y, concat, mask, high_interest_region, graph_structure,
                                                   truncated_backprop)[-1]
                    # no gradient clipping in this branch (:311 is commented out)
                    self._clip_and_step(self._grads_ready(self._world(), self.process_group), None)
                elif use_graph:
                    key = (tuple(x.shape), tuple(y.shape), None if concat is None else tuple(concat.shape))
                    if key not in graphed:      # first sight of this batch shape: its update runs eagerly, then the capture
                        graphed[key] = self.make_graphed_step(x, y, concat, mask=mask, high_interest_region=high_interest_region,
                                                              warmup=1, graph_structure=graph_structure,
                                                              max_norm=None if single_chunk else 10.0)
                        loss = self.last_warmup_loss
                    else:
                        loss = graphed[key](x, y, concat)
                else:
                    loss = self.train_step(x, y, concat, mask, high_interest_region, graph_structure)
                    if self.debug:      # gradient norms of the two halves of the model (:259-276; clipped like the reference's)
                        for part in ('encoder', 'decoder'):
                            gs = [p.grad.detach().norm() for p in getattr(self.model, part).parameters() if p.grad is not None]
                            self.writer.add_scalar(f'Grad/{part}/grad_norms', torch.stack(gs).norm().item(), batch_step)
                self.writer.add_scalar('Loss/train', loss.item(), batch_step)
                running += loss.item()
                steps += 1
                batch_step += 1
            running_test, steps_test = 0.0, 0
            for x, y, launch_date in loader_test:
                x, y = self._clip(x), self._clip(y)
                concat = self.get_climatology_array(climatology, launch_date) if climatology is not None else None
                with torch.no_grad():
                    running_test += self.forward_loss(x, y, concat, mask, high_interest_region, graph_structure).item()
                steps_test += 1
            check_tile_errors(always=True)      # once per epoch: the graph-replayed steps and the test loop report only here
            running, running_test = running / (steps + 1), running_test / (steps_test + 1)   # (+1 as the reference, :360-361)
            if np.isnan(running_test):
                raise ValueError('NaN loss :(')
            if running_test > 4:
                raise ValueError('Diverged :(')
            self.writer.add_scalar('Loss/test', running_test, epoch)
            self.scheduler.step()
            self.train_loss.append(running)
            self.test_loss.append(running_test)
            print(f'{self.experiment_name} | Epoch {epoch} train {self.loss_func_name}: {running:.4f}, '
                  f'test {self.loss_func_name}: {running_test:.4f}, lr: {self.scheduler.get_last_lr()[0]:.4f}, '
                  f'time_per_epoch: {(time.time() - st) / (epoch + 1):.1f}')
        print(f'Finished in {(time.time() - st) / 60} minutes')
        self.writer.flush()
        self.loss = pd.DataFrame({'train_loss': self.train_loss, 'test_loss': self.test_loss})

    def _clip(self, t):
        """Loader items are (1, T, W, H, C) like the reference's batch_size=1 loaders, or (B, T, W, H, C)."""
        t = t.to(self.device)
        return t.squeeze(0) if t.shape[0] == 1 else t

    def get_climatology_array(self, climatology, launch_date):
        """Daily normals of the output days, (T_out, W, H, 1) (mpnnlstm.py:389-400)."""
        doys = [int_to_datetime(launch_date.numpy()[0] + 8.640e13 * t).timetuple().tm_yday - 1
                for t in range(self.output_timesteps)]
        return torch.moveaxis(climatology[:, doys], 0, -1)

    @on_device(lambda self, *a, **k: self.device)
    def predict(self, loader, climatology=None, mask=None, high_interest_region=None, graph_structure=None):
        """Inference over a loader -> (n_clips, T_out, W, H, 1) array (mpnnlstm.py:402-440)."""
        image_shape = loader.dataset.image_shape
        self.model.to(self.device)
        preds = []
        for x, y, launch_date in loader:
            x = self._clip(x)
            concat = self.get_climatology_array(climatology, launch_date) if climatology is not None else None
            with torch.no_grad():
                y_hat, meshes = self.model(x, concat_layers=concat, teacher_forcing_ratio=0, mask=mask,
                                           high_interest_region=high_interest_region, graph_structure=graph_structure)
                frames = [unflatten(o, ms, image_shape, mask).cpu().numpy() for o, ms in zip(y_hat, meshes)]
            frames = np.stack(frames, axis=0 if x.dim() == 4 else 1)
            preds.extend([frames] if x.dim() == 4 else list(frames))
        check_tile_errors(always=True)
        return np.stack(preds, 0)

    def score(self, x, y, rollout=None):
        pass
