"""Adam + gradient clipping on the model's flat parameter vector (qtmpnn.flat) in two launches (qt_flat_adam).

Same update as the reference's `clip_grad_norm_(params, 10)` + `torch.optim.Adam(params, lr)` (model/mpnnlstm.py:174, 251-257):
Adam is elementwise and the clipping norm is the norm over all parameters, so one flat tensor gives the reference's numbers.
Keeps the torch.optim.Optimizer surface (`param_groups[0]['lr']`, `state`, `zero_grad`, StepLR works on it); with
capturable=True the learning rate lives in a device tensor and the step counter on the device, so the step can be captured
in a hipGraph and a scheduler's in-place update is seen by every replay.
"""
import torch

from . import _lib
from ._lib import ptr


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, flat_param, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        if not (torch.is_tensor(flat_param) and flat_param.dim() == 1 and flat_param.is_cuda and flat_param.dtype == torch.float32):
            raise ValueError('FlatAdam takes ONE flat fp32 CUDA parameter vector (qtmpnn.flat.FlatParams.param)')
        if capturable and not torch.is_tensor(lr):
            lr = torch.tensor(float(lr), device=flat_param.device)
        super().__init__([flat_param], dict(lr=lr, betas=betas, eps=eps, capturable=capturable, fused=True))
        self.last_norm = torch.zeros(2, device=flat_param.device)       # [|g| before clipping, -]

    def _state(self, p):
        st = self.state[p]
        if not st:
            st['step'] = torch.zeros(1, dtype=torch.int32, device=p.device)
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None, max_norm=None):
        """One update; max_norm: clip the gradient to this total norm first (in place, like clip_grad_norm_), None = no
        clipping.  The gradient norm before clipping is left in `self.last_norm[0]` (device)."""
        loss = closure() if closure is not None else None
        group = self.param_groups[0]
        p = group['params'][0]
        if p.grad is None:
            return loss
        g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
        st = self._state(p)
        lr = group['lr']
        b1, b2 = group['betas']
        _lib.call('qt_flat_adam', ptr(p), ptr(g), ptr(st['exp_avg']), ptr(st['exp_avg_sq']), p.numel(), ptr(st['step']),
                  ptr(lr) if torch.is_tensor(lr) else None, 0.0 if torch.is_tensor(lr) else float(lr), float(b1), float(b2),
                  float(group['eps']), float(max_norm) if max_norm is not None else 0.0, ptr(self.last_norm))
        if g is not p.grad:
            p.grad.copy_(g)
        return loss
