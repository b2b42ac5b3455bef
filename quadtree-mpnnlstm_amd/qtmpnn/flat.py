"""One flat fp32 buffer behind all parameters of a module, and one flat gradient vector per backward pass.

The reference model has 238 parameter tensors of a few hundred floats each (SURVEY KAT-6); per-tensor optimizer, clipping and
all-reduce launches cost more than the arithmetic.  Here every parameter is a VIEW of one buffer (module.parameters() order,
state-dict keys and shapes untouched), the weight packing (ops.PackPlan) reads that buffer directly and its backward returns
every parameter's gradient as a view of ONE vector -- so gradient all-reduce, clip_grad_norm_ and Adam are single-tensor
operations on `FlatParams.param` / `.param.grad`.
"""
import torch


def param_list(module):
    """list(module.parameters()), remembered on the module: walking ~300 submodules costs ~1.5 ms and an eager training step asks
    four times.  Every call re-checks that each (submodule, name) slot still holds the remembered Parameter -- 238 dict look-ups --
    so a re-assigned Parameter is seen (then the list is rebuilt); modules or parameters REGISTERED after the first call are not
    (nothing here does that: `del module.__dict__['_param_slots']` after such surgery)."""
    slots = module.__dict__.get('_param_slots')
    if slots is not None and all(m._parameters.get(n) is p for m, n, p in slots):
        return [p for _, _, p in slots]
    seen, slots = set(), []
    for m in module.modules():
        for n, p in m._parameters.items():
            if p is not None and id(p) not in seen:
                seen.add(id(p))
                slots.append((m, n, p))
    ref = list(module.parameters())
    assert len(ref) == len(slots) and all(a is b[2] for a, b in zip(ref, slots))       # (same order as module.parameters())
    module.__dict__['_param_slots'] = slots
    return ref


class FlatParams:
    def __init__(self, module):
        self.params = param_list(module)
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.device == dev and p.dtype == dt for p in self.params), 'parameters must share device and dtype'
        self.sizes = [p.numel() for p in self.params]
        self.offs = [0]
        for n in self.sizes:
            self.offs.append(self.offs[-1] + n)
        self.n = self.offs[-1]
        # (+1: a trailing zero that the packing gather uses for padding elements; +3 more keep 16-byte multiples)
        self.buffer = torch.zeros(self.n + 4, device=dev, dtype=dt)
        with torch.no_grad():
            torch.cat([p.detach().reshape(-1) for p in self.params], out=self.buffer[:self.n])
            for p, o, n in zip(self.params, self.offs, self.sizes):
                p.data = self.buffer[o:o + n].view(p.shape)
        self.param = torch.nn.Parameter(self.buffer[:self.n])       # the optimizer's single tensor (same storage)
        self.last_grad = None                                       # set by the packing gather's backward

    def intact(self, module=None):
        """True while every parameter still is the view this object made (module.to(), a re-assigned Parameter or .data break
        it: the caller then builds a new FlatParams)."""
        base, es = self.buffer.data_ptr(), self.buffer.element_size()
        ok = all(p.data_ptr() == base + es * o for p, o in zip(self.params, self.offs))
        if ok and module is not None:
            ps = param_list(module)
            ok = len(ps) == len(self.params) and all(a is b for a, b in zip(ps, self.params))
        return ok

    def grad_vector(self):
        """The flat gradient of the last backward pass when every parameter's .grad is a view of it, else None."""
        g = self.last_grad
        if g is None:
            return None
        base, es = g.data_ptr(), g.element_size()
        for p, o in zip(self.params, self.offs):
            if p.grad is None or p.grad.data_ptr() != base + es * o:
                return None
        return g[:self.n]

    def gather_grads(self):
        """Flat gradient by copy (parameters without a gradient count as zero): the fallback when the gradients did not come
        from one packing gather."""
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params])

    def zero_grad(self):
        for p in self.params:
            p.grad = None
        self.param.grad = None
        self.last_grad = None


def flat_params(module):
    """The module's FlatParams (made on first use, remade when the parameters were moved or replaced), or None when the
    parameters are not fp32 (model.double() / .half(): the packing gather and qt_flat_adam are fp32 kernels -- the callers
    then fall back to per-tensor packing and torch.optim.Adam)."""
    if any(p.dtype != torch.float32 for p in param_list(module)):
        return None
    fp = module.__dict__.get('_flat_params')
    if fp is None or not fp.intact(module):
        fp = FlatParams(module)
        module.__dict__['_flat_params'] = fp
    return fp
