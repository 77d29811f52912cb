"""Fused optimizers over flat parameter / gradient / state arenas.

``RMSprop`` and ``Adam`` reproduce ``torch.optim.RMSprop(lr)`` and
``torch.optim.Adam(lr, betas=(0, 0.9))`` as built by the reference
(segan/models/model.py:219-228) — same update rule, same ``state_dict`` layout, so
optimizer checkpoints are interchangeable — but every parameter of a network lives in
ONE contiguous fp32 arena (parameters become views of it), as do the gradients and the
optimizer state.  A step is then a single HIP kernel over the arena instead of one
launch per tensor, ``zero_grad`` is one fill, and the data-parallel gradient
all-reduce is one RCCL call on ``flat_grad`` (see distributed.py).
"""
import torch

from . import ops

_ALIGN = 64  # floats (256 B) between parameter views


class _FlatOptimizer(torch.optim.Optimizer):
    _state_names = ()

    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self._params = [p for g in self.param_groups for p in g['params']]
        if not self._params:
            raise ValueError('optimizer got an empty parameter list')
        self._build_arenas()

    # ---- arenas ---------------------------------------------------------------------
    def _build_arenas(self):
        dev = self._params[0].device
        offs, off = [], 0
        for p in self._params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError('all parameters must be fp32 on one device')
            offs.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self._offsets, self._total = offs, off
        self.flat_param = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self._flat_state = {n: torch.zeros(off, device=dev, dtype=torch.float32)
                            for n in self._state_names}
        with torch.no_grad():
            for p, o in zip(self._params, offs):
                view = self.flat_param[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                g = self.flat_grad[o:o + p.numel()].view(p.shape)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.grad = g
                st = self.state[p]
                st['step'] = torch.tensor(0.0, dtype=torch.float32)
                for n in self._state_names:
                    st[n] = self._flat_state[n][o:o + p.numel()].view(p.shape)
        ops.bump_weights_epoch()

    def _view(self, flat, i):
        p = self._params[i]
        o = self._offsets[i]
        return flat[o:o + p.numel()].view(p.shape)

    def _resync(self):
        """Re-attach parameters / gradients that were re-allocated behind our back
        (``model.to()``, ``p.grad = None`` followed by a backward, ...)."""
        dev = self.flat_param.device
        for i, p in enumerate(self._params):
            o = self._offsets[i]
            want_p = self.flat_param.data_ptr() + 4 * o
            if p.device != dev:
                self._build_arenas()
                return
            if p.data_ptr() != want_p:
                with torch.no_grad():
                    v = self._view(self.flat_param, i)
                    v.copy_(p.data)
                    p.data = v
            g = p.grad
            want_g = self.flat_grad.data_ptr() + 4 * o
            if g is None:
                v = self._view(self.flat_grad, i)
                v.zero_()
                p.grad = v
            elif g.data_ptr() != want_g:
                v = self._view(self.flat_grad, i)
                v.copy_(g)
                p.grad = v

    def zero_grad(self, set_to_none=False):
        """Zero the gradient arena (gradients stay views of it; ``set_to_none`` is
        accepted for API compatibility and ignored)."""
        self._resync()
        if self.flat_grad.is_cuda:
            ops.fill_(self.flat_grad, 0.0)
        else:
            self.flat_grad.zero_()

    def state_dict(self):
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # copy the loaded state into the arenas and re-point the views
        with torch.no_grad():
            for i, p in enumerate(self._params):
                st = self.state[p]
                for n in self._state_names:
                    v = self._view(self._flat_state[n], i)
                    if n in st:
                        v.copy_(st[n])
                    st[n] = v
                if 'step' not in st:
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)
                elif not torch.is_tensor(st['step']):
                    st['step'] = torch.tensor(float(st['step']), dtype=torch.float32)

    def _bump_steps(self):
        for p in self._params:
            st = self.state[p]
            st['step'] = st['step'] + 1 if torch.is_tensor(st['step']) else st['step'] + 1

    def _require_cuda(self):
        if not self.flat_param.is_cuda:
            raise RuntimeError('segan_pytorch_amd optimizers step only on an MI355X (HIP) device; '
                               'parameters are on {}'.format(self.flat_param.device))


class RMSprop(_FlatOptimizer):
    """torch.optim.RMSprop (momentum 0, not centered, no weight decay)."""
    _state_names = ('square_avg',)

    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0, momentum=0,
                 centered=False):
        if weight_decay != 0 or momentum != 0 or centered:
            raise NotImplementedError('only plain RMSprop (as model.py:221-222 builds it) is '
                                      'implemented')
        defaults = dict(lr=lr, momentum=momentum, alpha=alpha, eps=eps, centered=centered,
                        weight_decay=weight_decay, capturable=False, foreach=None,
                        maximize=False, differentiable=False)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._require_cuda()
        self._resync()
        g = self.param_groups[0]
        if len(self.param_groups) != 1:
            raise NotImplementedError('a single param group is supported')
        ops.rmsprop_step(self.flat_param, self.flat_grad, self._flat_state['square_avg'],
                         float(g['lr']), float(g['alpha']), float(g['eps']))
        self._bump_steps()
        ops.bump_weights_epoch(self._params)
        return loss


class Adam(_FlatOptimizer):
    """torch.optim.Adam without weight decay / amsgrad (model.py:224-225 uses
    betas=(0, 0.9))."""
    _state_names = ('exp_avg', 'exp_avg_sq')

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0,
                 amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError('weight decay / amsgrad are not implemented')
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad,
                        maximize=False, foreach=None, capturable=False, differentiable=False,
                        fused=None)
        super().__init__(params, defaults)
        self._nsteps = 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._require_cuda()
        self._resync()
        if len(self.param_groups) != 1:
            raise NotImplementedError('a single param group is supported')
        g = self.param_groups[0]
        st0 = self.state[self._params[0]]['step']
        self._nsteps = int(st0.item() if torch.is_tensor(st0) else st0) + 1
        ops.adam_step(self.flat_param, self.flat_grad, self._flat_state['exp_avg'],
                      self._flat_state['exp_avg_sq'], float(g['lr']), float(g['betas'][0]),
                      float(g['betas'][1]), float(g['eps']), self._nsteps)
        self._bump_steps()
        ops.bump_weights_epoch(self._params)
        return loss
