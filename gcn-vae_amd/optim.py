"""Gradient clipping + Adam over ONE flat parameter arena (kgvae/link_predict.py:227-228:
``clip_grad_norm_(model.parameters(), grad_norm); optimizer.step()``).

Parameters and their gradients are re-pointed at slices of two persistent device buffers, so the
whole optimiser step is two launches (per-block sums of squares + step counter; clip+Adam, each block finishing
the norm from the partials in a fixed order and clearing the gradient it has consumed) instead of torch's
per-tensor / foreach kernel sets, and every address is static -- the step is hipGraph-capturable.
Numerics follow ``torch.nn.utils.clip_grad_norm_`` (coefficient min(1, max_norm / (norm + 1e-6))) and
``torch.optim.Adam`` (bias correction, eps added outside the square root).
"""
import torch

from . import lib, ops
from .lib import ptr


class FlatAdam:
    ALIGN = 64   # floats: keeps every parameter 256-B aligned for the 16-B vector paths of the kernels

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=None, total_multiple=1, moments=True):
        """total_multiple: the arena length is rounded up to a multiple of it (a sharded optimiser cuts the arena into equal pieces);
        moments=False: the caller keeps its own (smaller) moment arenas."""
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('FlatAdam: no trainable parameters')
        dev = self.params[0].device
        if dev.type != 'cuda':
            raise RuntimeError('FlatAdam runs on a ROCm device only; move the model first (no CPU fallback)')
        self.lr, self.betas, self.eps, self.max_grad_norm = lr, betas, eps, max_grad_norm
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        total = (total + int(total_multiple) - 1) // int(total_multiple) * int(total_multiple)
        self.total = total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total if moments else 0, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total if moments else 0, dtype=torch.float32, device=dev)
        self.step_t = torch.zeros((), dtype=torch.float32, device=dev)
        self.sumsq = torch.zeros((), dtype=torch.float32, device=dev)
        self._ws = torch.empty(1024, dtype=torch.float32, device=dev)
        self._direct_keys = []
        self._clean = False
        ops.DIRECT_EPOCH[0] += 1
        self.offsets = {p: o for p, o in zip(self.params, offs)}     # parameter -> first float of its arena slice
        for p, o in zip(self.params, offs):
            n = p.numel()
            self.flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + n].view(p.shape)
            p.grad = self.flat_g[o:o + n].view(p.shape)
            ops.DIRECT_GRAD[p.data.data_ptr()] = p.grad      # backward kernels add straight into the arena
            self._direct_keys.append(p.data.data_ptr())

    def close(self):
        """Withdraw this optimiser's direct-gradient registrations.  Only entries that still point at THIS arena are touched
        (and the epoch only bumped if any was): an optimiser collected late -- e.g. out of a reference cycle, between a newer
        optimiser's forward and backward -- must not invalidate the live one's registrations."""
        mine = False
        for p in getattr(self, 'params', ()):
            k = p.data.data_ptr()
            g = ops.DIRECT_GRAD.get(k)
            if g is not None and g.data_ptr() >= self.flat_g.data_ptr() and \
                    g.data_ptr() < self.flat_g.data_ptr() + self.flat_g.numel() * 4:
                ops.DIRECT_GRAD.pop(k, None)
                ops.GRAD_FRESH.discard(g.data_ptr())
                mine = True
        if mine:
            ops.DIRECT_EPOCH[0] += 1
        self._direct_keys = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _mark_fresh(self):
        # every gradient slice is zero now: backward kernels that can store a gradient instead of adding it may do so once
        for p in self.params:
            ops.GRAD_FRESH.add(p.grad.data_ptr())

    def zero_grad(self):
        ops.backward_side_finish()
        # the clip+Adam kernel clears the arena as it consumes it: right after step() there is nothing to do
        if self._clean:
            self._clean = False
            self._mark_fresh()
            return
        self.flat_g.zero_()
        self._mark_fresh()

    def snapshot(self):
        """Copies of the parameter arena, both moment arenas and the step counter (device tensors): what ``restore`` puts back."""
        return tuple(t.clone() for t in (self.flat_p, self.exp_avg, self.exp_avg_sq, self.step_t))

    def restore(self, snap):
        """Put a ``snapshot`` back IN PLACE (every address stays what a captured step recorded) and clear the gradient arena."""
        for t, s in zip((self.flat_p, self.exp_avg, self.exp_avg_sq, self.step_t), snap):
            t.copy_(s)
        self.flat_g.zero_()
        self._clean = True

    def grad_norm(self):
        """Total gradient norm of the last ``step`` (device scalar)."""
        return self.sumsq.sqrt()

    def step(self):
        ops.backward_side_finish()        # (a no-op after a backward pass that ran to its end)
        if self.max_grad_norm is not None:
            lib.call('gv_clip_adam_step', ptr(self.flat_p), ptr(self.flat_g), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                     self.total, ptr(self._ws), ptr(self.sumsq), float(self.max_grad_norm), float(self.lr),
                     float(self.betas[0]), float(self.betas[1]), float(self.eps), ptr(self.step_t), 1, lib.stream())
            self._clean = True
            return
        self.step_t += 1
        sumsq = None
        lib.call('gv_adam_step', ptr(self.flat_p), ptr(self.flat_g), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                 self.total, ptr(sumsq), float(self.max_grad_norm or 0.0), float(self.lr), float(self.betas[0]),
                 float(self.betas[1]), float(self.eps), ptr(self.step_t), lib.stream())
