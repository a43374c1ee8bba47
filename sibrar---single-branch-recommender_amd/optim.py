"""Fused dense optimizers over ONE flat fp32 parameter buffer (train/trainer.py:62-68 builds a dense torch optimizer over
``model.parameters()``; every element of every embedding table is decayed and updated on every step).

``FlatParameters`` re-homes all parameters of a module into one contiguous buffer (parameters become views, strides kept,
so the column-major CSR projector weight stays column-major) and gives them a matching flat gradient buffer. One kernel
launch then updates the whole model, ``zero_grad`` is one memset, and the data-parallel gradient exchange is a single RCCL
all-reduce of the flat gradient buffer (parallel.py).
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops


class FlatParameters:
    def __init__(self, module: nn.Module):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError('module has no trainable parameters')
        dev = params[0].device
        if dev.type != 'cuda':
            raise RuntimeError('FlatParameters needs the module on a CUDA(HIP) device')
        self.params = params
        sizes = [p.numel() for p in params]
        # 64-element (256 B) alignment of every segment
        offs, total = [], 0
        for s in sizes:
            offs.append(total)
            total += (s + 63) // 64 * 64
        self.offsets, self.sizes, self.total = offs, sizes, total
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(total, device=dev, dtype=torch.float32)
        for p, o, s in zip(params, offs, sizes):
            self._rehome(p, self.flat, o, s, is_grad=False)
            self._rehome(p, self.grad, o, s, is_grad=True)

    @staticmethod
    def _view_like(p: torch.Tensor, buf: torch.Tensor, off: int, n: int) -> torch.Tensor:
        # dense, non-overlapping (possibly permuted) layouts only
        return torch.as_strided(buf, p.shape, p.stride(), off)

    def _rehome(self, p, buf, off, n, is_grad):
        dense = sorted(zip(p.stride(), p.shape), reverse=True)
        expect, ok = 1, True
        for st, sh in reversed(dense):
            if sh != 1 and st != expect:
                ok = False
            expect *= sh
        if not ok:
            raise ValueError('parameter layout is not dense; cannot flatten')
        view = self._view_like(p, buf, off, n)
        if is_grad:
            p.grad = view
        else:
            view.copy_(p.data)
            p.data = view

    def zero_grad(self):
        self.grad.zero_()
        for p, o, s in zip(self.params, self.offsets, self.sizes):   # autograd may have replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self._view_like(p, self.grad, o, s)


class FusedOptimizer:
    """AdamW / Adam / Adagrad with torch's defaults (betas 0.9/0.999, eps 1e-8; Adagrad eps 1e-10), one launch per step."""

    def __init__(self, module: nn.Module, name: str = 'adamw', lr: float = 1e-3, weight_decay: float = 0.):
        if name not in ('adamw', 'adam', 'adagrad'):
            raise KeyError(name)
        self.deferred = None         # engine.DeferredTable of a fused step that updates one lookup table row by row
        self.name, self._lr, self._wd = name, float(lr), float(weight_decay)
        self.fp = FlatParameters(module)
        self.step_count = 0
        n, dev = self.fp.total, self.fp.flat.device
        self.m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(n, device=dev, dtype=torch.float32) if name != 'adagrad' else None

    # lr / wd: a deferred table replays the zero-gradient steps its rows still owe with the hyper-parameters of the call that replays them
    # (the per-step schedule keeps only the bias corrections), so every row is brought up to date BEFORE a hyper-parameter changes:
    # the steps taken so far were taken with the old value on every row, exactly as the dense optimizer would have
    @property
    def lr(self) -> float:
        return self._lr

    @lr.setter
    def lr(self, value: float):
        if float(value) != self._lr and self.deferred is not None:
            self.deferred.flush()
        self._lr = float(value)

    @property
    def wd(self) -> float:
        return self._wd

    @wd.setter
    def wd(self, value: float):
        if float(value) != self._wd and self.deferred is not None:
            self.deferred.flush()
        self._wd = float(value)

    def _sync_grads(self):
        """Gradients that autograd produced as fresh tensors (instead of accumulating into the flat views) are copied in."""
        fp = self.fp
        for p, o, s in zip(fp.params, fp.offsets, fp.sizes):
            if p.grad is None:
                continue
            if p.grad.data_ptr() != fp.grad.data_ptr() + 4 * o:
                fp._view_like(p, fp.grad, o, s).copy_(p.grad)

    def step(self):
        self._sync_grads()
        self.step_flat()

    def step_flat(self, skip=None, zero_grad: bool = False, copy=None, rows=None):
        """Update from the flat gradient buffer as it is (the fused step writes gradients there directly). ``zero_grad``: the
        gradient buffer is reset by the optimizer's own launch (Adam / AdamW; Adagrad zeroes it with a fill). ``copy`` = (src, dst)
        small float64 tensors: copied by that launch too (returns True when it was; Adam / AdamW with ``zero_grad`` only).
        ``rows`` (with ``zero_grad``): ids / rows of the deferred table that received gradient -> one launch for everything.
        ``skip`` = (lo, hi): leave that range of the flat buffers alone — a lookup table whose rows the fused step updates
        itself, deferred row by row (engine.DeferredTable). A step without ``skip`` first brings such a table up to date."""
        if rows is not None and self.deferred is not None and zero_grad and skip is None:
            # ``rows``: the rows of the deferred lookup table that received gradient in this step — the table is updated row by
            # row inside the same launch that steps every other parameter densely (engine.DeferredTable.step)
            return self.deferred.step(rows, copy=copy)
        if skip is None and self.deferred is not None:
            self.deferred.flush()
        self.step_count += 1
        fp = self.fp
        if self.name == 'adagrad':
            ops.adagrad_step(fp.flat, fp.grad, self.m, self.lr, 1e-10, self.wd)
            if zero_grad:
                fp.grad.zero_()
            return False
        kind = 0 if self.name == 'adamw' else 1
        copied = False
        for lo, hi in ([(0, fp.total)] if skip is None else [(0, skip[0]), (skip[1], fp.total)]):
            if hi > lo:
                cp = copy if (zero_grad and copy is not None and not copied) else None
                ops.adam_step(kind, fp.flat[lo:hi], fp.grad[lo:hi], self.m[lo:hi], self.v[lo:hi], self.lr, 0.9, 0.999, 1e-8,
                              self.wd, self.step_count, zero_grad=zero_grad, copy=cp)
                copied = copied or cp is not None
        if skip is None and self.deferred is not None:
            self.deferred.mark_all_current()           # this step updated the table densely
        return copied

    def zero_grad(self):
        self.fp.zero_grad()
