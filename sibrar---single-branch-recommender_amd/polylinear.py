"""PolyLinear — the reference's MLP builder (modules/polylinear.py:17-77) with the same constructor, the same submodule
names (= state_dict keys ``layers.linear_{i}.*``, ``layers.batch_norm_{i}.*``, ``layers.batch_norm.*``) and the same layer
order, executed by the HIP kernels: each Linear is an fp32-MFMA GEMM with bias and activation fused into the epilogue, each
BatchNorm1d is the two-pass HIP BatchNorm with the following activation fused in.

The ``nn.Linear`` / ``nn.BatchNorm1d`` / activation modules inside ``self.layers`` are parameter holders only; their own
``forward`` is never called on the hot path.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
from torch import nn

from . import ops

ACTIVATION_FN_MAP = {
    'relu': nn.ReLU(),
    'tanh': nn.Tanh(),
    'sigmoid': nn.Sigmoid(),
    'selu': nn.SELU(),
}


def get_activation_fn(activation_fn):
    return ACTIVATION_FN_MAP[activation_fn] if isinstance(activation_fn, str) else activation_fn


def batch_norm_act(bn: nn.BatchNorm1d, x: torch.Tensor, act: int, training: bool) -> torch.Tensor:
    if training:
        return ops.BatchNormActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, act)
    return ops.batch_norm_eval(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, act)


_drop_counter = [0]


def dropout(x: torch.Tensor, p: float, training: bool) -> torch.Tensor:
    if not training or p is None or p == 0.:
        return x
    _drop_counter[0] += 1
    seed = (torch.initial_seed() * 1000003 + _drop_counter[0]) & 0x7FFFFFFFFFFFFFFF
    return ops.DropoutFn.apply(x, float(p), seed)


class PolyLinear(nn.Module):
    def __init__(self, layer_config: list, activation_fn='relu', output_fn='relu', input_dropout=None,
                 l1_weight_decay=None, apply_batch_norm_every: int = 0):
        super().__init__()
        assert len(layer_config) > 1, "For a linear network, we at least need one input and one output dimension"
        if l1_weight_decay and l1_weight_decay > 0.0:
            # the reference wraps the layer in torchlayers.regularization.L1 (polylinear.py:52-54), a package that is not
            # part of its environment either; no shipped configuration sets it
            raise NotImplementedError('l1_weight_decay needs the third-party "torchlayers" package and is not supported')

        self.layer_config = list(layer_config)
        self.activation_fn = get_activation_fn(activation_fn)
        self.output_fn = get_activation_fn(output_fn) if output_fn is not None else None
        self.n_layers = len(layer_config) - 1
        self.apply_batch_norm_every = apply_batch_norm_every
        self.input_dropout = input_dropout

        layer_dict = OrderedDict()
        if input_dropout is not None:
            layer_dict['input_dropout'] = nn.Dropout(p=input_dropout)
        for i, (d1, d2) in enumerate(zip(layer_config[:-1], layer_config[1:])):
            layer_dict[f'linear_{i}'] = nn.Linear(in_features=d1, out_features=d2)
            if apply_batch_norm_every > 0 and (i + 1) % apply_batch_norm_every == 0:
                layer_dict[f'batch_norm_{i}'] = nn.BatchNorm1d(num_features=d2)
            if i < self.n_layers - 1:
                layer_dict[f'{self.activation_fn.__class__.__name__.lower()}_{i}'] = self.activation_fn
        if apply_batch_norm_every == -1:
            layer_dict['batch_norm'] = nn.BatchNorm1d(num_features=layer_config[-1])
        if self.output_fn is not None:
            layer_dict[f'{self.output_fn.__class__.__name__.lower()}'] = self.output_fn
        self.layers = nn.Sequential(layer_dict)

        self._act = ops.act_code(self.activation_fn)
        self._out_act = ops.act_code(self.output_fn) if self.output_fn is not None else 0

    def layer_plan(self):
        """[(linear, bn | None, act code)] in execution order; the activation is fused into the BN when there is one."""
        plan = []
        for i in range(self.n_layers):
            lin = getattr(self.layers, f'linear_{i}')
            bn = getattr(self.layers, f'batch_norm_{i}', None)
            last = i == self.n_layers - 1
            if last and self.apply_batch_norm_every == -1:
                bn = self.layers.batch_norm
            plan.append((lin, bn, self._out_act if last else self._act))
        return plan

    def forward(self, x):
        lead = x.shape[:-1]
        x = x.reshape(-1, x.shape[-1])
        x = dropout(x, self.input_dropout, self.training)
        for lin, bn, act in self.layer_plan():
            if bn is None:
                x = ops.LinearActFn.apply(x, lin.weight, lin.bias, act)
            else:
                x = ops.LinearActFn.apply(x, lin.weight, lin.bias, 0)
                x = batch_norm_act(bn, x, act, self.training)
        return x.reshape(*lead, x.shape[-1])
