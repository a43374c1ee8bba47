"""Oracle (test infrastructure): CPU restatement of one Trainer step and of the dense optimizers.

Reference files followed (relative to /root/reference):
  train/trainer.py:62-68     -> ``make_optimizer`` (torch.optim.{Adam,Adagrad,AdamW}(params, lr, weight_decay))
  train/trainer.py:204-223   -> ``train_step``     (forward, rec loss + reg loss, backward, step, zero_grad)
``adamw_update`` / ``adam_update`` / ``adagrad_update`` spell out the torch update rules (torch defaults
beta=(0.9, 0.999), eps=1e-8; Adagrad eps=1e-10, lr_decay 0) so that the fused HIP optimizer step can be
checked element-wise; they are themselves checked against torch.optim in tests.
"""
from __future__ import annotations

import math
from typing import Dict

import torch


def make_optimizer(name: str, params, lr: float, wd: float):
    opt = {'adam': torch.optim.Adam, 'adagrad': torch.optim.Adagrad, 'adamw': torch.optim.AdamW}[name]
    return opt(params, lr=lr, weight_decay=wd)


def adamw_update(p, g, m, v, step: int, lr: float, wd: float, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor rule: decoupled decay, then Adam with bias correction."""
    p = p * (1 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


def adam_update(p, g, m, v, step: int, lr: float, wd: float, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam: L2 term added to the gradient."""
    g = g + wd * p
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


def adagrad_update(p, g, s, step: int, lr: float, wd: float, eps=1e-10):
    """torch.optim.Adagrad (lr_decay = 0, initial accumulator 0)."""
    g = g + wd * p
    s = s + g * g
    p = p - lr * g / (s.sqrt() + eps)
    return p, s


def train_step(net, rec_loss, optimizer, u_idx, i_idx, labels, user_mods=None, item_mods=None) -> Dict[str, float]:
    """trainer.py:209-223 for a ``RefSingleBranchNet`` whose state_dict leaves are the optimizer's params."""
    logits = net.forward(u_idx, i_idx, True, user_mods, item_mods)
    rl = rec_loss.compute_loss(logits, labels)
    reg = net.get_and_reset_other_loss()
    total = rl + reg['reg_loss']
    total.backward()
    optimizer.step()
    optimizer.zero_grad()
    return {'loss': float(total.detach()), 'rec_loss': float(rl.detach()), 'reg_loss': float(reg['reg_loss'].detach().sum()),
            'logits': logits.detach()}
