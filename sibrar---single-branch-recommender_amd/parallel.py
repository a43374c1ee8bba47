"""Multi-GPU paths (new functionality — the reference has no distributed code):

* training is data parallel over the interaction minibatch: every rank holds a full replica, draws the SAME global batch
  from the same host RNG streams, keeps its slice ``[rank::world]`` and, before the optimizer step, sums the ONE flat fp32
  gradient buffer over RCCL (xGMI) and divides by the world size — the gradient of the mean loss over the global batch.
  BatchNorm statistics stay rank-local (what DDP does by default; SURVEY.md §7 hard parts). The mean over ranks of the
  per-rank mean-loss gradients is the gradient of the global-batch mean because the loaders give every rank the same number
  of rows (an incomplete last global batch loses its surplus rows / is dropped: datasets.NegativeSamplingDataLoader).
* full-catalogue scoring shards by item: every rank scores all users against its item shard with the fused kernel, the
  per-shard top-k lists ``(score f32, global item idx i32)`` are all-gathered and merged (k-way, exact).
One process per GPU; ``backend='nccl'`` is RCCL on ROCm, ``gloo`` is used by the CPU tests.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def is_distributed() -> bool:
    """True when gradients / top-k lists are exchanged. ``SBR_FORCE_DIST=1`` turns the exchange on for a one-rank process
    group as well (test aid: the RCCL call sequence of a data-parallel step on a one-GPU box; sum over one rank / 1)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('SBR_FORCE_DIST', '0') == '1'


def shard_batch(u_idxs, i_idxs, labels, rank: int = None, world: int = None):
    """Rank r keeps rows r, r+W, r+2W, ... of the global batch (all ranks drew the same batch)."""
    if rank is None:
        rank, world = (dist.get_rank(), dist.get_world_size()) if is_distributed() else (0, 1)
    if world == 1:
        return u_idxs, i_idxs, labels
    return u_idxs[rank::world], i_idxs[rank::world], labels[rank::world]


def all_reduce_flat_(grad: torch.Tensor) -> torch.Tensor:
    """In-place mean over ranks of one flat gradient buffer (single collective)."""
    if is_distributed():
        dist.all_reduce(grad, op=dist.ReduceOp.SUM)
        grad.div_(dist.get_world_size())
    return grad


def world_size() -> int:
    return dist.get_world_size() if is_distributed() else 1


def all_reduce_async(t: torch.Tensor):
    """Starts the SUM all-reduce of a contiguous slice of the flat gradient buffer; returns the work handle (``wait()`` makes
    the current stream wait for it) or None when not distributed. The caller divides by the world size."""
    if not is_distributed():
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)


def all_reduce_grads(optimizer) -> None:
    """Called by Trainer.train_step between backward and step. ``optimizer`` is an optim.FusedOptimizer."""
    if not is_distributed():
        return
    optimizer._sync_grads()
    all_reduce_flat_(optimizer.fp.grad)


def item_shard(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous item range [lo, hi) of a rank."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def merge_topk(val: torch.Tensor, idx: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Exact k-way merge of per-shard top-k lists. val/idx: [Bu, W*k] (concatenated shards, idx global, -1 = empty slot).
    Order: score descending, then item index ascending — the same rule the single-GPU kernels use."""
    v = val.double()
    v = torch.where(idx < 0, torch.full_like(v, -float('inf')), v)
    # composite sort: primary score desc, secondary index asc (stable sort on the secondary key first)
    order = torch.argsort(idx, dim=1, stable=True)
    v2, i2 = torch.gather(v, 1, order), torch.gather(idx, 1, order)
    order2 = torch.argsort(v2, dim=1, descending=True, stable=True)
    v3, i3 = torch.gather(v2, 1, order2), torch.gather(i2, 1, order2)
    return v3[:, :k].float(), i3[:, :k]


def all_gather_topk(val: torch.Tensor, idx: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather every rank's [Bu, k] shard lists and merge them; every rank ends with the global top-k. On the GPU the lists
    are gathered into one [W, Bu, k] buffer each and merged by ``sbr_merge_topk`` (one wave per user); the torch formulation
    (``merge_topk``) serves CPU tensors (gloo tests) and W * k > 256."""
    if not is_distributed():
        return val, idx
    world = dist.get_world_size()
    if val.is_cuda and world * k <= 256 and idx.dtype == torch.int32:
        from ._lib import call, ptr, stream
        Bu = val.shape[0]
        vals = torch.empty(world, Bu, k, device=val.device, dtype=torch.float32)
        idxs = torch.empty(world, Bu, k, device=val.device, dtype=torch.int32)
        dist.all_gather_into_tensor(vals.view(-1), val.contiguous().float().view(-1))
        dist.all_gather_into_tensor(idxs.view(-1), idx.contiguous().view(-1))
        out_val, out_idx = torch.empty_like(val, dtype=torch.float32), torch.empty_like(idx)
        call('sbr_merge_topk', ptr(vals), ptr(idxs), world, Bu, k, ptr(out_val), ptr(out_idx), stream())
        return out_val, out_idx
    vals = [torch.empty_like(val) for _ in range(world)]
    idxs = [torch.empty_like(idx) for _ in range(world)]
    dist.all_gather(vals, val.contiguous())
    dist.all_gather(idxs, idx.contiguous())
    return merge_topk(torch.cat(vals, dim=1), torch.cat(idxs, dim=1), k)
