"""Trainer with the reference's constructor, ``fit`` / ``train`` / ``val`` hooks and result dictionaries
(train/trainer.py:18-256), driving the HIP engine:

  * one fused dense optimizer launch per step over the flat parameter buffer (optim.FusedOptimizer) instead of
    torch.optim's per-tensor kernels;
  * losses are accumulated on the device; the three ``.item()`` host syncs per step of trainer.py:217-219 happen once per
    epoch instead;
  * data-parallel: when torch.distributed is initialised the flat gradient buffer is all-reduced over RCCL before the
    optimizer step (parallel.all_reduce_grads).
W&B / Ray reporting of the reference is experiment management and is not reproduced.
"""
from __future__ import annotations

import logging
import sys

import torch

from . import parallel
from .evaluation import FullEvaluator, evaluate_recommender_algorithm
from .optim import FusedOptimizer


def _get(o, k, default=None):
    if o is None:
        return default
    return o.get(k, default) if isinstance(o, dict) else getattr(o, k, default)


class Trainer:
    def __init__(self, model, train_loader, val_loader, rec_loss, conf, train_val_loader=None):
        self.full_conf = conf
        self.train_loader, self.val_loader, self.train_val_loader = train_loader, val_loader, train_val_loader
        self.evaluate_train_loader = train_val_loader is not None
        if (train_val_loader is None) != (_get(conf, 'train_eval') is None):
            raise ValueError('Either both, a validation loader for the train set `train_val_loader` '
                             'and its validation configuration `conf.train_eval` must be specified, or neither one!')
        learn = _get(conf, 'learn')
        self.learning_config = learn
        self.device = _get(_get(conf, 'run_settings'), 'device', 'cuda')
        self.model = model
        self.pointer_to_model = model
        self.model.to(self.device)
        self.rec_loss = rec_loss
        self.lr, self.wd = _get(learn, 'lr', 1e-3), _get(learn, 'wd', 0.)
        self.optimizer = FusedOptimizer(self.model, _get(learn, 'optimizer', 'adam'), lr=self.lr, weight_decay=self.wd)
        self.n_epochs = _get(learn, 'n_epochs', 50)
        self.optimizing_metric = _get(learn, 'optimizing_metric', 'ndcg@10')
        self.max_patience = _get(learn, 'max_patience', sys.maxsize)
        self.max_batches = _get(learn, 'max_batches_per_epoch')
        self.model_path = _get(conf, 'results_path')
        self.batch_verbose = _get(_get(conf, 'run_settings'), 'batch_verbose', False)
        self.scorer = _get(conf, 'scorer', 'fp32')
        self.best_value = self.best_metrics = self.best_epoch = None
        # the fused launch choreography (engine.FusedTrainStep) replaces autograd when the model is a SingleBranchNet with
        # an entity item side; `conf.fused_step = False` keeps the autograd path (same kernels, same results)
        self.fused = None
        if _get(conf, 'fused_step', True):
            try:
                from .engine import FusedTrainStep
                from .sbnet import SingleBranchNet
                if isinstance(model, SingleBranchNet):
                    self.fused = FusedTrainStep(model, rec_loss, self.optimizer)
            except NotImplementedError:
                self.fused = None
        # a NegativeSamplingDataLoader of this package without its own hooks gets the fused step's batch preparation (draw,
        # plan, packed upload on the loader thread) and a prefetch depth; any other loader is consumed as it is
        if self.fused is not None and train_loader is not None:
            from .datasets import NegativeSamplingDataLoader
            if isinstance(train_loader, NegativeSamplingDataLoader) and train_loader.prepare_fn is None \
                    and train_loader.draw_fn is None:
                train_loader.prepare_fn = self.fused.prepare
                if train_loader.prefetch <= 0:
                    train_loader.prefetch = 4
        # an epoch that stops after `max_batches_per_epoch` (trainer.py:225-227) must not leave a producer running ahead on the
        # random streams: the loader is told the cap (its producers then stop exactly there) while the loss averages keep the
        # reference's denominator, the uncapped number of batches (trainer.py:233)
        self._n_batches_uncapped = None
        if train_loader is not None and self.max_batches is not None and hasattr(train_loader, 'max_batches'):
            self._n_batches_uncapped = len(train_loader)
            if train_loader.max_batches is None or train_loader.max_batches > self.max_batches:
                train_loader.max_batches = self.max_batches
        logging.info(f'Built Trainer module - optimizer: {self.optimizer.name} lr: {self.lr} wd: {self.wd}')

    def fit(self):
        current_patience = self.max_patience
        log_dict = self.val()
        self.best_value = log_dict['max_optimizing_metric'] = log_dict[self.optimizing_metric]
        self.best_epoch = log_dict['best_epoch'] = -1
        self.best_metrics = log_dict
        print(f'Init - {self.optimizing_metric}={self.best_value:.4f}')
        if self.model_path:
            self.pointer_to_model.save_model_to_path(self.model_path)
        for epoch in range(self.n_epochs):
            self.model.train()
            if current_patience == 0:
                print('Ran out of patience, stopping ')
                break
            epoch_losses = self.train()
            print(f'Epoch [{epoch:>3d}|{self.n_epochs:>d}] - average train loss {epoch_losses["train/loss"]:.4f} '
                  f'({epoch_losses["train/rec_loss"]:.4f} recommendation loss + {epoch_losses["train/reg_loss"]:.4f} '
                  f'regularization loss)')
            if self.evaluate_train_loader:
                epoch_losses.update(**self.train_val())
            metrics_values = self.val()
            curr_value = metrics_values[self.optimizing_metric]
            if curr_value > self.best_value:
                self.best_value = metrics_values['max_optimizing_metric'] = curr_value
                self.best_epoch = metrics_values['best_epoch'] = epoch
                self.best_metrics = metrics_values
                if self.model_path:
                    self.pointer_to_model.save_model_to_path(self.model_path)
                current_patience = self.max_patience
            else:
                metrics_values['max_optimizing_metric'] = self.best_value
                current_patience -= 1
        return self.best_metrics

    def train(self):
        return self._train()

    def train_step(self, u_idxs, i_idxs, labels, draws=None):
        """trainer.py:205-223 for one batch; returns the device-side loss tensors (no host sync)."""
        if self.fused is not None:
            total, rec, reg = self.fused.step(u_idxs, i_idxs, labels, draws)
            return total, rec, {'reg_loss': reg}
        from ._lib import to_device
        u_idxs, i_idxs, labels = (to_device(t, self.device) for t in (u_idxs, i_idxs, labels))
        out = self.model(u_idxs, i_idxs)
        rec_loss = self.rec_loss.compute_loss(out, labels)
        reg_losses = self.pointer_to_model.get_and_reset_other_loss()
        reg_loss = reg_losses['reg_loss'].to(rec_loss.device)
        total_loss = rec_loss + reg_loss
        total_loss.backward()
        from . import parallel
        parallel.all_reduce_grads(self.optimizer)
        self.optimizer.step()
        self.optimizer.zero_grad()
        return total_loss.detach(), rec_loss.detach(), {k: v.detach() for k, v in reg_losses.items()}

    def _train(self):
        self.model.train()
        sums = {}
        n_batches = self._n_batches_uncapped if self._n_batches_uncapped is not None else len(self.train_loader)
        acc3 = None                                    # fused step: (loss, rec_loss, reg_loss) accumulated by ONE launch per step
        for batch_count, batch in enumerate(self.train_loader):
            total, rec, regs = self.train_step(*batch)
            out3 = getattr(self.fused, 'last_out3', None) if self.fused is not None else None
            if out3 is not None and set(regs) == {'reg_loss'}:
                acc3 = out3.clone() if acc3 is None else acc3.add_(out3)
            else:
                vals = {'loss': total, 'rec_loss': rec, **regs}
                for k, v in vals.items():
                    v = v.double().sum()
                    sums[k] = v if k not in sums else sums[k] + v
            if self.max_batches is not None and self.max_batches <= batch_count + 1:
                print(f'limit of {self.max_batches} batches hit, thus stopping this training cycle.')
                if hasattr(self.train_loader, 'close'):
                    self.train_loader.close()                                    # stop producers before anything else draws
                break
        if hasattr(self.pointer_to_model, 'check_index_errors'):
            self.pointer_to_model.check_index_errors()                               # ids without a feature row -> KeyError
        if acc3 is not None:
            for k, v in zip(('loss', 'rec_loss', 'reg_loss'), acc3.unbind(0)):
                sums[k] = v if k not in sums else sums[k] + v
        return {f'train/{k}': float(v) / n_batches for k, v in sums.items()}        # one host sync per epoch

    @torch.no_grad()
    def _eval_loader(self, loader, config, evaluator_name: str = None):
        self.model.eval()
        evaluator = FullEvaluator(config=config, evaluator_name=evaluator_name, dataset=loader.dataset)
        # a data-parallel Trainer evaluates on every rank (fit() is collective): the catalogue is item-sharded over the ranks
        return evaluate_recommender_algorithm(self.pointer_to_model, loader, evaluator, self.device,
                                              verbose=self.batch_verbose, scorer=self.scorer, shard_items=parallel.is_distributed())

    def train_val(self):
        return self._eval_loader(self.train_val_loader, _get(self.full_conf, 'train_eval'), 'train')

    def val(self):
        return self._eval_loader(self.val_loader, _get(self.full_conf, 'eval'))
