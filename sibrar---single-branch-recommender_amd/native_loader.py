"""Python side of the native batch producer (csrc/producer.hip): one C++ thread collates, draws the modalities, plans and uploads
the batches of an epoch ahead of the launch thread; this module hands it the generator states (numpy's global legacy MT19937, the
entities' PCG64 generators) when the epoch starts, turns its descriptors into ``engine.PreparedBatch`` objects and puts the
states back when the epoch ends — so the random streams continue exactly as if the reference's own loader
(data/dataloader.py:134-198) and modality sampler (algorithms/sgd_alg.py:1904-1927) had produced the batches.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np
import torch

from ._lib import SibrarHipError, lib, stream

MASK62 = 0x3FFFFFFFFFFFFFFF
_REG_K = {'no_regularization': (1, False), 'pairwise_single': (2, False), 'central_modality': (2, True)}


def _entity_spec(ent):
    """-> (n_mod, k, central position | -1) of a SingleBranchNetEntity's training draw, or None when the producer cannot draw it."""
    order = ent.train_modality_order
    k, central = _REG_K.get(ent._reg_type.value, (None, None))
    if k is None or len(order) > 8 or k > len(order):
        return None
    c = -1
    if central:
        if ent.entity_config.central_modality not in order or len(order) < 2:
            return None
        c = list(order).index(ent.entity_config.central_modality)
    return len(order), k, c


def _pcg_words(rng: np.random.Generator):
    st = rng.bit_generator.state
    if st.get('bit_generator') != 'PCG64':
        return None
    s, inc = int(st['state']['state']), int(st['state']['inc'])
    m = (1 << 64) - 1
    return [s >> 64, s & m, inc >> 64, inc & m, int(st['has_uint32']), int(st['uinteger'])]


def _set_pcg(rng: np.random.Generator, w):
    st = rng.bit_generator.state
    st['state']['state'] = (int(w[0]) << 64) | int(w[1])
    st['has_uint32'], st['uinteger'] = int(w[4]), int(w[5])
    rng.bit_generator.state = st


def eligible(loader, fused) -> bool:
    """The producer covers the default path: uniform_recbole collate-level sampling with the membership test on the device,
    one rank or rank-local sampling, entities with at most 8 modalities and the three regularisation modes."""
    if os.environ.get('SBR_NATIVE_PRODUCER', '1') == '0' or loader._device is None or loader.dataset_sampler:
        return False
    if loader.strategy != 'uniform_recbole' or (loader.world > 1 and loader.dp_sampling != 'local'):
        return False
    if not hasattr(loader.positives, 'indptr') or len(loader.dataset.items_in_split) - 1 > 0xFFFFFFFF:
        return False
    net = fused.net
    if _entity_spec(net.item_embedding_module) is None:
        return False
    if net.is_user_sb_module and _entity_spec(net.user_embedding_module) is None:
        return False
    for ent in ([net.user_embedding_module] if net.is_user_sb_module else []) + [net.item_embedding_module]:
        if _pcg_words(ent._rng) is None:
            return False
    return np.random.get_state()[0] == 'MT19937'


class NativeBatchProducer:
    N_SLOTS = 8

    def __init__(self, loader, fused):
        from .engine import FusedTrainStep
        self.loader, self.fused = loader, fused
        net = fused.net
        dev = loader._device
        self.device = dev
        self.B, self.n_neg = int(loader.batch_size), int(loader.n_neg)
        self.N = 1 + self.n_neg
        self.ents = [net.user_embedding_module if net.is_user_sb_module else None, net.item_embedding_module]
        self.specs = [(_entity_spec(e) if e is not None else None) for e in self.ents]
        ku = self.specs[0][1] if self.specs[0] else 0
        ki = self.specs[1][1]
        B, N = self.B, self.N
        self.slot_bytes = (B + 1) * 8 + (B * N + 1) * 8 + B * ku + B * N * ki + 8 + 6 * 16
        self.slots = [torch.empty(self.slot_bytes, dtype=torch.uint8, device=dev) for _ in range(self.N_SLOTS)]
        pos = loader.positives
        items = np.asarray(loader.dataset.items_in_split)
        self._items = None if loader._identity_items else np.ascontiguousarray(items, dtype=np.int64)
        slot_ptrs = (ctypes.c_void_p * self.N_SLOTS)(*[t.data_ptr() for t in self.slots])
        self._keep = (pos._h_indptr, pos._h_indices, pos.indptr, pos.indices, slot_ptrs)
        L = lib()
        self.handle = L.sbr_producer_create(
            dev.index if dev.index is not None else torch.cuda.current_device(), B, self.n_neg, len(items),
            None if self._items is None else self._items.ctypes.data, pos._h_indptr.ctypes.data, pos._h_indices.ctypes.data,
            pos.indptr.data_ptr(), pos.indices.data_ptr(), int(pos.HOST_BELOW), 1 if fused.use_graph else 0, self.N_SLOTS,
            self.slot_bytes, ctypes.cast(slot_ptrs, ctypes.c_void_p))
        if not self.handle:
            raise SibrarHipError(L.sbr_last_error().decode())
        for w, spec in enumerate(self.specs):
            n_mod, k, c = spec if spec else (0, 0, -1)
            self._check(L.sbr_producer_set_entity(self.handle, w, 1 if spec else 0, n_mod, k, c))
        self._live = False
        self._desc = (ctypes.c_long * 32)()

    @staticmethod
    def _check(rc):
        if rc != 0:
            raise SibrarHipError(lib().sbr_last_error().decode())

    # ---- epoch --------------------------------------------------------------------------------------------------------------
    def start(self, rows_e: np.ndarray, cols_e: np.ndarray, first: int, stride: int, n_batches: int):
        self.stop()
        self._rows_e, self._cols_e = np.ascontiguousarray(rows_e, dtype=np.int64), np.ascontiguousarray(cols_e, dtype=np.int64)
        st = np.random.get_state()
        key = np.ascontiguousarray(st[1], dtype=np.uint32)
        pcg = (ctypes.c_ulonglong * 12)()
        for w, ent in enumerate(self.ents):
            if ent is not None:
                pcg[6 * w:6 * w + 6] = _pcg_words(ent._rng)
        seed_base = (torch.initial_seed() * 1000003) & MASK62
        self._np_state = st
        self._check(lib().sbr_producer_start(self.handle, self._rows_e.ctypes.data, self._cols_e.ctypes.data, len(self._rows_e),
                                             int(first), int(stride), int(n_batches), key.ctypes.data, int(st[2]),
                                             ctypes.cast(pcg, ctypes.c_void_p), seed_base, int(self.fused._n_prepared)))
        self._live = True

    def stop(self):
        """Stops the thread (if an epoch runs) and puts the generator states back where the producer left them."""
        if not self._live:
            return
        self._live = False
        key = np.empty(624, dtype=np.uint32)
        pos = ctypes.c_int(0)
        pcg = (ctypes.c_ulonglong * 12)()
        counters = (ctypes.c_long * 2)()
        rc = lib().sbr_producer_stop(self.handle, key.ctypes.data, ctypes.byref(pos), ctypes.cast(pcg, ctypes.c_void_p),
                                     ctypes.cast(counters, ctypes.c_void_p))
        st = self._np_state
        np.random.set_state((st[0], key, pos.value, st[3], st[4]))
        for w, ent in enumerate(self.ents):
            if ent is not None:
                _set_pcg(ent._rng, pcg[6 * w:6 * w + 6])
        self.fused._n_prepared = int(counters[1])
        self._check(rc)

    def close(self):
        if self.handle:
            self.stop()
            lib().sbr_producer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- consumer side ----------------------------------------------------------------------------------------------------------
    def wait(self, slot: int):
        self._check(lib().sbr_producer_wait(self.handle, slot, stream()))

    def release(self, slot: int):
        self._check(lib().sbr_producer_release(self.handle, slot, stream()))

    def _labels(self, B: int):
        cache = self.fused._label_cache
        key = ('first_column_positive', (B, self.N))
        lab = cache.get(key)
        if lab is None:
            host = torch.zeros(B, self.N, dtype=torch.float64)
            host[:, 0] = 1.
            lab = cache[key] = host.reshape(-1).to(self.device)
        return lab

    def next_batch(self):
        """-> PreparedBatch of the next batch, or None at the end of the epoch. Blocks without the interpreter lock."""
        from .engine import FusedTrainStep, PreparedBatch
        d = self._desc
        rc = lib().sbr_producer_next(self.handle, ctypes.cast(d, ctypes.c_void_p))
        if rc == 1:
            return None
        self._check(rc)
        slot, B, nbytes = int(d[0]), int(d[1]), int(d[2])
        N = self.N
        Ru, Ri = int(d[9]), int(d[10])
        packed = self.slots[slot][:nbytes]
        layout = ((int(d[3]), (B + 1) * 8), (int(d[4]), (B * N + 1) * 8), (int(d[5]), 0), (int(d[5]), Ru), (int(d[6]), Ri), (int(d[7]), 8))
        pb = PreparedBatch()
        pb.packed, pb.layout = packed, layout
        pb.u, pb.i, _, pb.su, pb.si, pb.seed = FusedTrainStep._views(packed, layout, self.specs[0] is not None)
        pb.lab, pb.lab_cached = self._labels(B), True
        pad = bool(self.fused.use_graph)
        pb.pu = None
        if self.specs[0] is not None:
            n_mod, k, _ = self.specs[0]
            pb.pu = (None, k, tuple(int(d[11 + m]) for m in range(n_mod)), tuple(self.ents[0].train_modality_order), Ru, pad)
        n_mod, k, _ = self.specs[1]
        pb.pi = (None, k, tuple(int(d[19 + m]) for m in range(n_mod)), tuple(self.ents[1].train_modality_order), Ri, pad)
        pb.u_shape, pb.i_shape = (B,), (B, N)
        pb.event = None
        pb.native = (self, slot)
        return pb
