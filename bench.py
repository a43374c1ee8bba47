#!/usr/bin/env python3
"""bench.py — training interactions/s (+ full-catalogue scores/s) of the HIP SingleBranchNet engine on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL. Either an external launcher provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
(``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``), or — WORLD_SIZE unset — this script starts
the N rank processes itself before touching the GPU (``launch_ranks``). A process group whose size differs from ``--gpus``
or whose backend is not RCCL is a hard error.

Workload (BASELINE.json configs[1], "c2"): synthetic 100k users x 50k items, ~5M interactions, one 768-d dense item
modality + item-id embedding, common/shared dim 128, hidden [128], sampled-softmax loss, 10 negatives, AdamW(1e-3, 1e-6);
user side = embedding lookup. A "step" is one full training batch through the reference's hot loop
(train/trainer.py:204-223): draw the batch (shuffled epoch order + bit-exact negative sampling), draw the modalities,
forward, sampled-softmax loss, backward, dense AdamW over every parameter. Features, tables, parameters and optimizer
state are resident in HBM before the timed region; only the index tensors of each batch cross PCIe.

Prints ONE JSON line (rank 0). ``value`` = interactions (positive rows) per second over all ranks, weak scaling (per-GPU
batch fixed). Extra objects: ``roofline`` (dominant training kernel, HIP-event timed live), ``scoring`` (fused fp16
score+mask+top-k over the full catalogue: scores/s and its own roofline), ``cpu_baseline`` (the CPU oracle restatement of
the same step timed on this box's host cores, rank 0, N=1 only), ``b256`` (the same step at the reference's default batch).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_MFMA_F32 = 157.3     # TFLOP/s, MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_MFMA_F16 = 2500.0    # TFLOP/s dense, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
PEAK_HBM = 8000.0         # GB/s
SETTLE = 30               # extra untimed steps after --warmup (see bench_training)
TABLES = {}               # per-kernel tables: written to a side file (bench_tables.json), not into the one JSON line
LAST_LOSS = {}            # batch size -> total loss of the last timed step
EXCHANGE = {}             # batch size -> how the user-table gradient was exchanged (data-parallel runs)

C2 = dict(n_users=100_000, n_items=50_000, nnz=5_000_000, feat_dim=768, emb_dim=128, n_neg=10)


def model_config(emb_dim):
    return {'shared_common_dim': emb_dim, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
            'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'item_embedding'}],
                     'single_branch_hidden_layers': [emb_dim], 'preference_hidden_layers': [], 'common_modality_dim': emb_dim}}


def build(S, cfg, device, seed=0):
    ds = S.SyntheticDataset(cfg['n_users'], cfg['n_items'], cfg['nnz'], item_dense={'text': cfg['feat_dim']}, seed=seed,
                            n_negative_samples=cfg['n_neg'], negative_sampling_strategy='uniform_recbole',
                            holdout_per_user=0)
    torch.manual_seed(42)
    np.random.seed(42)
    net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(model_config(cfg['emb_dim'])), ds).to(device)
    return ds, net


class _Conf:
    """the parts of ExperimentConfig the Trainer reads (data/config_classes.py:198-248)"""
    def __init__(self, device):
        self.learn = {'lr': 1e-3, 'wd': 1e-6, 'optimizer': 'adamw', 'n_epochs': 1, 'optimizing_metric': 'ndcg@10'}
        self.run_settings = {'device': device, 'batch_verbose': False}
        self.results_path = None
        self.eval = None
        self.train_eval = None


def epochs(loader):
    """Batches of consecutive epochs (a long --steps run crosses the epoch boundary like Trainer.fit does)."""
    while True:
        yield from loader


def run_steps(S, trainer, loader_iter, n, world):
    out = None
    for _ in range(n):
        out = trainer.train_step(*next(loader_iter))
    return out


def bench_training(S, ds, net, device, batch, steps, warmup, rank, world, time_kernels, loss=None):
    import torch.distributed as dist
    if loss is None:
        loss = S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole',
                                       neg_train=ds.n_negative_samples)
    trainer = S.Trainer(net, None, None, loss, _Conf(device))
    net.train()
    # three Python threads hand batches to each other (collate -> prepare -> launch). With CPython's default 5 ms switch
    # interval a waiting thread can sit behind the GIL for longer than a whole small-batch step; measured best: 0.2 ms at
    # B = 256 (0.36 vs 0.59 ms per step), 1 ms at B = 8192.
    sys.setswitchinterval(2e-4 if batch <= 1024 else 1e-3)
    # weak scaling: every rank collates its own per-GPU batch (dp_sampling='local': contiguous slice of the shared epoch order,
    # rank-seeded negative stream) — the bit-exact 'global' mode makes every rank draw the whole global batch on its host
    np.random.seed(42 + rank)
    loader = S.NegativeSamplingDataLoader(ds, batch_size=batch, shuffle=True, rank=rank, world=world, device=device,
                                          dp_sampling='local',
                                          prefetch=4, prepare_fn=trainer.fused.prepare if trainer.fused is not None else None)
    it = epochs(loader)
    # W warm-up steps (graph capture, allocator), then SETTLE more untimed steps: the loader's two pipeline stages run up to
    # 2 * prefetch + 2 batches ahead while the first steps compile / capture, and a timed region that starts on that head start
    # would report the consumer's burst rate instead of the sustained rate of the whole pipeline.
    run_steps(S, trainer, it, warmup, world)
    # no cyclic-garbage collection inside the timed region (a generation-2 pass over the loader's queues and tensors is a
    # multi-millisecond pause on the launch thread); collected before, re-enabled after. The collection runs BEFORE the settle steps:
    # a pause of tens of milliseconds right in front of the timed region leaves the GPU idle long enough for its clocks to drop, and
    # the first ~10 ms of the timed steps then run slower (0.61 instead of 0.57 ms per step over 20 steps, tools/lab/host_timeline.py).
    import gc
    gc.collect()
    gc.disable()
    run_steps(S, trainer, it, SETTLE, world)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    last = run_steps(S, trainer, it, steps, world)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    LAST_LOSS[batch] = float(last[0]) if last is not None else None      # sanity of the timed steps (read after the clock)
    if LAST_LOSS[batch] is not None and not (0.0 < LAST_LOSS[batch] < 1e3):
        raise RuntimeError(f'training loss after the timed region is {LAST_LOSS[batch]}: the timed steps did not train')
    timings = {}
    if time_kernels:
        # per-kernel HIP events cannot be recorded inside a hipGraph replay: the next `steps` batches of the same loader are
        # run with plain launches (KernelTimer on switches the graph off) and every GEMM launch is bracketed by events on
        # the launch stream. Outside the timed region; same kernels, same shapes, same data stream.
        # Each of these steps starts behind a ~2 ms spin kernel so that the host has queued the whole step before its first
        # kernel starts: otherwise the GPU idles between plain launches and every bracketed kernel starts "cold" (measured:
        # 55 us instead of the 38 us rocprofv3 reports for the same dispatch inside the graph replay).
        S.ops.KernelTimer.reset(True)
        for _ in range(steps):
            torch.cuda._sleep(5_000_000)
            trainer.train_step(*next(it))
        timings = S.ops.KernelTimer.results()
        S.ops.KernelTimer.reset(False)
    loader.close()
    if trainer.fused is not None:
        sp = trainer.fused._sparse
        EXCHANGE[batch] = (f'all-gather of (row, gradient) pairs, capacity {sp[2]} rows per rank' if sp
                           else 'dense all-reduce') if world > 1 or sp is not None else None
        trainer.fused.close()
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    return dt, timings


def gemm_bytes(mode, M, N, K):
    """Algorithmic HBM bytes of one GEMM launch (fp32 operands read once, result written once; NN in the backward pass also reads the
    activation whose derivative its epilogue applies — counted by the caller where it applies)."""
    return 4.0 * ((M * K + N * K + M * N) if mode != 2 else (K * M + K * N + M * N))


def gemm_price(mode, M, N, K, gathered, avg_ms):
    """Honest pricing of one GEMM signature: the bound is the larger of the time the kernel's OWN matrix pipe needs and the time HBM
    needs for the algorithmic bytes. A bf16-split kernel issues 6 bf16 MFMA terms per fp32 multiply-add, so its pipe time is
    6 * 2 M N K / 2.5 PFLOP/s (not 2 M N K against the fp32 peak: that ratio can exceed 1). ``frac`` = bound time / measured time <= 1
    for any kernel that does the work."""
    sym, what, terms, pipe = gemm_kernel(mode, M, N, K, gathered)
    flop = 2.0 * M * N * K
    t = avg_ms * 1e-3
    peak_pipe = PEAK_MFMA_F16 if pipe == 'bf16' else PEAK_MFMA_F32
    t_pipe = flop * terms / (peak_pipe * 1e12)
    byts = gemm_bytes(mode, M, N, K)
    t_hbm = byts / (PEAK_HBM * 1e9)
    if t_pipe >= t_hbm:
        bound, achieved, peak, unit = 'mfma', flop * terms / t / 1e12, peak_pipe, 'TFLOP/s'
    else:
        bound, achieved, peak, unit = 'hbm', byts / t / 1e9, PEAK_HBM, 'GB/s'
    return {'bound': bound, 'achieved': round(achieved, 2), 'peak': peak, 'unit': unit, 'frac': round(max(t_pipe, t_hbm) / t, 4),
            'pipe': ('bf16 MFMA (v_mfma_f32_32x32x16_bf16), 6 terms per fp32 multiply-add' if pipe == 'bf16'
                     else 'fp32 MFMA (v_mfma_f32_32x32x2_f32)'),
            'pipe_frac': round(t_pipe / t, 4), 'hbm_frac': round(t_hbm / t, 4), 'algorithmic_flop': flop, 'algorithmic_bytes': byts,
            'fp32_equivalent_tflops': round(flop / t / 1e12, 2), 'served_by': sym.strip().rstrip(',').strip(), 'what': what}


def gemm_table(timings, steps):
    """Every GEMM signature of the step, HIP-event timed, priced by ``gemm_price``; by time per step."""
    rows = []
    for key, ts in timings.items():
        if key[0] != 'gemm_f32' or not ts:
            continue
        _, mode, M, N, K, gathered = key
        avg_ms = sum(ts) / len(ts)
        p = gemm_price(mode, M, N, K, gathered, avg_ms)
        rows.append({'kernel': f'{["NT", "NN", "TN"][mode]} M={M} N={N} K={K}' + (' gathered' if gathered else ''),
                     'served_by': p['served_by'], 'launches_per_step': round(len(ts) / steps, 2), 'avg_launch_ms': round(avg_ms, 4),
                     'ms_per_step': round(sum(ts) / steps, 4), 'bound': p['bound'], 'frac': p['frac'], 'pipe_frac': p['pipe_frac'],
                     'hbm_frac': p['hbm_frac'], 'fp32_equivalent_tflops': p['fp32_equivalent_tflops']})
    return sorted(rows, key=lambda r: -r['ms_per_step'])


def gemm_kernel(mode, M, N, K, gathered):
    """-> (kernel symbol as the kernel trace names it, what it is, MFMA terms per fp32 multiply-add, pipe) of the launch that serves
    this GEMM signature — asked of the library's own predicates (the decisions ops.linear_nt / matmul_nn / matmul_tn take)."""
    import importlib
    ops = importlib.import_module('sibrar_amd').ops
    lib = importlib.import_module(ops.__name__.rsplit('.', 1)[0] + '._lib').lib()
    split = bool(getattr(ops, '_SPLIT', False))
    if mode == 2:
        if split and lib.sbr_gemm_tn_split_supported(int(M), int(N), int(K)):
            return (f'void gemm_split_tn_kernel<{"true" if (N // 128) * (M // 128) > 1 else "false"}>', 'bf16-split dW kernel (csrc/gemm_split_tn_f32.hip)', 6, 'bf16')
        return ('void gemm_ring_kernel<1, true, true, 2>', 'fp32 MFMA ring kernel (csrc/gemm_ring_f32.hip)', 1, 'f32')
    if split and M >= getattr(ops, '_SPLIT_MIN_ROWS', 4096):
        if lib.sbr_gemm_split_supported(int(M), int(N), int(K)) and not gathered:
            return (f'void gemm_split_kernel<{mode}, ', 'bf16-split shared-MLP kernel (csrc/gemm_split_f32.hip)', 6, 'bf16')
        if mode == 0 and lib.sbr_gemm_split_proj_supported(int(M), int(N), int(K)):
            return ('gemm_split_proj_kernel', 'bf16-split projector kernel (csrc/gemm_split_f32.hip)', 6, 'bf16')
        if lib.sbr_gemm_split_wide_supported(int(M), int(N), int(K)) and ops.wide_pays(int(M), int(N), int(K)):
            return ('gemm_split_wide_kernel', 'wide bf16-split kernel (csrc/gemm_split_wide_f32.hip)', 6, 'bf16')
    return ('void gemm_ring_kernel<', 'fp32 MFMA ring kernel (csrc/gemm_ring_f32.hip)', 1, 'f32')


PMC_SCORER_PREFIX = '_Z23score_topk_f16_n_kernel'        # fused scorer dispatches in the same PMC passes (the only launches of that kernel there)
# entry point -> kernel symbol of the kernel trace / PMC summaries, for the kernels that are no GEMMs
ENTRY_KERNEL = {'sbr_adam_step_zero_grad': 'void adamw_kernel<true>', 'sbr_adam_step': 'void adamw_kernel<false>',
                'sbr_adam_step_rows': 'adam_step_rows_kernel'}


def csrc_sha16():
    """Hash of the kernel sources the library was built from (the PMC summary records the one it was collected with)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, 'sibrar---single-branch-recommender_amd', 'csrc', '*.h*'))):
        h.update(os.path.basename(p).encode())
        h.update(open(p, 'rb').read())
    return h.hexdigest()[:16]


def _pmc_file(cfg=''):
    """The newest committed PMC summary (profiles/rNN[_c5|_c3]_bench_hbm_traffic.csv, written by tools/profile_round.sh + make_profiles.py)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]{cfg}_bench_hbm_traffic.csv')))
    return files[-1] if files else None


def _pmc_rows(cfg=''):
    """-> (relative path, csrc hash the summary was collected with | None, rows)"""
    import csv
    import re
    path = _pmc_file(cfg)
    if path is None:
        return None, None, []
    lines = open(path).read().splitlines()
    sha = next((m.group(1) for l in lines if l.startswith('#') for m in [re.search(r'csrc_sha16=([0-9a-f]{16})', l)] if m), None)
    return os.path.relpath(path, ROOT), sha, list(csv.reader(l for l in lines if not l.startswith('#')))


def pmc_symbol_traffic(sym, batch, one_shape=True):
    """HBM bytes per launch of a kernel symbol from the committed PMC summary -> (bytes | None, note). The PMC counters cannot be
    read inside this process (they need rocprofv3 around it), so the figure is QUOTED from the summary of the same command; the note
    says which summary, and whether the kernel sources have changed since it was collected."""
    if batch != 8192 or sym is None:
        return None, None
    path, sha, rows = _pmc_rows()
    if path is None:
        return None, None
    hit = [(float(r[2]), float(r[5])) for r in rows if len(r) >= 6 and r[0].startswith(sym)]
    n = sum(h[0] for h in hit)
    if not hit or n == 0 or (one_shape and sym.startswith('void gemm_ring_kernel<1, true') and len(hit) > 1):
        return None, None
    cur = csrc_sha16()
    note = (f'HBM bytes per launch quoted from {path} (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of this command, gfx950 '
            f'correction of MI355X_MICROARCH.md); ' +
            ('collected with the kernel sources of this run' if sha == cur else
             f'STALE: collected with kernel sources {sha or "unrecorded"}, this run has {cur}'))
    return sum(h[0] * h[1] for h in hit) / n * 1e6, note


def pmc_scorer_traffic(n_users, cfg=''):
    """HBM bytes of one fused scoring launch from the PMC passes of the same command: over the c2 catalogue (cfg '') or the c5 shard
    shape ('_c5': bench.py --only c5); None for another shape."""
    path, sha, rows = _pmc_rows(cfg)
    if path is None or n_users != C2['n_users']:
        return None, None
    for r in rows:
        if r[0].startswith(PMC_SCORER_PREFIX):
            cur = csrc_sha16()
            return float(r[5]) * 1e6, path + ('' if sha == cur else f' [STALE: kernel sources {sha or "unrecorded"} vs {cur}]')
    return None, None


def kernel_table(timings, steps):
    """Every C-ABI entry point of the step by time per step (HIP events around every call of K plain-launch steps)."""
    rows = []
    for key, ts in timings.items():
        if key[0] != 'call' or not ts:
            continue
        rows.append({'entry_point': key[1], 'launches_per_step': round(len(ts) / steps, 2), 'avg_launch_ms': round(sum(ts) / len(ts), 4),
                     'ms_per_step': round(sum(ts) / steps, 4)})
    return sorted(rows, key=lambda r: -r['ms_per_step'])


TIMING_NOTE = ('HIP events around every launch over K plain-launch steps run right after the timed region (the timed region replays a '
               'hipGraph, which cannot carry per-kernel events); each of those steps is queued behind a spin kernel so that its '
               'kernels run back to back as they do in the replay')


def dominant_gemm(timings, steps, batch=None):
    """-> roofline dict of the GEMM signature with the largest total time over the timed steps."""
    best = None
    for key, ts in timings.items():
        if key[0] != 'gemm_f32' or not ts:
            continue
        tot = sum(ts)
        if best is None or tot > best[1]:
            best = (key, tot, ts)
    if best is None:
        return None
    (_, mode, M, N, K, gathered), tot, ts = best
    avg_ms = tot / len(ts)
    reduce_ms = 0.0
    if mode == 2:
        # the split-K slabs of all dW products of a step are summed by ONE shared launch (sbr_splitk_reduce_multi): this
        # product is charged its share of that launch by slab bytes, so that `achieved` stays "kernel + its reduction"
        for key, rts in timings.items():
            if key[0] != 'splitk_reduce_multi' or not rts:
                continue
            mine = sum(m * n * z for (m, n, z) in key[1] if (m, n) == (M, N))
            total = sum(m * n * z for (m, n, z) in key[1])
            if mine:
                reduce_ms += (sum(rts) / len(rts)) * mine / total / max(1, sum(1 for (m, n, z) in key[1] if (m, n) == (M, N)))
        avg_ms += reduce_ms
    out = gemm_price(mode, M, N, K, gathered, avg_ms)
    traffic, note = pmc_symbol_traffic(out['served_by'], batch)
    out.update({'traffic': traffic, 'traffic_source': note,
                'kernel': f'{out.pop("what")}: {out["served_by"]}, mode={["NT","NN","TN"][mode]} M={M} N={N} K={K} gather={bool(gathered)}'
                          + (f' + its share ({reduce_ms * 1e3:.1f} us, by slab bytes) of the step\'s shared split-K slab reduction '
                             f'(splitk_reduce_multi_kernel)' if mode == 2 else ''),
                'avg_launch_ms': round(avg_ms, 4), 'launches': len(ts),
                'kernel_ms_per_step': round(tot / steps + reduce_ms * len(ts) / steps, 4), 'timing': TIMING_NOTE})
    return out


def step_roofline(timings, steps, batch, n_params, deferred_table=0, table_dim=0):
    """The ``roofline`` object of the line: the kernel with the largest time per step over ALL kernels of the step (GEMM signatures
    and every other entry point), priced on the resource that bounds it; the largest GEMM rides along as ``dominant_gemm``."""
    gemm = dominant_gemm(timings, steps, batch)
    calls = kernel_table(timings, steps)
    gemm_entries = {'sbr_gemm_f32', 'sbr_gemm_split_f32', 'sbr_gemm_split_bnstats_f32', 'sbr_gemm_split_proj_f32', 'sbr_gemm_wres_f32',
                    'sbr_gemm_nt_splitk_f32', 'sbr_gemm_tn_f32', 'sbr_gemm_tn_f32_slabs'}
    other = [r for r in calls if r['entry_point'] not in gemm_entries]
    out = None
    if other and (gemm is None or other[0]['ms_per_step'] > gemm['kernel_ms_per_step']):
        top = other[0]
        name = top['entry_point']
        if name in ('sbr_adam_step_zero_grad', 'sbr_adam_step', 'sbr_adam_step_rows') and n_params:
            # dense AdamW (train/trainer.py:62-68 -> optimizer.step(), zero_grad()): reads p, g, m, v and writes p, m, v = 28 bytes per
            # parameter (the gradient reset only writes elements that are not +0 already). With the lookup user table updated row by
            # row (sbr_adam_step_rows) only the rows of the batch are touched: 28 bytes per element of at most `batch` rows.
            # (+ the launch's sweep: 1/16 of the table's rows per step read and written without a gradient, 24 bytes per element)
            byts = 28.0 * n_params if name != 'sbr_adam_step_rows' else (28.0 * (n_params - deferred_table + min(batch * table_dim, deferred_table))
                                                                          + 24.0 * deferred_table / 16)
            t = top['avg_launch_ms'] * 1e-3
            traffic, note = pmc_symbol_traffic(ENTRY_KERNEL.get(name), batch)
            out = {'bound': 'hbm', 'achieved': round(byts / t / 1e9, 1), 'peak': PEAK_HBM, 'unit': 'GB/s', 'frac': round(byts / t / 1e9 / PEAK_HBM, 4),
                   'traffic': traffic, 'traffic_source': note, 'algorithmic_bytes': byts,
                   'kernel': f'{ENTRY_KERNEL.get(name)} ({name}): AdamW step + gradient reset + loss read-out over {n_params} parameters' +
                             (f' ({deferred_table} of them a lookup table updated row by row: the batch\'s rows + a 1/16 sweep per step)' if name == 'sbr_adam_step_rows' else '') +
                             ', 28 bytes per touched parameter', 'avg_launch_ms': top['avg_launch_ms'], 'kernel_ms_per_step': top['ms_per_step'],
                   'timing': TIMING_NOTE}
    if out is None:
        out = dict(gemm) if gemm else None
    if out is not None:
        out['selection'] = 'largest time per step over ALL kernels of the step (GEMM signatures and every other entry point)'
        if gemm is not None and out.get('kernel') != gemm.get('kernel'):
            out['dominant_gemm'] = gemm
        TABLES['c2_step_all_gemms'] = gemm_table(timings, steps)
        TABLES['c2_step_all_kernels'] = calls
    return out


def compact_roofline(r):
    """The roofline object of the line without prose: numbers, the kernel's name, where its traffic figure comes from."""
    keep = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'pipe_frac', 'hbm_frac', 'fp32_equivalent_tflops', 'algorithmic_bytes',
            'algorithmic_flop', 'avg_launch_ms', 'launches', 'kernel_ms_per_step', 'gemm_ms_per_step', 'kernel', 'traffic_source')
    out = {k: r[k] for k in keep if k in r and r[k] is not None or k == 'traffic' and k in r}
    if isinstance(out.get('traffic_source'), str):
        out['traffic_source'] = out['traffic_source'].split(' (rocprofv3')[0] + ('; STALE' + out['traffic_source'].split('STALE')[1] if 'STALE' in out['traffic_source'] else '')
    if 'dominant_gemm' in r:
        g = r['dominant_gemm']
        out['dominant_gemm'] = {k: g[k] for k in ('kernel', 'bound', 'frac', 'pipe_frac', 'hbm_frac', 'avg_launch_ms', 'fp32_equivalent_tflops', 'traffic') if k in g}
    out['timing'] = 'HIP events around every launch of K plain-launch steps after the timed region'
    return out


def write_tables():
    """Per-kernel tables of this run (every GEMM signature and entry point of the c2 and c3 steps, priced like ``roofline``) -> side file;
    returns its path relative to the repository (the line only names it)."""
    path = os.path.join(ROOT, 'gpurun_out', 'bench_tables.json')
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            json.dump(TABLES, f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError as e:
        return f'not written ({e})'


def _time_scorer(S, u16, i16, k, users, excl, lo, world, reps, warm):
    """-> (seconds per pass, avg ms of the call's launches with the exclusion mask resident in the scorer's layout, avg ms of a call
    that also converts the exclusion CSR into that layout). The mask of an evaluation split is constant (eval/eval.py:219), the
    product converts it once per split (evaluation.py); the timed passes therefore start with it resident, like the CSR itself."""
    import torch.distributed as dist
    holder = S.ops.ScorerExclusions()
    for _ in range(warm):                                # the chip's clock settles over the first few passes of a burst
        S.ops.score_topk_f16(u16, i16, k, users, excl[0], excl[1], item_offset=lo, exclusions=holder)
    S.ops.KernelTimer.reset(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        val, idx = S.ops.score_topk_f16(u16, i16, k, users, excl[0], excl[1], item_offset=lo, exclusions=holder)
        if world > 1:
            val, idx = S.parallel.all_gather_topk(val, idx, k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / reps
    res = S.ops.KernelTimer.results()
    ts = [t for key, v in res.items() if key[0] == 'score_topk_f16' for t in v]
    S.ops.KernelTimer.reset(True)
    for _ in range(max(reps // 4, 2)):                   # the same call when it also builds the event stream from the CSR
        S.ops.score_topk_f16(u16, i16, k, users, excl[0], excl[1], item_offset=lo)
    tb = [t for key, v in S.ops.KernelTimer.results().items() if key[0] == 'score_topk_f16' for t in v]
    S.ops.KernelTimer.reset(False)
    if world > 1:
        t = torch.tensor([dt], device=u16.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    return dt, sum(ts) / len(ts), sum(tb) / len(tb)


def _time_two_pass(S, u16, i16, k, users, excl, lo, reps=8, warm=4):
    """The same pass through the opt-in two-pass scorer (sbr_score_topk_f16_route(2); same lists bit for bit): HIP-event
    time of the call's four launches, exclusion mask resident. None when the shape does not take that route."""
    if i16.shape[0] < 8192:
        return None
    prev = S.ops.score_topk_route(2)
    try:
        holder = S.ops.ScorerExclusions()
        for _ in range(warm):
            S.ops.score_topk_f16(u16, i16, k, users, excl[0], excl[1], item_offset=lo, exclusions=holder)
        S.ops.KernelTimer.reset(True)
        for _ in range(reps):
            S.ops.score_topk_f16(u16, i16, k, users, excl[0], excl[1], item_offset=lo, exclusions=holder)
        ts = [t for key, v in S.ops.KernelTimer.results().items() if key[0] == 'score_topk_f16' for t in v]
        S.ops.KernelTimer.reset(False)
    finally:
        S.ops.score_topk_route(prev)
    avg = sum(ts) / len(ts)
    flops = 2.0 * u16.shape[0] * i16.shape[0] * u16.shape[1]
    return {'avg_launch_ms': round(avg, 4), 'frac': round(flops / (avg * 1e-3) / 1e12 / PEAK_MFMA_F16, 4)}


def scoring_roofline(n_users, n_items, D, k, avg_ms, traffic=None, traffic_source=None, kernel=None):
    flops = 2.0 * n_users * n_items * D
    achieved = flops / (avg_ms * 1e-3) / 1e12
    return {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': PEAK_MFMA_F16, 'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_MFMA_F16, 4),
            'traffic': traffic, 'traffic_source': traffic_source,
            'algorithmic_bytes': (n_users + n_items) * D * 2 + n_users * k * 8,
            'kernel': kernel or 'fused fp16 scorer: sbr_score_topk_f16 (score_topk_f16_n_kernel + score_topk_finalize_kernel; exclusion mask resident as event stream)', 'avg_launch_ms': round(avg_ms, 4)}


def bench_scoring(S, ds, net, device, rank, world, k=20, reps=20, warm=8):
    """Full-catalogue scoring with the fused fp16 kernel: all users x all items (item-sharded over ranks), top-k."""
    net.eval()
    with torch.no_grad():
        i_repr = net.get_item_representations(torch.arange(ds.n_items, device=device))
        lo, hi = S.parallel.item_shard(ds.n_items, rank, world)
        i16 = S.ops.cast_f16(i_repr[lo:hi].contiguous())
        users = torch.arange(ds.n_users, device=device)
        u16 = S.ops.cast_f16(net.get_user_representations(users))
        excl = S.evaluation._csr_to_device(ds.user_sampling_matrix_train, device)
        dt, avg_ms, build_ms = _time_scorer(S, u16, i16, k, users, excl, lo, world, reps, warm)
        two = _time_two_pass(S, u16, i16, k, users, excl, lo) if world == 1 else None
    tr, src = pmc_scorer_traffic(ds.n_users) if world == 1 else (None, None)
    return {'metric': 'full-catalogue scores/s (fused fp16 score+mask+top-20)', 'value': ds.n_users * ds.n_items / dt,
            'unit': 'scores/s', 'ms_per_pass': round(dt * 1e3, 3), 'users': ds.n_users, 'items': ds.n_items, 'dim': int(i16.shape[1]),
            'sharding': f'items/{world}', 'exclusions': int(excl[1].numel()),
            'exclusion_mask': 'resident in the scorer\'s layout (built once per split by the first evaluation; a call that also converts '
                              f'the CSR takes {build_ms:.3f} ms)', 'ms_per_call_with_mask_conversion': round(build_ms, 4),
            'roofline': scoring_roofline(ds.n_users, hi - lo, int(i16.shape[1]), k, avg_ms, tr, src),
            **({'two_pass_route': two} if two else {})}


def bench_c5_shard(S, device, k=20, reps=12, warm=6):
    """BASELINE configs[4], ONE of its eight item shards at its own size on one GPU: 100k users x 25k items x 256 fp16 N(0, 1)/16
    representations, 50 excluded items per user, top-20 — what every rank of the 8-GPU job runs per 100k users."""
    g = torch.Generator(device='cpu').manual_seed(5)
    U, I, D = 100_000, 25_000, 256
    u16 = (torch.randn(U, D, generator=g) / 16).half().to(device)
    i16 = (torch.randn(I, D, generator=g) / 16).half().to(device)
    rng = np.random.default_rng(5)
    cols = np.sort(rng.integers(0, I, size=(U, 50)), axis=1)
    import scipy.sparse as sp
    m = sp.csr_matrix((np.ones(U * 50, dtype=np.int8), cols.reshape(-1), np.arange(0, U * 50 + 1, 50)), shape=(U, I))
    m.sum_duplicates()
    excl = S.evaluation._csr_to_device(m, device)
    users = torch.arange(U, device=device)
    with torch.no_grad():
        dt, avg_ms, build_ms = _time_scorer(S, u16, i16, k, users, excl, 0, 1, reps, warm)
        two = _time_two_pass(S, u16, i16, k, users, excl, 0)
    return {'two_pass_route': two, 'workload': 'BASELINE configs[4], one of eight item shards: 100k users x 25k items x 256 fp16, 50 exclusions per user, top-20',
            'value': round(U * I / dt, 1), 'unit': 'scores/s', 'ms_per_pass': round(dt * 1e3, 3),
            'ms_per_call_with_mask_conversion': round(build_ms, 4),
            'roofline': scoring_roofline(U, I, D, k, avg_ms, *pmc_scorer_traffic(U, '_c5'))}


def host_cores():
    """CPU cores this process may actually use: min(os.cpu_count(), scheduler affinity, cgroup CPU quota). The GPU box
    exposes every host thread through os.cpu_count() but grants a one-GPU job only a share of them; an OpenMP pool sized to
    the whole machine then thrashes inside the quota (measured: 16 s per oracle step with 256 threads vs ~1.2 s)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(S, ds, net, batch, budget_s=20.0):
    """The CPU oracle restatement (oracle/) of the same training step on this box's host cores: same model parameters,
    same literal per-row sampling calls as the reference, torch-CPU fp32 ops, torch.optim.AdamW."""
    from oracle import model_ref, losses_ref, sampling_ref, train_ref
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    for v in sd.values():
        if v.dtype.is_floating_point:
            v.requires_grad_(True)
    ut = {'user_embedding': model_ref.RefTable('categorical', np.arange(ds.n_users), n_categories=ds.n_users)}
    it = {'text': model_ref.table_from_feature(ds.item_features['text']),
          'item_embedding': model_ref.RefTable('categorical', np.arange(ds.n_items), n_categories=ds.n_items)}
    ref = model_ref.RefSingleBranchNet(sd, model_config(net.config.shared_common_dim), ut, it)
    loss = losses_ref.RefRecLoss('sampled_softmax', n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole',
                                 neg_train=ds.n_negative_samples)
    params = [p for k, p in sd.items() if p.requires_grad and 'running' not in k]
    opt = train_ref.make_optimizer('adamw', params, 1e-3, 1e-6)
    inter = ds.user_sampling_matrix
    positives = [inter.indices[inter.indptr[u]:inter.indptr[u + 1]] for u in range(ds.n_users)]
    coo = ds.interaction_matrix
    rng = np.random.default_rng(0)
    n_steps, t_total, step_s = 0, 0.0, []
    while t_total < budget_s and n_steps < 50:
        sel = rng.integers(0, coo.nnz, size=batch)
        t0 = time.perf_counter()
        u, i, l = sampling_ref.recbole_collate(coo.row[sel], coo.col[sel], ds.n_negative_samples, ds.items_in_split, positives)
        train_ref.train_step(ref, loss, opt, torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(l))
        dt = time.perf_counter() - t0
        step_s.append(round(dt, 3))
        if n_steps > 0 or dt > budget_s / 2:      # first step warms caches / allocators
            t_total += dt
        n_steps += 1
    timed = max(n_steps - 1, 1) if t_total > 0 else 1
    return {'value': round(batch * timed / max(t_total, 1e-9), 1), 'unit': 'interactions/s', 'cores': cores, 'kind': 'port',
            'sample': f'{timed} training steps of batch {batch} on the same c2 synthetic data (CPU oracle restatement, '
                      f'torch {torch.__version__} CPU fp32, {cores} threads)', 'step_seconds': step_s}


# ---- BASELINE configs[0] ("c1"): the reference's own CPU-runnable case, GPU engine and CPU port side by side ----------------------
C1 = dict(n_users=5816, n_items=3299, nnz=651_034, n_neg=10)
C1_MODEL = {'shared_common_dim': 64, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
            'item': {'features': [{'feature_name': 'genres'}, {'feature_name': 'text'}],
                     'single_branch_hidden_layers': [64], 'preference_hidden_layers': [], 'common_modality_dim': 64,
                     'embedding_regularization_type': 'pairwise_single', 'regularization_temperature': 0.1,
                     'regularization_weight': 1e-3, 'normalize_single_branch_input': True}}


C1_TRAIN_STEPS = 300      # recorded batches both sides train on before the evaluation that is compared
C1_GPU_RUNS, C1_CPU_RUNS = 8, 3
C1_SNAP = (1, 10, 100, 300)


def _mean_sd(xs):
    m = sum(xs) / len(xs)
    return m, (sum((x - m) ** 2 for x in xs) / max(len(xs) - 1, 1)) ** 0.5


def bench_c1(S, device, steps):
    """ML-1M-shaped synthetic data (U 5,816, I 3,299, 651k interactions drawn with Zipf(1) item popularity, 18 genre
    tags + 768-d text, C = D = 64, pairwise InfoNCE, BPR, AdamW) — BASELINE.md section 3: GPU interactions/s at the reference's batch
    256 and at 4096; the CPU port (oracle restatement: same torch-CPU ops, same per-row numpy sampling calls as the reference) timed on
    the same inputs and parameters, cores stated; and "matched NDCG@10" as a MEASUREMENT: the GPU engine trains C1_GPU_RUNS times and
    the CPU port C1_CPU_RUNS times on the SAME C1_TRAIN_STEPS recorded batches and modality draws from the SAME initial parameters
    (train/trainer.py:204-223), every trained model is evaluated once (eval/eval.py:205-222), and the two means must lie within twice
    the pooled standard deviation of the runs. Run-to-run differences on one side come from rounding only (GPU: float-atomic scatter
    order; CPU: the runs use different thread counts, i.e. different reduction orders), which is exactly what separates the two sides.
    The relative parameter distance between the sides after 1 / 10 / 100 / 300 steps shows the rounding-level start and its growth."""
    from oracle import eval_ref, losses_ref, model_ref, sampling_ref, train_ref
    ds = S.SyntheticDataset(C1['n_users'], C1['n_items'], C1['nnz'], item_dense={'text': 768}, item_tags={'genres': (18, 3)}, seed=0,
                            n_negative_samples=C1['n_neg'], negative_sampling_strategy='uniform_recbole', holdout_per_user=1,
                            item_popularity=1.0)
    torch.manual_seed(42)
    np.random.seed(42)
    net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(C1_MODEL), ds).to(device)
    sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    bpr = S.RecBayesianPersonalizedRankingLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole',
                                               neg_train=ds.n_negative_samples)
    out = {'workload': 'BASELINE configs[0] shape: synthetic ML-1M (5,816 x 3,299, ~650k interactions, Zipf(1) items, 18 genre tags + 768-d '
                       'text, C = D = 64, hidden [64], pairwise InfoNCE, BPR, 10 negatives, AdamW 1e-3 / 1e-6), user = lookup'}
    cores = host_cores()
    ut = {'user_embedding': model_ref.RefTable('categorical', np.arange(ds.n_users), n_categories=ds.n_users)}
    it = {k: model_ref.table_from_feature(f) for k, f in ds.item_features.items()}
    orders = {'item_train': net.item_embedding_module.train_modality_order, 'item_eval': net.item_embedding_module.eval_modality_order}
    rloss = losses_ref.RefRecLoss('bpr', n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole',
                                  neg_train=ds.n_negative_samples)
    inter = ds.user_sampling_matrix
    positives = [inter.indices[inter.indptr[u]:inter.indptr[u + 1]] for u in range(ds.n_users)]
    coo = ds.interaction_matrix
    ev = ds.eval_view()
    excl, labels = ev.exclude_data.tocsr(), ev.user_sampling_matrix.tocsr()
    import re
    trainable = [k for k, v in sd0.items() if v.dtype.is_floating_point and 'running' not in k]
    # a Linear bias directly in front of a BatchNorm has a mathematically zero gradient (the BatchNorm subtracts the batch mean): what
    # either side computes for it is rounding noise, and Adam turns noise into +-lr steps of either sign from the first step on. The
    # parameter distance is therefore reported with and without those biases.
    def _shadowed(k_):
        m = re.match(r'(.*)layers\.linear_(\d+)\.bias$', k_)
        if not m:
            return False
        pre, i_ = m.group(1), int(m.group(2))
        if f'{pre}layers.batch_norm_{i_}.weight' in sd0:
            return True
        m2 = re.match(r'(.*sb_net\.)(\d+)\.$', pre)
        n_lin = max(int(x) for x in re.findall(re.escape(pre) + r'layers\.linear_(\d+)\.bias', ' '.join(sd0))) if m2 else -1
        return bool(m2) and i_ == n_lin and f'{m2.group(1)}{int(m2.group(2)) + 1}.running_mean' in sd0
    solid = [k for k in trainable if not _shadowed(k)]

    def flat(sd, keys=None):
        return torch.cat([sd[k].detach().double().reshape(-1).cpu() for k in (keys or trainable)])

    solid_ix = torch.cat([torch.full((sd0[k].numel(),), k in solid, dtype=torch.bool) for k in trainable])
    theta0 = flat(sd0)

    def cpu_run(threads, recorded):
        """One training of the CPU port from sd0 (records the batches when ``recorded`` is empty) -> (ref, losses, step times, snapshots)"""
        torch.set_num_threads(threads)
        sd = {k: v.clone() for k, v in sd0.items()}
        for v in sd.values():
            if v.dtype.is_floating_point:
                v.requires_grad_(True)
        ref = model_ref.RefSingleBranchNet(sd, C1_MODEL, ut, it, orders=orders)
        opt = train_ref.make_optimizer('adamw', [p for k, p in sd.items() if p.requires_grad and 'running' not in k], 1e-3, 1e-6)
        record = not recorded
        rng = np.random.default_rng(0)
        np.random.seed(42)
        times, losses, snaps = [], [], {}
        for s_ in range(C1_TRAIN_STEPS):
            t0 = time.perf_counter()
            if record:
                sel = rng.integers(0, coo.nnz, size=256)
                u, i, l = sampling_ref.recbole_collate(coo.row[sel], coo.col[sel], ds.n_negative_samples, ds.items_in_split, positives)
                mods = ref.sides['item'].sample_modalities(i.shape, True)             # the per-row rng.choice calls of utilities/utils.py:69
                recorded.append((u, i, l, mods))
            else:
                u, i, l, mods = recorded[s_]
            losses.append(train_ref.train_step(ref, rloss, opt, torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(l), None, mods)['loss'])
            times.append(time.perf_counter() - t0)
            if s_ + 1 in C1_SNAP:
                snaps[s_ + 1] = flat(sd)
        return ref, sd, losses, times, snaps

    def cpu_eval(ref):
        t0 = time.perf_counter()
        with torch.no_grad():
            i_repr = ref.item_repr(torch.arange(ds.n_items), False)
            nd = []
            for lo in range(0, ds.n_users, 256):
                ub = torch.arange(lo, min(lo + 256, ds.n_users))
                r = eval_ref.evaluate(ref.user_repr(ub, False), i_repr, excl[lo:lo + 256].toarray(), labels[lo:lo + 256].toarray(), ks=(10,))
                nd.append(r['ndcg@10'])
        return float(torch.cat(nd).mean()), time.perf_counter() - t0

    def gpu_eval(scorer):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evaluator = S.FullEvaluator(config=S.evaluation._Cfg(top_k=(10,), metrics=['ndcg'], calculate_std=False), dataset=ev)
        m = S.evaluate_recommender_algorithm(net, type('L', (), {'dataset': ev, 'batch_size': 256})(), evaluator, device, scorer=scorer)
        torch.cuda.synchronize()
        return m['ndcg@10'], time.perf_counter() - t0

    # ---- CPU port: C1_CPU_RUNS trainings (the first records every batch and modality draw and is the timed one)
    recorded, cpu_ndcgs, cpu_threads = [], [], []
    for r in range(C1_CPU_RUNS):
        threads = max(1, cores >> r)                        # e.g. 16, 8, 4: different reduction orders, same arithmetic
        ref, sd, losses, times, snaps = cpu_run(threads, recorded)
        torch.set_num_threads(cores)
        nd, ev_s = cpu_eval(ref)
        cpu_ndcgs.append(nd)
        cpu_threads.append(threads)
        if r == 0:
            cpu_losses, cpu_step, cpu_snaps, cpu_eval_s = losses, sum(times[5:55]) / 50, snaps, ev_s
            cpu_sd = {k: v.detach().clone() for k, v in sd.items()}
    # ---- GPU engine: C1_GPU_RUNS trainings on the same recorded batches and draws from the same initial parameters
    order = list(net.item_embedding_module.train_modality_order)
    lut = {m: q for q, m in enumerate(order)}
    pos_of = [np.vectorize(lut.__getitem__, otypes=[np.int8])(mods).reshape(-1, mods.shape[-1]) for (_, _, _, mods) in recorded]
    gpu_ndcgs, gpu16_ndcgs = [], []
    for r in range(C1_GPU_RUNS):
        net.load_state_dict({k: v.to(device) for k, v in sd0.items()})
        net.train()
        gopt = S.FusedOptimizer(net, 'adamw', lr=1e-3, weight_decay=1e-6)
        fused = S.FusedTrainStep(net, bpr, gopt)
        losses, snaps = [], {}
        for s_, (u, i, l, mods) in enumerate(recorded):
            losses.append(fused.step(torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(l), (None, (pos_of[s_], order)))[0])
            if r == 0 and s_ + 1 in C1_SNAP:
                snaps[s_ + 1] = flat(net.state_dict())
        fused.close()
        torch.cuda.synchronize()
        if r == 0:
            gpu_losses, gpu_snaps = [float(x) for x in losses], snaps
        gpu_eval_t = {}
        for scorer in ('fp32', 'fp16_fused'):
            for rep in range(2 if r == 0 else 1):             # first run: second pass with resident CSRs and warm kernels is the timed one
                gpu_eval_t[scorer] = gpu_eval(scorer)
        gpu_ndcgs.append(gpu_eval_t['fp32'][0])
        gpu16_ndcgs.append(gpu_eval_t['fp16_fused'][0])
        if r == 0:
            gpu_eval_first = dict(gpu_eval_t)
    # the same parameters on both evaluators: the CPU-trained parameters through the GPU evaluation
    net.load_state_dict({k: v.to(device) for k, v in cpu_sd.items()})
    cross, _ = gpu_eval('fp32')
    mc, sc = _mean_sd(cpu_ndcgs)
    mg, sg = _mean_sd(gpu_ndcgs)
    pooled = (((len(cpu_ndcgs) - 1) * sc ** 2 + (len(gpu_ndcgs) - 1) * sg ** 2) / max(len(cpu_ndcgs) + len(gpu_ndcgs) - 2, 1)) ** 0.5
    within = abs(mg - mc) <= 2.0 * pooled
    def _rel(k_, ix=None):
        a, b, z = gpu_snaps[k_], cpu_snaps[k_], theta0
        if ix is not None:
            a, b, z = a[ix], b[ix], z[ix]
        return float(f'{float((a - b).norm() / (b - z).norm().clamp_min(1e-30)):.3g}')
    dist = {str(k): _rel(k) for k in C1_SNAP}
    dist_solid = {str(k): _rel(k, solid_ix) for k in C1_SNAP}
    first = max(abs(a - b) / max(abs(a), 1e-12) for a, b in zip(cpu_losses[:20], gpu_losses[:20]))
    tail_cpu, tail_gpu = sum(cpu_losses[-50:]) / 50, sum(gpu_losses[-50:]) / 50
    if abs(cross - cpu_ndcgs[0]) > 1e-5 + 1e-4 * abs(cpu_ndcgs[0]):
        raise RuntimeError(f'c1: the CPU-trained parameters give NDCG@10 {cpu_ndcgs[0]:.6f} on the CPU evaluator and {cross:.6f} on the GPU evaluator')
    if abs(mg - mc) > 0.05 * abs(mc) and abs(mg - mc) > 5e-4 and not within:
        raise RuntimeError(f'c1: NDCG@10 after {C1_TRAIN_STEPS} identical steps: CPU port {cpu_ndcgs}, GPU engine {gpu_ndcgs}')
    # ---- GPU training throughput (from the initial parameters again)
    net.load_state_dict({k: v.to(device) for k, v in sd0.items()})
    gpu = {}
    for B in (256, 4096):
        n_steps = max(steps, 50)
        dt, _ = bench_training(S, ds, net, device, B, n_steps, 5, 0, 1, time_kernels=False, loss=bpr)
        gpu[f'b{B}'] = {'value': round(B * n_steps / dt, 1), 'unit': 'interactions/s', 'ms_per_step': round(dt / n_steps * 1e3, 3),
                        'steps': n_steps}
    n_scores = ds.n_users * ds.n_items
    out.update({
        'gpu': gpu,
        'cpu': {'value': round(256 / cpu_step, 1), 'unit': 'interactions/s', 'ms_per_step': round(cpu_step * 1e3, 2), 'batch': 256,
                'timed_steps': 50, 'warmup_steps': 5, 'cores': cores, 'kind': 'port'},
        'speedup_vs_cpu': {'b256': round(gpu['b256']['value'] / (256 / cpu_step), 1),
                           'b4096_vs_cpu_b256': round(gpu['b4096']['value'] / (256 / cpu_step), 1)},
        'trained_ndcg': {'what': f'NDCG@10 after the SAME {C1_TRAIN_STEPS} recorded batches of 256 from the same initial parameters: '
                                 f'{C1_GPU_RUNS} GPU trainings, {C1_CPU_RUNS} CPU-port trainings (threads {cpu_threads})',
                         'cpu_runs': [round(x, 6) for x in cpu_ndcgs], 'gpu_runs_fp32_scorer': [round(x, 6) for x in gpu_ndcgs],
                         'gpu_runs_fp16_fused_scorer': [round(x, 6) for x in gpu16_ndcgs],
                         'cpu_mean': round(mc, 6), 'cpu_sd': float(f'{sc:.3g}'), 'gpu_mean': round(mg, 6), 'gpu_sd': float(f'{sg:.3g}'),
                         'abs_diff_of_means': float(f'{abs(mg - mc):.3g}'), 'pooled_sd': float(f'{pooled:.3g}'), 'within_spread': bool(within),
                         'criterion': '|mean_gpu - mean_cpu| <= 2 pooled sd',
                         'same_parameters_both_evaluators': {'cpu_params_cpu_eval': round(cpu_ndcgs[0], 6), 'cpu_params_gpu_eval': round(cross, 6)},
                         'param_distance_gpu_vs_cpu_over_cpu_travel': dist,
                         'param_distance_without_biases_in_front_of_a_batchnorm': dist_solid},
        'loss_curves': {'max_rel_diff_first_20_steps': float(f'{first:.3g}'),
                        'rel_diff_mean_last_50': float(f'{abs(tail_cpu - tail_gpu) / max(abs(tail_cpu), 1e-12):.3g}')},
        'eval': {'scores': n_scores, 'cpu_scores_per_s': round(n_scores / cpu_eval_s, 1), 'cpu_cores': cores,
                 'gpu_fp32_scores_per_s': round(n_scores / gpu_eval_first['fp32'][1], 1),
                 'gpu_fp16_fused_scores_per_s': round(n_scores / gpu_eval_first['fp16_fused'][1], 1)}})
    return out


C3_MODEL = {'shared_common_dim': 128, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
            'item': {'features': [{'feature_name': 'interactions'}, {'feature_name': 'genres'}, {'feature_name': 'audio'}],
                     'single_branch_hidden_layers': [512, 512, 512, 256, 256], 'preference_hidden_layers': [],
                     'common_modality_dim': 512, 'embedding_regularization_type': 'pairwise_single',
                     'regularization_temperature': 0.1, 'regularization_weight': 1e-4}}


def bench_c3(S, device, steps):
    """BASELINE configs[2] (Onion18 shape, conf/single/algorithms/sbnet_onion18_huge_no-user_conf.yml:39-54) on one GPU: 5,192 users x
    13,610 items, CSR interactions + 853-tag bag + 1024-d audio, C = 512, hidden [512, 512, 512, 256, 256], D = 128, two modalities per
    slot with pairwise InfoNCE, BPR, AdamW — interactions/s at the reference's batch 256 and at 4096, with the roofline of the
    dominant GEMM at 4096."""
    ds = S.SyntheticDataset(5192, 13610, 326_000, item_dense={'audio': 1024}, item_tags={'genres': (853, 5)}, seed=0,
                            n_negative_samples=10, negative_sampling_strategy='uniform_recbole')
    torch.manual_seed(42)
    np.random.seed(42)
    net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(C3_MODEL), ds).to(device)
    bpr = S.RecBayesianPersonalizedRankingLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=10)
    out = {'workload': 'BASELINE configs[2] shape: Onion18 (5,192 x 13,610, interactions CSR + 853 tags + audio 1024-d, C = 512, hidden '
                       '[512, 512, 512, 256, 256], D = 128, k = 2 pairwise InfoNCE, BPR), user = lookup, 1 GPU'}
    n_steps = max(min(steps, 60), 30)
    for B in (256, 4096):
        dt, timings = bench_training(S, ds, net, device, B, n_steps, 5, 0, 1, time_kernels=(B == 4096), loss=bpr)
        out[f'b{B}'] = {'value': round(B * n_steps / dt, 1), 'unit': 'interactions/s', 'ms_per_step': round(dt / n_steps * 1e3, 3), 'steps': n_steps}
        if B == 4096 and timings:
            roof = dominant_gemm(timings, n_steps, None)
            if roof:
                TABLES['c3_b4096_all_gemms'] = gemm_table(timings, n_steps)
                TABLES['c3_b4096_all_kernels'] = kernel_table(timings, n_steps)
                gemm_ms = sum(r['ms_per_step'] for r in gemm_table(timings, n_steps))
                roof['gemm_ms_per_step'] = round(gemm_ms, 4)
                out['roofline'] = roof
    return out


def cpu_scoring_sample(S, ds, net, n_users=2048, k=20):
    """CPU port of the full-catalogue scoring loop of eval/eval.py:205-222 on a bounded sample of the c2 workload: all item
    representations once, then `n_users` users in batches of 256 (scores, exclusion mask, top-k). -> scores/s on this box's cores."""
    from oracle import eval_ref, model_ref
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = {k_: v.detach().cpu().clone() for k_, v in net.state_dict().items()}
    ut = {'user_embedding': model_ref.RefTable('categorical', np.arange(ds.n_users), n_categories=ds.n_users)}
    it = {'text': model_ref.table_from_feature(ds.item_features['text']),
          'item_embedding': model_ref.RefTable('categorical', np.arange(ds.n_items), n_categories=ds.n_items)}
    ref = model_ref.RefSingleBranchNet(sd, model_config(net.config.shared_common_dim), ut, it)
    excl = ds.user_sampling_matrix_train.tocsr()
    t0 = time.perf_counter()
    with torch.no_grad():
        i_repr = ref.item_repr(torch.arange(ds.n_items), False)
        t_items = time.perf_counter() - t0
        t1 = time.perf_counter()
        for lo in range(0, n_users, 256):
            ub = torch.arange(lo, lo + 256)
            sc = eval_ref.masked_scores(ref.user_repr(ub, False), i_repr, excl[lo:lo + 256].toarray())
            eval_ref.topk(sc, k)
        t_users = time.perf_counter() - t1
    # a full pass = item representations once + every user batch at the sampled rate
    full = t_items + t_users * ds.n_users / n_users
    return {'value': round(ds.n_users * ds.n_items / full, 1), 'unit': 'scores/s', 'cores': cores, 'kind': 'port',
            'sample': f'item representations of all {ds.n_items} items ({t_items:.2f} s) + {n_users} of {ds.n_users} users in batches of '
                      f'256 ({t_users:.2f} s: scores, exclusion mask, top-{k}), extrapolated to the full pass'}


def replica_checksum(net, device):
    """Data-parallel sanity: after identical initialisation and exchanged gradients every replica must hold the same trainable
    parameters (BatchNorm running statistics are rank-local by design and are not parameters) -> (relative spread, checksums)"""
    import torch.distributed as dist
    with torch.no_grad():
        chk = torch.stack([torch.cat([p.detach().double().reshape(-1) for p in net.parameters()]).sum(),
                           torch.cat([p.detach().double().abs().reshape(-1) for p in net.parameters()]).sum()])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return float(((hi - lo).abs() / hi.abs().clamp_min(1e-30)).max()), [float(c) for c in chk]


C4 = dict(n_users=1_000_000, n_items=200_000, nnz=4_000_000, n_neg=10)
C4_MODEL = {'shared_common_dim': 256, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
            'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'image'}], 'single_branch_hidden_layers': [256],
                     'preference_hidden_layers': [], 'common_modality_dim': 256}}


def bench_c4_dp(S, device, rank, world, steps, small=False):
    """BASELINE configs[3] at its own size in the data-parallel job (world > 1): AmazonVideo2024-shaped synthetic data, 1M users x 200k
    items, text 768-d + image 2048-d, C = D = 256, user = embedding lookup (1 GB table per replica), sampled softmax, AdamW; every rank
    trains its own batch of 256 / 8192 interactions per step (weak scaling), the lookup table's gradient goes through the sparse
    (row, gradient) all-gather, everything else through the flat all-reduce."""
    c = dict(C4)
    if small:
        c.update(n_users=50_000, n_items=10_000, nnz=300_000)
    ds = S.SyntheticDataset(c['n_users'], c['n_items'], c['nnz'], item_dense={'text': 768, 'image': 2048}, seed=0,
                            n_negative_samples=c['n_neg'], negative_sampling_strategy='uniform_recbole', holdout_per_user=0)
    torch.manual_seed(42)
    np.random.seed(42)
    net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(C4_MODEL), ds).to(device)
    out = {'workload': f'BASELINE configs[3]: {c["n_users"]} users x {c["n_items"]} items, text 768 + image 2048, C = D = 256, user = lookup, '
                       f'sampled softmax, AdamW, dp{world}' + (' [SMALL DEBUG SIZE]' if small else ''),
           'params': sum(p.numel() for p in net.parameters())}
    n_steps = max(min(steps, 50), 10)
    for B in (256, 8192):
        dt, _ = bench_training(S, ds, net, device, B, n_steps, 5, rank, world, time_kernels=False)
        out[f'b{B}'] = {'value': round(B * world * n_steps / dt, 1), 'unit': 'interactions/s', 'ms_per_step': round(dt / n_steps * 1e3, 3),
                        'batch_per_gpu': B, 'steps': n_steps, 'user_table_gradient_exchange': EXCHANGE.get(B),
                        'loss_after_timed_steps': LAST_LOSS.get(B)}
    spread, chk = replica_checksum(net, device)
    out['replica_param_checksum_spread'] = spread
    out['param_checksum'] = chk
    return out


def bench_c5_dist(S, device, rank, world, small=False, k=20, chunk=100_000):
    """BASELINE configs[4] at its own size in the item-sharded job (world > 1): 1M users x 200k items x 256 fp16 N(0, 1)/16
    representations, 50 excluded items per user, top-20. Rank r holds items [lo_r, hi_r) and scores every 100k-user chunk against them
    with the fused kernel (item_offset = lo_r), the [chunk, k] lists are all-gathered and merged exactly (parallel.all_gather_topk ->
    sbr_merge_topk). One timed pass over all users with every chunk's exclusion mask resident (eval/eval.py:219: constant per split)."""
    import torch.distributed as dist
    import scipy.sparse as sp
    U, I, D = (1_000_000, 200_000, 256) if not small else (60_000, 20_000, 256)
    chunk = chunk if not small else 20_000
    lo, hi = S.parallel.item_shard(I, rank, world)
    g = torch.Generator(device='cpu').manual_seed(5)                      # the same tensors on every rank
    i_all = (torch.randn(I, D, generator=g) / 16).half()
    i16 = i_all[lo:hi].contiguous().to(device)
    del i_all
    chunks = []
    rng = np.random.default_rng(5)
    for s in range(0, U, chunk):
        n = min(chunk, U - s)
        u16 = (torch.randn(n, D, generator=g) / 16).half().to(device)
        cols = np.sort(rng.integers(0, I, size=(n, 50)), axis=1)
        m = sp.csr_matrix((np.ones(n * 50, dtype=np.int8), cols.reshape(-1), np.arange(0, n * 50 + 1, 50)), shape=(n, I))
        m.sum_duplicates()
        chunks.append((u16, torch.arange(n, device=device), S.evaluation._csr_to_device(m, device), S.ops.ScorerExclusions()))

    def one_pass():
        last = None
        for u16, users, excl, holder in chunks:
            val, idx = S.ops.score_topk_f16(u16, i16, k, users, excl[0], excl[1], item_offset=lo, exclusions=holder)
            last = S.parallel.all_gather_topk(val, idx, k)
        return last

    with torch.no_grad():
        one_pass()                                                         # builds the resident masks, warms the kernels
        one_pass()
        S.ops.KernelTimer.reset(True)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        val, idx = one_pass()
        torch.cuda.synchronize()
        dist.barrier()
        dt = time.perf_counter() - t0
        ts = [t for key, v in S.ops.KernelTimer.results().items() if key[0] == 'score_topk_f16' for t in v]
        S.ops.KernelTimer.reset(False)
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)
    ok = bool((idx[:, 0] >= 0).all()) and bool((idx.max() < I)) and bool((val[:, :-1] >= val[:, 1:]).all())
    avg_ms = sum(ts) / len(ts)
    return {'workload': f'BASELINE configs[4]: {U} users (chunks of {chunk}) x {I} items x {D} fp16, item-sharded over {world} ranks '
                        f'({hi - lo} items on this rank), 50 exclusions per user, top-{k}, all-gather + exact merge of the lists'
                        + (' [SMALL DEBUG SIZE]' if small else ''),
            'value': round(U * I / dt, 1), 'unit': 'scores/s', 'ms_per_pass': round(dt * 1e3, 3), 'chunks': len(chunks),
            'sharding': f'items/{world}', 'lists_sorted_and_in_range': ok,
            'roofline': scoring_roofline(chunks[0][0].shape[0], hi - lo, D, k, avg_ms,
                                         kernel='fused fp16 scorer on this rank\'s item shard, one 100k-user chunk (rank 0\'s launches)')}


def launch_ranks(n: int) -> int:
    """``python bench.py --gpus N`` without an external launcher: start N rank processes of this script (one per GPU, the
    environment torch.distributed.run would give them, rendezvous on 127.0.0.1) BEFORE this process has made any GPU / HIP
    call, wait for them and return the worst exit code. Rank 0 prints the JSON line on the inherited stdout. When a rank
    fails the others — which would wait for it in a collective forever — are terminated by PID."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), SBR_SELF_LAUNCHED='1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC only on these hosts (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:
                    q.terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch-size', type=int, default=8192, help='per-GPU batch (positive interactions per step)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-scoring', action='store_true')
    ap.add_argument('--no-b256', action='store_true')
    ap.add_argument('--no-c1', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help='skip the c3 and c5-shard objects')
    ap.add_argument('--small', action='store_true', help='1/10-size workload (debug only; never a reportable number)')
    ap.add_argument('--only', choices=['c3', 'c5'], default=None,
                    help='profiling passes (tools/profile_round.sh): run ONLY the c3 step (B = 4096) or the c5 shard scoring and print its object')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher around us: become the launcher (nothing in this process has touched the GPU yet)
        sys.exit(launch_ranks(args.gpus))

    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    # torch CPU ops of the loader threads: stay inside the job's CPU quota, shared by the ranks of the node
    torch.set_num_threads(max(1, host_cores() // int(os.environ.get('LOCAL_WORLD_SIZE', world))))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # SBR_DIST_BACKEND=gloo + fewer GPUs than ranks: rehearsal of the multi-process path on a one-GPU box (ranks share cuda:0,
    # collectives staged through the host) — never a reportable number
    backend = os.environ.get('SBR_DIST_BACKEND', 'nccl')
    rehearsal = backend != 'nccl'
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch exactly one rank per GPU '
                         f'(python bench.py --gpus N starts them itself)')
    n_dev = torch.cuda.device_count()                     # counting devices does not initialise the GPU
    if world > 1 and not rehearsal and n_dev < world:
        raise SystemExit(f'bench.py: {world} ranks over RCCL need {world} GPUs, this node shows {n_dev} '
                         f'(SBR_DIST_BACKEND=gloo rehearses the multi-process path on fewer GPUs; never a reportable number)')
    local = local % max(n_dev, 1)
    # SBR_FORCE_DIST=1: one-rank process group with the gradient / top-k exchange switched on (rehearsal of the RCCL call
    # sequence on a one-GPU box — never a reportable number)
    force_dist = world == 1 and os.environ.get('SBR_FORCE_DIST', '0') == '1'
    if force_dist:
        os.environ.setdefault('MASTER_PORT', '29517')
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(f'cuda:{local}'))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    device = f'cuda:{local}'
    torch.cuda.set_device(local)
    ranks_seen, backend_used = 1, None
    if dist.is_initialized():
        # the communicator that actually formed: every rank adds one
        seen = torch.ones(1, device=device, dtype=torch.int64)
        dist.all_reduce(seen)
        ranks_seen, backend_used = int(seen.item()), dist.get_backend()
        if dist.get_world_size() != args.gpus or (ranks_seen != args.gpus and not force_dist):
            raise SystemExit(f'bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks '
                             f'({ranks_seen} answered)')
        if backend_used != 'nccl' and not rehearsal:
            raise SystemExit(f'bench.py: backend {backend_used!r} is not RCCL')

    import sibrar_amd as S
    if args.only is not None:
        if world != 1:
            raise SystemExit('bench.py --only runs on one GPU')
        obj = bench_c5_shard(S, device) if args.only == 'c5' else bench_c3(S, device, args.steps)
        if 'roofline' in obj:
            obj['roofline'] = compact_roofline(obj['roofline'])
        print(json.dumps({'only': args.only, args.only: obj, 'tables': write_tables()}))
        return
    cfg = dict(C2)
    if args.small:
        cfg.update(n_users=10_000, n_items=5_000, nnz=500_000)
    ds, net = build(S, cfg, device)

    dt, timings = bench_training(S, ds, net, device, args.batch_size, args.steps, args.warmup, rank, world, time_kernels=True)
    value = args.batch_size * world * args.steps / dt
    out = {
        'metric': 'training interactions/s', 'value': round(value, 1), 'unit': 'interactions/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[1]: synthetic 100k users x 50k items, feat_dim=768, emb_dim=128, '
                               'sampled-softmax, 10 negatives, AdamW; user = embedding lookup, item = SingleBranchNet entity '
                               '(text 768-d + item-id embedding, hidden [128], BatchNorm)' + (' [SMALL DEBUG SIZE]' if args.small else ''),
                   'batch_per_gpu': args.batch_size, 'global_batch': args.batch_size * world, 'n_negatives': cfg['n_neg'],
                   'parallelism': f'dp{world}' if world > 1 else 'single', 'ranks_seen': ranks_seen,
                   'backend': ('rccl (torch.distributed nccl)' if backend_used == 'nccl' else
                               f'{backend_used} [REHEARSAL: not a reportable number]') if backend_used else None,
                   'settle_steps': SETTLE,
                   'loss_after_timed_steps': LAST_LOSS.get(args.batch_size)},
    }
    if world > 1:
        spread, chk = replica_checksum(net, device)
        out['config']['replica_param_checksum_spread'] = spread
        out['config']['param_checksum'] = chk
        out['config']['user_table_gradient_exchange'] = EXCHANGE.get(args.batch_size)
    n_params = sum(p.numel() for p in net.parameters())
    utab = net.user_embedding_module.embedding_layer.weight
    roof = step_roofline(timings, args.steps, args.batch_size, n_params, utab.numel(), utab.shape[1]) if rank == 0 else None
    if roof:
        out['roofline'] = compact_roofline(roof)
    b256 = sc = c5 = c3 = c1 = None
    if not args.no_b256:
        dt256, _ = bench_training(S, ds, net, device, 256, max(args.steps, 30), args.warmup, rank, world, time_kernels=False)
        b256 = {'value': round(256 * world * max(args.steps, 30) / dt256, 1), 'unit': 'interactions/s',
                'ms_per_step': round(dt256 / max(args.steps, 30) * 1e3, 3), 'batch_per_gpu': 256}
    if not args.no_scoring:
        sc = bench_scoring(S, ds, net, device, rank, world)
        sc['value'] = round(sc['value'], 1)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(S, ds, net, args.batch_size)
        TABLES['cpu_baseline_step_seconds'] = cb.pop('step_seconds')
        cb['speedup_vs_cpu'] = round(value / cb['value'], 1)
        out['cpu_baseline'] = cb
        if sc is not None:
            sc['cpu_baseline'] = cpu_scoring_sample(S, ds, net)
            sc['cpu_baseline']['speedup_vs_cpu'] = round(sc['value'] / sc['cpu_baseline']['value'], 1)
    if rank == 0 and world == 1 and not args.no_configs and not args.small:
        del ds, net
        c5 = bench_c5_shard(S, device)
        c3 = bench_c3(S, device, args.steps)
        if 'roofline' in c3:
            c3['roofline'] = compact_roofline(c3['roofline'])
    if rank == 0 and world == 1 and not args.no_c1 and not args.small:
        c1 = bench_c1(S, device, args.steps)
    c4dp = c5d = None
    if world > 1 and not args.no_configs:
        # the two BASELINE configs that name 8 GPUs, at their own sizes, next to the c2 curve (every rank takes part)
        del ds, net
        torch.cuda.empty_cache()
        c5d = bench_c5_dist(S, device, rank, world, small=args.small)
        c4dp = bench_c4_dp(S, device, rank, world, args.steps, small=args.small)
    if rank == 0:
        # the metric's second half (scores/s) also rides inside the two objects every record keeps, and the compact objects of the other
        # measurements come LAST so that a tail of the line always shows them; the per-kernel tables go to a side file
        if sc is not None and 'roofline' in out:
            out['roofline']['scoring_gemm'] = {k: sc['roofline'][k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms')}
            out['roofline']['scoring_gemm']['scores_per_s'] = sc['value']
        if sc is not None and 'cpu_baseline' in out and 'cpu_baseline' in sc:
            out['cpu_baseline']['scoring'] = {k: sc['cpu_baseline'][k] for k in ('value', 'unit', 'cores', 'kind', 'speedup_vs_cpu')}
        for key, obj in (('c1', c1), ('c3', c3), ('c4_dp', c4dp), ('c5', c5d), ('b256', b256), ('c5_shard', c5), ('scoring', sc)):
            if obj is not None:
                out[key] = obj
        out['tables'] = write_tables()
        line = json.dumps(out)
        if len(line) > 7600:                                  # the record keeps an 8 KB tail: shed prose before numbers
            for path in (('c1', 'workload'), ('c3', 'workload'), ('c1', 'trained_ndcg', 'what'), ('scoring', 'exclusion_mask'),
                         ('roofline', 'traffic_source'), ('roofline', 'timing'), ('c5_shard', 'workload'), ('scoring', 'cpu_baseline', 'sample'),
                         ('cpu_baseline', 'sample'), ('roofline', 'dominant_gemm')):
                o = out
                for k_ in path[:-1]:
                    o = o.get(k_, {}) if isinstance(o, dict) else {}
                if isinstance(o, dict):
                    o.pop(path[-1], None)
                line = json.dumps(out)
                if len(line) <= 7600:
                    break
        print(line)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
