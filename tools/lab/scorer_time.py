"""Fused scorer on the GPU box: correctness against an fp32 GEMM + top-k on a small shape, then launch times of the c2 shape
(100k x 50k x 128) and the c5 shard shape (100k x 25k x 256), with and without exclusions, exclusion mask resident / rebuilt.
SBR_ST_PRE (prefix tiles) is read per call by the library: pass a list to sweep.   usage: python tools/lab/scorer_time.py [pre ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, scipy.sparse as sp
import sibrar_amd as S
dev = 'cuda:0'
ops = S.ops


def excl_csr(U, I, per, seed):
    rng = np.random.default_rng(seed)
    cols = rng.integers(0, I, size=(U, per))
    m = sp.csr_matrix((np.ones(U * per, dtype=np.int8), cols.reshape(-1), np.arange(0, U * per + 1, per)), shape=(U, I))
    m.sum_duplicates()
    return S.evaluation._csr_to_device(m, dev)


def check(U, I, D, k, seed=0):
    g = torch.Generator().manual_seed(seed)
    u = (torch.randn(U, D, generator=g) / 4).half().to(dev)
    it = (torch.randn(I, D, generator=g) / 4).half().to(dev)
    ex = excl_csr(U, I, 7, seed)
    users = torch.arange(U, device=dev)
    val, idx = ops.score_topk_f16(u, it, k, users, ex[0], ex[1])
    sc = u.float() @ it.float().t()
    ops.mask_scores_(sc, users, ex[0], ex[1])
    rv, ri = ops.topk_rows(sc, k)
    ok_idx = bool((idx == ri).all())
    err = float((val - rv).abs().max())
    print(f'check {U}x{I}x{D} k={k}: idx equal {ok_idx}, max |dval| {err:.2e}', flush=True)
    return ok_idx


def t_ms(fn, warm=6, reps=12):
    for _ in range(warm): fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in evs)
    return ts[len(ts) // 2]


ok = all([check(300, 1000, 128, 20), check(1000, 5000, 256, 20, 1), check(2000, 20000, 128, 20, 2), check(257, 3299, 64, 10, 3)])
if not ok:
    print('MISMATCH'); sys.exit(1)
pres = sys.argv[1:] or ['default']
for (U, I, D) in ((100_000, 50_000, 128), (100_000, 25_000, 256)):
    g = torch.Generator().manual_seed(1)
    u = (torch.randn(U, D, generator=g) / 8).half().to(dev)
    it = (torch.randn(I, D, generator=g) / 8).half().to(dev)
    ex = excl_csr(U, I, 50, 5)
    users = torch.arange(U, device=dev)
    flop = 2.0 * U * I * D
    for pre in pres:
        if pre == 'default': os.environ.pop('SBR_ST_PRE', None)
        else: os.environ['SBR_ST_PRE'] = pre
        h = ops.ScorerExclusions()
        a = t_ms(lambda: ops.score_topk_f16(u, it, 20))
        b = t_ms(lambda: ops.score_topk_f16(u, it, 20, users, ex[0], ex[1], exclusions=h))
        c = t_ms(lambda: ops.score_topk_f16(u, it, 20, users, ex[0], ex[1]))
        print(f'{U}x{I}x{D} pre={pre}: no excl {a:.3f} ms ({flop / a / 1e9 / 2500 * 100:.1f} %)  excl resident {b:.3f} ms ({flop / b / 1e9 / 2500 * 100:.1f} %)  '
              f'excl rebuilt {c:.3f} ms', flush=True)
