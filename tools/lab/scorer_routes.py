"""In-process A/B of the two fused scorers (one-pass kernel vs two-pass scorer) on the c2 and c5-shard shapes, exclusions resident.
usage: python tools/lab/scorer_routes.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, scipy.sparse as sp
import sibrar_amd as S
ops = S.ops
dev = 'cuda:0'
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 7


def excl_csr(U, I, per, seed):
    rng = np.random.default_rng(seed)
    cols = rng.integers(0, I, size=(U, per))
    m = sp.csr_matrix((np.ones(U * per, dtype=np.int8), cols.reshape(-1), np.arange(0, U * per + 1, per)), shape=(U, I))
    m.sum_duplicates()
    return S.evaluation._csr_to_device(m, dev)


def once(fn):
    evs = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    return min(x.elapsed_time(y) for x, y in evs)


for (U, I, D) in ((100_000, 50_000, 128), (100_000, 25_000, 256), (100_000, 50_000, 64)):
    g = torch.Generator().manual_seed(1)
    u = (torch.randn(U, D, generator=g) / 8).half().to(dev)
    it = (torch.randn(I, D, generator=g) / 8).half().to(dev)
    ex = excl_csr(U, I, 50, 5)
    users = torch.arange(U, device=dev)
    flop = 2.0 * U * I * D
    res = {}
    for excl in (False, True):
        holders = {1: ops.ScorerExclusions(), 2: ops.ScorerExclusions()}
        times = {1: [], 2: []}

        def run(route):
            ops.score_topk_route(route)
            if excl: return once(lambda: ops.score_topk_f16(u, it, 20, users, ex[0], ex[1], exclusions=holders[route]))
            return once(lambda: ops.score_topk_f16(u, it, 20))
        for route in (1, 2):
            for _ in range(2): run(route)
        for r in range(ROUNDS):
            for route in (1, 2):
                times[route].append(run(route))
        for route in (1, 2):
            ts = sorted(times[route]); med = ts[len(ts) // 2]
            print(f'{U}x{I}x{D} excl={int(excl)} route={"one-pass" if route == 1 else "two-pass"}: median {med:.3f} ms min {ts[0]:.3f} max {ts[-1]:.3f}  ({flop / med / 1e9 / 2500 * 100:.1f} % of the fp16 peak)', flush=True)
ops.score_topk_route(0)
