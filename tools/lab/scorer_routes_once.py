"""A few launches of each fused scorer route on the c2 (or c5-shard) shape for rocprofv3 --kernel-trace --stats.
usage: python tools/lab/scorer_routes_once.py [D] [excl 0|1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, scipy.sparse as sp
import sibrar_amd as S
ops = S.ops
dev = 'cuda:0'
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
EXCL = (sys.argv[2] if len(sys.argv) > 2 else '1') == '1'
U, I = 100_000, (25_000 if D == 256 else 50_000)
g = torch.Generator().manual_seed(1)
u = (torch.randn(U, D, generator=g) / 8).half().to(dev)
it = (torch.randn(I, D, generator=g) / 8).half().to(dev)
rng = np.random.default_rng(5)
cols = rng.integers(0, I, size=(U, 50))
m = sp.csr_matrix((np.ones(U * 50, dtype=np.int8), cols.reshape(-1), np.arange(0, U * 50 + 1, 50)), shape=(U, I))
m.sum_duplicates()
ex = S.evaluation._csr_to_device(m, dev)
users = torch.arange(U, device=dev)
for route in (1, 2):
    ops.score_topk_route(route)
    h = ops.ScorerExclusions()
    for _ in range(6):
        if EXCL: ops.score_topk_f16(u, it, 20, users, ex[0], ex[1], exclusions=h)
        else: ops.score_topk_f16(u, it, 20)
torch.cuda.synchronize()
