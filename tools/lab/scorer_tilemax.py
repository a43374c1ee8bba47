"""Timing-only ablations for the two-pass scorer design (needs a lab library: SBR_LAB_LIB=tools/lab/bin/libsibrar_lab.so built by
`bash tools/lab/build_scorer_variants.sh "lab"`): the complete kernel (0), its MFMA loop alone (1), + threshold compares (2), and the
class-maxima-only main pass (8: one v_max3 per accumulator register pair per tile, 16 floats per lane stored every 32 / 64 tiles; 9: with
the exclusion bits applied first), all in ONE process on the same inputs.   usage: python tools/lab/scorer_tilemax.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, scipy.sparse as sp
import sibrar_amd as S
ops = S.ops
dev = 'cuda:0'
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
assert os.environ.get('SBR_LAB_LIB'), 'set SBR_LAB_LIB to a library built with -DSBR_LAB'


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


for (U, I, D) in ((100_000, 50_000, 128), (100_000, 25_000, 256)):
    g = torch.Generator().manual_seed(1)
    u = (torch.randn(U, D, generator=g) / 8).half().to(dev)
    it = (torch.randn(I, D, generator=g) / 8).half().to(dev)
    ex = excl_csr(U, I, 50, 5)
    users = torch.arange(U, device=dev)
    h = ops.ScorerExclusions()
    flop = 2.0 * U * I * D
    variants = [('0', None, False), ('0', None, True), ('1', '0', False), ('2', None, False), ('8', '0', False), ('9', '0', True), ('8', '0', True)]
    times = {v: [] for v in variants}

    def run(v):
        dbg, pre, excl = v
        os.environ['SBR_ST_DEBUG'] = dbg
        if pre is None: os.environ.pop('SBR_ST_PRE', None)
        else: os.environ['SBR_ST_PRE'] = pre
        if excl: return once(lambda: ops.score_topk_f16(u, it, 20, users, ex[0], ex[1], exclusions=h))
        return once(lambda: ops.score_topk_f16(u, it, 20))

    for v in variants:
        for _ in range(2): run(v)
    for r in range(ROUNDS):
        for v in variants:
            times[v].append(run(v))
    for v, ts in times.items():
        ts = sorted(ts)
        med = ts[len(ts) // 2]
        print(f'{U}x{I}x{D} DBG={v[0]} pre={v[1]} excl={int(v[2])}: median {med:.3f} ms min {ts[0]:.3f} max {ts[-1]:.3f}  ({flop / med / 1e9 / 2500 * 100:.1f} % of the fp16 peak)', flush=True)
os.environ['SBR_ST_DEBUG'] = '0'
