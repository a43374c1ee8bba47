"""In-process A/B of scorer builds: the product library and every tools/lab/bin/libsibrar_*.so (tools/lab/build_scorer_variants.sh) are
loaded into ONE process and timed in rotation on the same inputs — launch times from different processes / boxes differ by several
per cent, more than most variants do. Also checks every variant's result against the product library's.
usage: python tools/lab/scorer_ab.py [D] [excl 0|1] [rounds]"""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, scipy.sparse as sp
import sibrar_amd as S
from importlib import import_module
L = import_module('sibrar---single-branch-recommender_amd._lib')
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
EXCL = (sys.argv[2] if len(sys.argv) > 2 else '0') == '1'
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 7
U, I, K = int(os.environ.get('AB_USERS', 100_000)), (25_000 if D == 256 else 50_000), 20
dev = 'cuda:0'
g = torch.Generator().manual_seed(1)
u = (torch.randn(U, D, generator=g) / 8).half().to(dev)
it = (torch.randn(I, D, generator=g) / 8).half().to(dev)
users = torch.arange(U, device=dev)
ex = None
if EXCL:
    rng = np.random.default_rng(5)
    cols = rng.integers(0, I, size=(U, 50))
    m = sp.csr_matrix((np.ones(U * 50, dtype=np.int8), cols.reshape(-1), np.arange(0, U * 50 + 1, 50)), shape=(U, I))
    m.sum_duplicates()
    ex = S.evaluation._csr_to_device(m, dev)
hdr = L.parse_header()
paths = {'product': L.LIB_PATH}
for p in sorted(glob.glob(os.path.join(ROOT, 'tools', 'lab', 'bin', 'libsibrar_*.so'))):
    paths[os.path.basename(p)[len('libsibrar_'):-3]] = p
libs = {}
for name, p in paths.items():
    h = ctypes.CDLL(p)
    for fn in ('sbr_score_topk_f16', 'sbr_score_topk_f16_workspace', 'sbr_score_topk_f16_events_bytes', 'sbr_last_error'):
        f = getattr(h, fn)
        f.restype, f.argtypes = hdr[fn][0], hdr[fn][1]
    libs[name] = h
stream = torch.cuda.current_stream().cuda_stream
state = {}
for name, h in libs.items():
    ws = torch.empty(int(h.sbr_score_topk_f16_workspace(U, I, K)) + 64, dtype=torch.uint8, device=dev)
    nnz = int(ex[1].numel()) if ex else 0
    ev = torch.empty(int(h.sbr_score_topk_f16_events_bytes(U, nnz)) + 64, dtype=torch.uint8, device=dev) if ex else None
    state[name] = dict(ws=ws, ev=ev, val=torch.empty(U, K, device=dev), idx=torch.empty(U, K, dtype=torch.int32, device=dev), built=False)


def launch(name):
    h, st = libs[name], state[name]
    build = 0 if st['built'] else 1
    rc = h.sbr_score_topk_f16(u.data_ptr(), it.data_ptr(), D, U, I, users.data_ptr() if ex else None, ex[0].data_ptr() if ex else None,
                              ex[1].data_ptr() if ex else None, int(ex[1].numel()) if ex else 0, 0, K, st['val'].data_ptr(), st['idx'].data_ptr(),
                              st['ws'].data_ptr(), st['ws'].numel(), st['ev'].data_ptr() if ex else None, st['ev'].numel() if ex else 0, build, stream)
    if rc != 0:
        raise RuntimeError(f'{name}: {h.sbr_last_error().decode()}')
    st['built'] = True


for name in libs:
    for _ in range(4): launch(name)
torch.cuda.synchronize()
ref = state['product']
for name, st in state.items():
    same = bool((st['idx'] == ref['idx']).all()) and bool((st['val'] == ref['val']).all())
    print(f'{name}: result {"== product" if same else "DIFFERS from product (timing-only variant?)"}', flush=True)
times = {n: [] for n in libs}
for r in range(ROUNDS):
    for name in libs:
        evs = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); launch(name); b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        times[name].append(min(x.elapsed_time(y) for x, y in evs))
flop = 2.0 * U * I * D
for name, ts in times.items():
    ts = sorted(ts)
    med = ts[len(ts) // 2]
    print(f'D={D} excl={int(EXCL)} {name:12s} median {med:.3f} ms  min {ts[0]:.3f}  max {ts[-1]:.3f}   {flop / med / 1e9 / 2500 * 100:.1f} % of the fp16 peak', flush=True)
