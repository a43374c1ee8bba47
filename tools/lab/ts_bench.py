"""Lab: the dW (TN) products of the c2 step, slab pass only (no reducer), HIP-event timed. SBR_LAB_LIB picks a variant library,
SBR_TN_SPLIT=0 the fp32-pipe ring kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sibrar_amd as S
if os.environ.get('SBR_LAB_LIB'):
    from importlib import import_module
    import_module('sibrar---single-branch-recommender_amd._lib').LIB_PATH = os.path.abspath(os.environ['SBR_LAB_LIB'])
ops = S.ops
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn(50000, 768, device=dev, generator=g)
rows = torch.randint(0, 50000, (45824,), device=dev, generator=g, dtype=torch.int32)
H = torch.randn(90112, 128, device=dev, generator=g)
dZ = torch.randn(90112, 128, device=dev, generator=g)
dZt = dZ[:45824].contiguous()
o1, o2 = torch.empty(128, 128, device=dev), torch.empty(128, 768, device=dev)
d = ops.DeferredTN()


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in evs)
    return ts[len(ts) // 2]


def run(key, *a, **k):
    d.matmul_tn(key, *a, **k)
    d.pending = []


only = sys.argv[1] if len(sys.argv) > 1 else ''
for name, fl, fn in [c for c in [('128x128 over 90112', 2 * 90112 * 128 * 128, lambda: run('a', dZ, H, out=o1)),
                     ('128x768 over 45824 gathered', 2 * 45824 * 128 * 768, lambda: run('b', dZt, X, b_idx=rows, n_rows=45824, out=o2))] if only in c[0]]:
    ms = timeit(fn)
    print(f'{os.environ.get("SBR_LAB_LIB", "product"):40s} {name:30s} {ms * 1e3:8.1f} us {fl / ms / 1e9:8.1f} TFLOP/s', flush=True)
