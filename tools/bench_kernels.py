"""Micro-benchmarks of individual C-ABI kernels on the GPU box (HIP-event timed, median of reps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import sibrar_amd as S
if os.environ.get('SBR_LAB_LIB'):
    from importlib import import_module
    import_module('sibrar---single-branch-recommender_amd._lib').LIB_PATH = os.path.abspath(os.environ['SBR_LAB_LIB'])
ops = S.ops
dev = 'cuda'


def timeit(fn, reps=20, warm=3):
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


def gemm_suite():
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn(50000, 768, device=dev, generator=g)
    rows = torch.randint(0, 50000, (45056,), device=dev, generator=g, dtype=torch.int32)
    W = torch.randn(128, 768, device=dev, generator=g)
    H = torch.randn(90112, 128, device=dev, generator=g)
    W2 = torch.randn(128, 128, device=dev, generator=g)
    dZ = torch.randn(90112, 128, device=dev, generator=g)
    dZt = dZ[:45056].contiguous()
    cases = [
        ('NT proj  45056x128x768 gather', 2 * 45056 * 128 * 768, lambda: ops.linear_nt(X, W, None, 1, a_idx=rows)),
        ('NT mlp   90112x128x128', 2 * 90112 * 128 * 128, lambda: ops.linear_nt(H, W2, None, 1)),
        ('NN dx    90112x128x128', 2 * 90112 * 128 * 128, lambda: ops.matmul_nn(dZ, W2)),
        ('TN dW    128x128 over 90112', 2 * 90112 * 128 * 128, lambda: ops.matmul_tn(dZ, H)),
        ('TN dWproj 128x768 over 45056 gather', 2 * 45056 * 128 * 768, lambda: ops.matmul_tn(dZt, X, b_idx=rows, n_rows=45056)),
    ]
    outb = torch.empty(90112, 128, device=dev)
    for name, flops, fn in cases:
        ms = timeit(fn)
        print(f'{name:42s} {ms*1e3:9.1f} us  {flops/ms/1e9:8.2f} TFLOP/s')
    for split in (True, False):
        ops._SPLIT = split
        tag = 'bf16x3 split' if split else 'fp32 pipe   '
        for name, fn in [('NT mlp', lambda: ops.linear_nt(H, W2, None, 1, out=outb)), ('NN dx', lambda: ops.matmul_nn(dZ, W2, out=outb)),
                         ('NN dx*act\'(Y)', lambda: ops.matmul_nn_actgrad(dZ, W2, H, 1, outb, None))]:
            ms = timeit(fn)
            print(f'{tag} {name:18s} 90112x128x128 {ms*1e3:9.1f} us  {2*90112*128*128/ms/1e9:8.2f} TFLOP/s  {2*90112*512/ms/1e6:7.1f} GB/s')


def score_suite():
    g = torch.Generator(device=dev).manual_seed(1)
    for (Bu, I, D) in [(100000, 50000, 128), (100000, 25000, 256)]:
        u = (torch.randn(Bu, D, device=dev, generator=g) / 8).half()
        it = (torch.randn(I, D, device=dev, generator=g) / 8).half()
        import scipy.sparse as sp
        rng = np.random.default_rng(0)
        cnt = np.full(Bu, 50)
        indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)).to(dev)
        cols = np.sort(rng.integers(0, I, size=(Bu, 50)), axis=1).astype(np.int32).reshape(-1)
        indices = torch.from_numpy(cols).to(dev)
        uidx = torch.arange(Bu, device=dev)
        for excl in (False, True):
            fn = (lambda: ops.score_topk_f16(u, it, 20, uidx, indptr, indices)) if excl else (lambda: ops.score_topk_f16(u, it, 20))
            ms = timeit(fn, reps=15, warm=5)
            print(f'score_topk_f16 {Bu}x{I}x{D} excl={excl}: {ms:8.3f} ms  {2.0*Bu*I*D/ms/1e9:8.1f} TFLOP/s')


if __name__ == '__main__':
    which = sys.argv[1:] or ['gemm', 'score']
    if 'gemm' in which:
        gemm_suite()
    if 'score' in which:
        score_suite()
