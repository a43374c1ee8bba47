"""Lab: the fused scorer + loss + statistics kernel (sbr_bn_score_loss_fwd_bwd) against the three launches it replaces, at the c2
shape (B = 8192, N = 11, D = 128), HIP-event timed. SBR_LAB_LIB picks a variant library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sibrar_amd as S
from importlib import import_module
L = import_module('sibrar---single-branch-recommender_amd._lib')
if os.environ.get('SBR_LAB_LIB'):
    L.LIB_PATH = os.path.abspath(os.environ['SBR_LAB_LIB'])
dev = 'cuda'
B, N, D = 8192, 11, 128
g = torch.Generator(device=dev).manual_seed(0)
z, u = torch.randn(B * N, D, device=dev, generator=g), torch.randn(B, D, device=dev, generator=g)
mean, rstd = torch.zeros(D, device=dev), torch.ones(D, device=dev)
w, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)
labels = torch.zeros(B, N, device=dev, dtype=torch.float64); labels[:, 0] = 1
ws = torch.zeros(17 * 2 * D, device=dev, dtype=torch.float64)
lws = torch.zeros(int(L.lib().sbr_bn_score_loss_workspace()) // 8, device=dev, dtype=torch.float64)
lg, dl, du = torch.empty(B, N, device=dev), torch.empty(B, N, device=dev), torch.empty(B, D, device=dev)
l1, out3 = torch.zeros(1, device=dev, dtype=torch.float64), torch.zeros(3, device=dev, dtype=torch.float64)
st = L.stream()


def fused(kind):
    L.call('sbr_bn_score_loss_fwd_bwd', z.data_ptr(), u.data_ptr(), mean.data_ptr(), rstd.data_ptr(), w.data_ptr(), beta.data_ptr(), kind,
           labels.data_ptr(), 1.0 / B, 0.3, None, dl.data_ptr(), du.data_ptr(), l1.data_ptr(), out3.data_ptr(), B, N, D, ws.data_ptr(),
           lws.data_ptr(), lws.numel() * 8, st)


def three(kind):
    L.call('sbr_bn_score_fwd', z.data_ptr(), u.data_ptr(), mean.data_ptr(), rstd.data_ptr(), w.data_ptr(), beta.data_ptr(), lg.data_ptr(), B, N, D, st)
    L.call('sbr_rec_loss_fwd_bwd', kind, lg.data_ptr(), labels.data_ptr(), B, N, 1.0 / B, 0.3, l1.data_ptr(), dl.data_ptr(), st)
    L.call('sbr_bn_score_bwd_stats', dl.data_ptr(), u.data_ptr(), z.data_ptr(), du.data_ptr(), B, N, D, w.data_ptr(), beta.data_ptr(),
           mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), st)


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


for kind in (2, 1, 0):
    print(f'{os.environ.get("SBR_LAB_LIB", "product"):36s} kind {kind}: fused {timeit(lambda: fused(kind)) * 1e3:7.1f} us   three launches {timeit(lambda: three(kind)) * 1e3:7.1f} us', flush=True)
