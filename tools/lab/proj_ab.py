"""In-process A/B of the bf16-split projector kernel (sbr_gemm_split_proj_f32) at the c2 shape — 45,824 gathered rows of a [50k, 768]
feature matrix x W^T [128, 768], slot scatter — between the product library and every tools/lab/bin/libsibrar_*.so (e.g. from
tools/lab/build_proj_variants.sh), timed in rotation with HIP events; every variant's result is compared with the product's.
usage: python tools/lab/proj_ab.py [rounds]"""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import sibrar_amd as S
from importlib import import_module
L = import_module('sibrar---single-branch-recommender_amd._lib')
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 9
M, N, K, ROWS = int(os.environ.get('PJ_M', 45824)), 128, int(os.environ.get('PJ_K', 768)), 50_000
dev = 'cuda:0'
g = torch.Generator().manual_seed(3)
X = torch.randn(ROWS, K, generator=g).to(dev)
W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
b = torch.randn(N, generator=g).to(dev)
a_idx = torch.randint(0, ROWS, (M,), generator=g, dtype=torch.int32).to(dev)
c_idx = torch.randperm(2 * M, generator=g)[:M].to(torch.int32).to(dev)
hdr = L.parse_header()
paths = {'product': L.LIB_PATH}
for p in sorted(glob.glob(os.path.join(ROOT, 'tools', 'lab', 'bin', 'libsibrar_*.so'))):
    paths[os.path.basename(p)[len('libsibrar_'):-3]] = p
libs, outs = {}, {}
for name, p in paths.items():
    h = ctypes.CDLL(p)
    f = h.sbr_gemm_split_proj_f32
    f.restype, f.argtypes = hdr['sbr_gemm_split_proj_f32'][0], hdr['sbr_gemm_split_proj_f32'][1]
    libs[name] = f
    outs[name] = torch.zeros(2 * M, N, device=dev)
stream = torch.cuda.current_stream().cuda_stream


def run(name):
    rc = libs[name](X.data_ptr(), K, a_idx.data_ptr(), W.data_ptr(), K, b.data_ptr(), outs[name].data_ptr(), N, c_idx.data_ptr(), M, N, K, 1, stream)
    assert rc == 0, rc


for name in libs:
    run(name)
torch.cuda.synchronize()
for name in libs:
    d = (outs[name] - outs['product']).abs().max().item()
    print(f'{name}: max |difference to product| {d:.3g}')
times = {n: [] for n in libs}
for r in range(ROUNDS):
    for name in libs:
        for _ in range(3):
            run(name)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(name)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 10 * 1e3)
for name, ts in times.items():
    ts = sorted(ts)
    print(f'{name:12s} median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f}  max {ts[-1]:7.1f}')
