import os, sys
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/tools') else os.getcwd())
import torch
import sibrar_amd as S
from importlib import import_module
L = import_module('sibrar---single-branch-recommender_amd._lib')
dev='cuda'
g = torch.Generator(device=dev).manual_seed(0)
n, D = 45824, 128
d = torch.randn(90113, D, device=dev, generator=g)
slots = torch.randperm(90112, device=dev, generator=g)[:n].to(torch.int32)
rows = torch.randint(0, 50000, (n,), device=dev, generator=g, dtype=torch.int32)
def run(out):
    L.call('sbr_scatter_add_rows', d.data_ptr(), D, slots.data_ptr(), rows.data_ptr(), out.data_ptr(), D, n, D, L.stream())
res = {}
for flag in ('own', '1', '0'):
    os.environ['SBR_SCATTER_OWNED'] = '1' if flag == 'own' else '0'
    os.environ['SBR_SCATTER_V1'] = '0' if flag == 'own' else flag
    out = torch.zeros(50000, D, device=dev)
    run(out); res[flag] = out.clone()
    for _ in range(5): run(out)
    evs=[]
    for _ in range(20):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); run(out); b.record(); evs.append((a,b))
    torch.cuda.synchronize()
    ts=sorted(x.elapsed_time(y) for x,y in evs)
    print('variant %s (own = owner kernel, 1 = one element per thread, 0 = persistent waves): %.1f us' % (flag, ts[len(ts)//2]*1e3))
ref = torch.zeros(50000, D, device=dev, dtype=torch.float64)
ref.index_add_(0, rows.long(), d[slots.long()].double())
print('max err owner %.3g persistent %.3g old %.3g' % tuple(float((res[k].double()-ref).abs().max()) for k in ('own', '0', '1')))

# cold destination: a 512 MB write between two runs pushes the table gradient out of L2 / the Infinity Cache (as the rest of a
# training step does between the gradient reset and the scatter)
big = torch.empty(128 * 1024 * 1024, device=dev)
for flag in ('own', '1', '0'):
    os.environ['SBR_SCATTER_OWNED'] = '1' if flag == 'own' else '0'
    os.environ['SBR_SCATTER_V1'] = '0' if flag == 'own' else flag
    out = torch.zeros(50000, D, device=dev)
    ts = []
    for _ in range(12):
        big.fill_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(out); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    print('cold destination, SBR_SCATTER_V1=%s: %.1f us' % (flag, ts[len(ts) // 2] * 1e3))
    # warm again: zero the table right before the scatter (the zeros stay in the Infinity Cache)
    ts = []
    for _ in range(12):
        big.fill_(1.0)
        out.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(out); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    print('zeroed just before, SBR_SCATTER_V1=%s: %.1f us' % (flag, ts[len(ts) // 2] * 1e3))
