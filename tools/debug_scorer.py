import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import sibrar_amd as S
ops = S.ops
torch.manual_seed(0)
for (Bu, I, D, k, excl) in [(32, 64, 64, 3, False), (32, 128, 64, 3, False), (32, 640, 64, 3, False), (64, 640, 64, 3, False),
                            (300, 640, 128, 20, False), (32, 640, 64, 3, True)]:
    u = (torch.randn(Bu, D) / 4).half()
    it = (torch.randn(I, D) / 4).half()
    ref = u.double() @ it.double().t()
    kw = {}
    if excl:
        import scipy.sparse as sp
        m = sp.random(Bu, I, density=0.05, format='csr', random_state=2); m.sort_indices()
        ref[torch.from_numpy(m.toarray() != 0)] = -float('inf')
        kw = dict(u_idx=torch.arange(Bu, device='cuda'), excl_indptr=torch.from_numpy(m.indptr.astype(np.int64)).cuda(),
                  excl_indices=torch.from_numpy(m.indices.astype(np.int32)).cuda())
    val, idx = ops.score_topk_f16(u.cuda(), it.cuda(), k, **kw)
    torch.cuda.synchronize()
    tv, ti = torch.topk(ref, k, sorted=True)
    bad = (val.cpu().double() - tv).abs() > 1e-4
    print((Bu, I, D, k, excl), 'bad rows:', int(bad.any(1).sum()), 'of', Bu, flush=True)
    if bad.any():
        r = int(torch.nonzero(bad.any(1))[0])
        print('  row', r, 'got', val[r].cpu().tolist()[:5], idx[r].cpu().tolist()[:5], 'exp', tv[r].tolist()[:5], ti[r].tolist()[:5], flush=True)
