"""Cycle stamps of the 64-users-per-wave scorer (SBR_ST_DEBUG=4): per consumer wave total cycles, time waiting for tiles, time
inside candidate blocks, numbers of candidates / blocks / inserts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['SBR_ST_DEBUG'] = '4'
import torch, numpy as np
import sibrar_amd as S
from importlib import import_module
L = import_module('sibrar---single-branch-recommender_amd._lib')
g = torch.Generator(device='cuda').manual_seed(1)
Bu, I, D = 100000, 50000, int(sys.argv[1]) if len(sys.argv) > 1 else 128
u = (torch.randn(Bu, D, device='cuda', generator=g) / 8).half()
it = (torch.randn(I, D, device='cuda', generator=g) / 8).half()
nwg = (Bu + 447) // 448
dbg = torch.zeros(nwg * 7 * 8, dtype=torch.int64, device='cuda')
val = torch.empty(Bu, 20, device='cuda'); idx = torch.empty(Bu, 20, dtype=torch.int32, device='cuda')
for _ in range(2):
    L.call('sbr_score_topk_f16', u.data_ptr(), it.data_ptr(), D, Bu, I, None, None, None, 0, 20, val.data_ptr(), idx.data_ptr(), dbg.data_ptr(), dbg.numel() * 8, L.stream())
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(nwg * 7, 8).astype(np.float64)
d = d[d[:, 0] > 0]
tot, wait, evt, ncand, nevt, nins = [d[:, i] for i in range(6)]
print(f'per wave (mean over {len(d)} waves): total {tot.mean():.3g} cyc | tile wait {wait.mean():.3g} ({100*wait.mean()/tot.mean():.1f}%) | '
      f'candidate blocks {evt.mean():.3g} ({100*evt.mean()/tot.mean():.1f}%), blocks={nevt.mean():.0f}, candidates={ncand.mean():.0f}, '
      f'inserts={nins.mean():.0f}, {evt.mean()/max(ncand.mean(),1):.0f} cyc/candidate')
