"""Cycle stamps of the EARLIER fused scorer kernels (SBR_SCORER_V3=1 is set here; the narrow-wave kernel that replaced them is
measured by tools/scorer_lab.py) (SBR_ST_DEBUG=4): per consumer wave total cycles, time waiting for tiles, time in candidate
blocks, compaction time and counts. D = 64 | 128: the 64-users-per-wave kernel (448 users per workgroup); D = 256 or
SBR_SCORER_V1=1: the 32-users-per-wave kernel (224 users per workgroup; fields: total, wait, events, overflow)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('SBR_ST_DEBUG', '4')
os.environ.setdefault('SBR_SCORER_V3', '1')
import torch, numpy as np
import sibrar_amd as S
from importlib import import_module
L = import_module('sibrar---single-branch-recommender_amd._lib')
g = torch.Generator(device='cuda').manual_seed(1)
Bu, I, D = 100000, 50000, int(sys.argv[1]) if len(sys.argv) > 1 else 128
wide = D in (64, 128, 256) and os.environ.get('SBR_SCORER_V1', '0') == '0'
u = (torch.randn(Bu, D, device='cuda', generator=g) / 8).half()
it = (torch.randn(I, D, device='cuda', generator=g) / 8).half()
rows = 448 if wide else 224
nwg = (Bu + rows - 1) // rows
need = int(L.lib().sbr_score_topk_f16_workspace(Bu, I, 20))
ws = torch.zeros(max(need, nwg * 7 * 64), dtype=torch.uint8, device='cuda')
val = torch.empty(Bu, 20, device='cuda'); idx = torch.empty(Bu, 20, dtype=torch.int32, device='cuda')
for _ in range(2):
    L.call('sbr_score_topk_f16', u.data_ptr(), it.data_ptr(), D, Bu, I, None, None, None, 0, 0, 20, val.data_ptr(), idx.data_ptr(), ws.data_ptr(), ws.numel(), None, 0, 1, L.stream())
torch.cuda.synchronize()
stamps = ws[need - nwg * 7 * 64:need] if wide else ws[:nwg * 7 * 64]
d = stamps.view(torch.int64).cpu().numpy().reshape(nwg * 7, 8).astype(np.float64)
d = d[d[:, 0] > 0]
tot = d[:, 0].mean()
if wide and os.environ.get('SBR_SCORER_V2', '0') == '0':
    raw = stamps.view(torch.int64).cpu().numpy().reshape(nwg * 7, 8)
    raw = raw[raw[:, 0] > 0]
    n_evt, t_issue = raw[:, 4] & 0xFFFFF, raw[:, 4] >> 20
    n_ins, t_ladder = raw[:, 5] & 0xFFFFF, raw[:, 5] >> 20
    clk = raw[:, 0].astype(np.float64) / np.maximum(raw[:, 7], 1) * 100e6 / 1e9
    print(f'[transposed kernel, SBR_ST_DEBUG={os.environ["SBR_ST_DEBUG"]}] per wave (mean over {len(raw)} waves): total {tot:.3g} cyc at {np.median(clk):.2f} GHz '
          f'(wall {raw[:,7].mean()/100:.0f} us) | tile wait {d[:,1].mean():.3g} ({100*d[:,1].mean()/tot:.1f}%) | issue (reads, MFMA issue, release, exclusion walk) '
          f'{t_issue.mean():.3g} ({100*t_issue.mean()/tot:.1f}%) | ladder (MFMA completion, compares, blocks) {t_ladder.mean():.3g} ({100*t_ladder.mean()/tot:.1f}%) '
          f'| of it candidate blocks {d[:,2].mean():.3g}, blocks={n_evt.mean():.0f}, candidates={d[:,3].mean():.0f} | compactions {d[:,6].mean():.3g} '
          f'({100*d[:,6].mean()/tot:.1f}%), n={n_ins.mean():.0f}')
elif wide:
    print(f'per wave (mean over {len(d)} waves): total {tot:.3g} cyc | tile wait {d[:,1].mean():.3g} ({100*d[:,1].mean()/tot:.1f}%) | '
          f'candidate blocks {d[:,2].mean():.3g} ({100*d[:,2].mean()/tot:.1f}%), blocks={d[:,4].mean():.0f}, candidates={d[:,3].mean():.0f}, '
          f'{d[:,2].mean()/max(d[:,4].mean(),1):.0f} cyc/block | compactions {d[:,6].mean():.3g} ({100*d[:,6].mean()/tot:.1f}%), n={d[:,5].mean():.0f}')
else:
    print(f'per wave (mean over {len(d)} waves): total {tot:.3g} cyc | tile wait {d[:,1].mean():.3g} ({100*d[:,1].mean()/tot:.1f}%) | '
          f'event blocks {d[:,2].mean():.3g} ({100*d[:,2].mean()/tot:.1f}%), n={d[:,4].mean():.0f} | overflow {d[:,3].mean():.3g} '
          f'({100*d[:,3].mean()/tot:.1f}%), n={d[:,5].mean():.0f}')
