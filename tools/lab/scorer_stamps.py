"""Per-wave cycle stamps of the fused scorer (SBR_ST_DEBUG=4 build of the kernel: about +10 % run time): where a consumer wave's
cycles go — waiting for tiles, issuing a tile's MFMAs, ladder + appends, overflow selections — and how many register pairs fired /
thresholds were refreshed. c2 shape.   usage: python tools/lab/scorer_stamps.py [D]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sibrar_amd as S
from importlib import import_module
L = import_module('sibrar---single-branch-recommender_amd._lib')
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
U, I, K = 100_000, (25_000 if D == 256 else 50_000), 20
dev = 'cuda:0'
g = torch.Generator().manual_seed(1)
u = (torch.randn(U, D, generator=g) / 8).half().to(dev)
it = (torch.randn(I, D, generator=g) / 8).half().to(dev)
lib = L.lib()
need = int(lib.sbr_score_topk_f16_workspace(U, I, K))
ws = torch.zeros(need + 64, dtype=torch.uint8, device=dev)
val = torch.empty(U, K, device=dev); idx = torch.empty(U, K, dtype=torch.int32, device=dev)
os.environ['SBR_ST_DEBUG'] = '4'
for _ in range(3):
    L.call('sbr_score_topk_f16', u.data_ptr(), it.data_ptr(), D, U, I, None, None, None, 0, 0, K, val.data_ptr(), idx.data_ptr(), ws.data_ptr(),
           ws.numel(), None, 0, 1, L.stream())
torch.cuda.synchronize()
os.environ['SBR_ST_DEBUG'] = '0'
MAXW, CAPH = 14, 256                 # S5_MAXW, S5_CAPH of csrc/score_topk_f16_n.hip; the workspace layout of s5_workspace_bytes
N_CU = torch.cuda.get_device_properties(0).multi_processor_count
padded = -(-U // 32) * 32 + 32 * MAXW + 32 * N_CU
off = padded * 2 * CAPH * 8 + ((padded * 2 * 8 + 15) & ~15)
n_wg_max = (U + 31) // 32 + MAXW + N_CU
dbg = ws[off:off + n_wg_max * MAXW * 64].view(torch.int64).cpu().numpy().reshape(-1, 8)
full = dbg.reshape(-1, MAXW, 8)
for w in range(MAXW):
    sel = full[:, w, :]
    sel = sel[sel[:, 0] > 0]
    if len(sel):
        print(f'  wave {w:2d}: n {len(sel):4d}  total/tile {np.median(sel[:, 0]) / (-(-I // (32 if D == 256 else 64))):.0f}  wait/tile {np.median(sel[:, 1]) / (-(-I // (32 if D == 256 else 64))):.0f}  issue/tile {np.median(sel[:, 4] >> 20) / (-(-I // (32 if D == 256 else 64))):.0f}  ladder/tile {np.median(sel[:, 5] >> 20) / (-(-I // (32 if D == 256 else 64))):.0f}')
dbg = dbg[dbg[:, 0] > 0]
tot, wait = dbg[:, 0].astype(np.float64), dbg[:, 1].astype(np.float64)
issue, ladder = (dbg[:, 4] >> 20).astype(np.float64), (dbg[:, 5] >> 20).astype(np.float64)
pairs, over = (dbg[:, 4] & 0xFFFFF).astype(np.float64), (dbg[:, 5] & 0xFFFFF).astype(np.float64)
refresh, tcmp, real = dbg[:, 3].astype(np.float64), dbg[:, 6].astype(np.float64), dbg[:, 7].astype(np.float64)
n_tiles = -(-I // (32 if D == 256 else 64))
print(f'D={D}: {len(dbg)} consumer waves, {n_tiles} tiles; per wave (median): total {np.median(tot):.0f} cycles = {np.median(real) / 100:.0f} us '
      f'(clock {np.median(tot / real) * 100:.0f} MHz)')
print(f'  per tile: total {np.median(tot) / n_tiles:.0f}  wait-for-tile {np.median(wait) / n_tiles:.0f}  MFMA issue phase {np.median(issue) / n_tiles:.0f}  '
      f'ladder+appends {np.median(ladder) / n_tiles:.0f}  (overflow selections {np.median(tcmp) / n_tiles:.0f})')
print(f'  pairs fired per tile {np.median(pairs) / n_tiles:.2f}  refreshes {np.median(refresh):.0f}  overflow selections per wave {np.median(over):.1f}')
