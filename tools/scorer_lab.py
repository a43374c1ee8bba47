"""Narrow-wave fused scorer (score_topk_f16_n.hip) on the GPU box: launch time of the real kernel and of its timing-only ablations
(SBR_ST_DEBUG=1: MFMA loop only, 2: + threshold compares), cycle stamps per wave (SBR_ST_DEBUG=4), sweeps over the number of
consumer waves (SBR_ST_WAVES) and the prefix length (SBR_ST_PRE).   usage: python tools/scorer_lab.py [D] [I] [Bu] [what ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import sibrar_amd as S
from importlib import import_module
L = import_module('sibrar---single-branch-recommender_amd._lib')
if os.environ.get('SBR_LAB_LIB'):
    L.LIB_PATH = os.path.abspath(os.environ['SBR_LAB_LIB'])          # a variant built by tools/lab/build_scorer_variants.sh
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
I = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
Bu = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
what = sys.argv[4:] or ['time', 'ablate', 'stamps', 'waves', 'pre']
K = 20
g = torch.Generator(device='cuda').manual_seed(1)
u = (torch.randn(Bu, D, device='cuda', generator=g) / 8).half()
it = (torch.randn(I, D, device='cuda', generator=g) / 8).half()
need = int(L.lib().sbr_score_topk_f16_workspace(Bu, I, K))
ws = torch.zeros(need + (1 << 20), dtype=torch.uint8, device='cuda')
val = torch.empty(Bu, K, device='cuda'); idx = torch.empty(Bu, K, dtype=torch.int32, device='cuda')


def launch():
    L.call('sbr_score_topk_f16', u.data_ptr(), it.data_ptr(), D, Bu, I, None, None, None, 0, 0, K, val.data_ptr(), idx.data_ptr(),
           ws.data_ptr(), ws.numel(), None, 0, 1, L.stream())


def timed(env, reps=15, warm=10):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        for _ in range(warm): launch()
        evs = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); launch(); b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        ts = sorted(x.elapsed_time(y) for x, y in evs)
        return ts[len(ts) // 2]
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


MAXW = int(os.environ.get('SBR_LAB_MAXW', '14'))           # 16 - S5_NL of the library in use


def pick_waves(bu, n_cu=256, maxw=MAXW):
    if os.environ.get('SBR_ST_WAVES'):
        return max(1, min(maxw, int(os.environ['SBR_ST_WAVES'])))
    units = -(-bu // 32)
    rounds = -(-units // (n_cu * maxw))
    return max(1, min(maxw, -(-units // (rounds * n_cu))))


flop = 2.0 * Bu * I * D
if 'time' in what:
    ms = timed({})
    print(f'D={D} I={I} Bu={Bu} W={pick_waves(Bu)}: {ms:.3f} ms = {flop / ms / 1e9:.0f} TFLOP/s = {flop / ms / 1e9 / 2500 * 100:.1f} % of 2.5 PF')
if 'ablate' in what:
    for dbg, name in ((1, 'MFMA loop only'), (5, 'MFMA loop without loads / hand-off'), (6, 'MFMA loop without fragment reads'), (7, 'MFMA loop without MFMAs'), (2, 'MFMA loop + threshold compares')):
        ms = timed({'SBR_ST_DEBUG': dbg})
        print(f'  ablation {dbg} ({name}): {ms:.3f} ms')
if 'stamps' in what:
    os.environ['SBR_ST_DEBUG'] = '4'
    ws.zero_()
    launch(); launch()
    torch.cuda.synchronize()
    W = pick_waves(Bu)
    n_wg = -(-Bu // (32 * W))
    off = n_wg * 32 * W * 2 * 64 * 8
    raw = ws[off:off + n_wg * MAXW * 64].view(torch.int64).cpu().numpy().reshape(n_wg * MAXW, 8)
    raw = raw[raw[:, 0] > 0]
    d = raw.astype(np.float64)
    tot = d[:, 0].mean()
    n_evt, t_issue = raw[:, 4] & 0xFFFFF, raw[:, 4] >> 20
    n_ins, t_ladder = raw[:, 5] & 0xFFFFF, raw[:, 5] >> 20
    clk = d[:, 0] / np.maximum(d[:, 7], 1) * 100e6 / 1e9
    print(f'  stamps per wave (mean over {len(raw)} waves, W={W}, {n_wg} workgroups): total {tot:.3g} cyc at {np.median(clk):.2f} GHz (wall {d[:,7].mean()/100:.0f} us, '
          f'max {d[:,7].max()/100:.0f} us) | tile wait {d[:,1].mean():.3g} ({100*d[:,1].mean()/tot:.1f}%) | issue (reads, MFMA issue, release, exclusion walk) '
          f'{t_issue.mean():.3g} ({100*t_issue.mean()/tot:.1f}%) | ladder {t_ladder.mean():.3g} ({100*t_ladder.mean()/tot:.1f}%) of it candidate blocks '
          f'{d[:,2].mean():.3g} (blocks={n_evt.mean():.0f}, candidates={d[:,3].mean():.0f} = {d[:,3].mean()/32:.0f} per user) | compactions {d[:,6].mean():.3g} '
          f'({100*d[:,6].mean()/tot:.1f}%), n={n_ins.mean():.0f} = {n_ins.mean()/32:.2f} per user')
    os.environ.pop('SBR_ST_DEBUG')
if 'clock' in what:
    W = pick_waves(Bu)
    n_wg = -(-Bu // (32 * W))
    off = n_wg * 32 * W * 2 * 64 * 8
    for dbg, name in ((1, 'MFMA loop only'), (6, 'without fragment reads'), (5, 'without loads / hand-off'), (2, '+ compares'), (4, 'real kernel + stamps')):
        os.environ['SBR_ST_DEBUG'] = str(dbg)
        ws.zero_()
        for _ in range(12): launch()
        torch.cuda.synchronize()
        raw = ws[off:off + n_wg * MAXW * 64].view(torch.int64).cpu().numpy().reshape(n_wg * MAXW, 8)
        raw = raw[raw[:, 0] > 0].astype(np.float64)
        clk = raw[:, 0] / np.maximum(raw[:, 7], 1) * 100e6 / 1e9
        print(f'  in-kernel clock, ablation {dbg} ({name}): median {np.median(clk):.2f} GHz (p10 {np.percentile(clk,10):.2f}, p90 {np.percentile(clk,90):.2f}), '
              f'wave wall mean {raw[:,7].mean()/100:.0f} us, cycles {raw[:,0].mean():.3g}')
    os.environ.pop('SBR_ST_DEBUG')
if 'phases' in what:
    W = pick_waves(Bu)
    n_wg = -(-Bu // (32 * W))
    off = n_wg * 32 * W * 2 * 64 * 8
    os.environ['SBR_ST_DEBUG'] = '3'
    ws.zero_()
    for _ in range(12): launch()
    torch.cuda.synchronize()
    raw = ws[off:off + n_wg * MAXW * 64].view(torch.int64).cpu().numpy().reshape(n_wg * MAXW, 8)
    raw = raw[raw[:, 0] > 0]
    t_issue, t_ladder = (raw[:, 4] >> 20).astype(np.float64), (raw[:, 5] >> 20).astype(np.float64)
    tot = raw[:, 0].astype(np.float64)
    nt = -(-I // (64 if D != 256 else 32))
    print(f'  light stamps (3 per tile), per wave: total {tot.mean():.3g} cyc, wall {raw[:,7].mean()/100:.0f} us | main pass: poll + MFMA phase {t_issue.mean():.3g} '
          f'({t_issue.mean()/nt:.0f} per tile) | release + exclusion walk + ladder + blocks {t_ladder.mean():.3g} ({t_ladder.mean()/nt:.0f} per tile) | rest '
          f'(prefix pass, compactions, final selection) {(tot - t_issue - t_ladder).mean():.3g}')
    os.environ.pop('SBR_ST_DEBUG')
if 'waves' in what:
    for w in (14, 13, 12, 10, 8, 7, 13, 14):
        print(f'  SBR_ST_WAVES={w}: {timed({"SBR_ST_WAVES": w}):.3f} ms ({-(-Bu // (32 * w))} workgroups)')
if 'pre' in what:
    nt = -(-I // (64 if D != 256 else 32))
    for frac in (0, 24, 16, 12, 8, 6):
        pre = 0 if frac == 0 else nt // frac
        print(f'  SBR_ST_PRE={pre} tiles (1/{frac}): {timed({"SBR_ST_PRE": pre}):.3f} ms')
if 'v3' in what:
    print(f'  transposed 64-users-per-wave kernel (SBR_SCORER_V3=1): {timed({"SBR_SCORER_V3": 1}):.3f} ms')
