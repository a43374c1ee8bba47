"""Condenses rocprofv3 output of `bench.py` runs (gpurun_out/<run>/{stats,fetch,write}) into the committed summaries under
profiles/:

  <tag>_bench_kernel_stats.csv     rocprofv3 --kernel-trace --stats kernel table, as emitted
  <tag>_bench_kernel_by_shape.csv  the same dispatches grouped by (kernel, grid, workgroup): one kernel template serves
                                   several GEMM shapes, the per-shape average is what bench.py's roofline object quotes
  <tag>_bench_hbm_traffic.csv      --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs), per (kernel, grid), with the
                                   gfx950 correction of MI355X_MICROARCH.md (HBM): FETCH_SIZE counts 128-byte requests as 64
                                   bytes for wide coalesced reads -> x2; WRITE_SIZE exact for 16-B/lane stores and float atomics

usage: python tools/make_profiles.py gpurun_out/r1c r01
"""
import csv
import os
import shutil
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, 'profiles')
os.makedirs(out, exist_ok=True)


def short(name):
    return name.split('(')[0][:80]


stats = os.path.join(src, 'stats', 'bench_kernel_stats.csv')
if os.path.exists(stats):
    shutil.copy(stats, os.path.join(out, f'{tag}_bench_kernel_stats.csv'))

trace = os.path.join(src, 'stats', 'bench_kernel_trace.csv')
if os.path.exists(trace):
    groups = defaultdict(list)
    for r in csv.DictReader(open(trace)):
        key = (short(r['Kernel_Name']), int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']),
               int(r['Workgroup_Size_X']), int(r['LDS_Block_Size']), int(r['VGPR_Count']) + int(r['Accum_VGPR_Count']))
        groups[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    rows = sorted(groups.items(), key=lambda kv: -sum(kv[1]))
    with open(os.path.join(out, f'{tag}_bench_kernel_by_shape.csv'), 'w') as f:
        f.write('# rocprofv3 --kernel-trace: dispatches grouped by (kernel, total grid threads, workgroup size)\n')
        f.write('kernel,grid_threads,workgroup,lds_bytes,vgprs,launches,total_us,avg_us,min_us,max_us\n')
        for (name, grid, wg, lds, vg), d in rows:
            f.write(f'"{name}",{grid},{wg},{lds},{vg},{len(d)},{sum(d) / 1e3:.1f},{sum(d) / len(d) / 1e3:.2f},{min(d) / 1e3:.2f},{max(d) / 1e3:.2f}\n')


def pmc(path, counter):
    acc = defaultdict(list)
    if not os.path.exists(path):
        return acc
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            acc[(short(r['Kernel_Name']), int(r['Grid_Size']))].append(float(r['Counter_Value']))
    return acc


fetch = pmc(os.path.join(src, 'fetch', 'bench_counter_collection.csv'), 'FETCH_SIZE')
write = pmc(os.path.join(src, 'write', 'bench_counter_collection.csv'), 'WRITE_SIZE')
if fetch or write:
    keys = sorted(set(fetch) | set(write), key=lambda k: -(2 * sum(fetch.get(k, [0])) + sum(write.get(k, [0]))))
    with open(os.path.join(out, f'{tag}_bench_hbm_traffic.csv'), 'w') as f:
        f.write('# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, tools/profile_round.sh) -- python3 bench.py '
                '--steps 50 --warmup 5 --no-cpu-baseline --no-b256 --no-c1\n')
        import hashlib, glob
        h = hashlib.sha256()
        for p_ in sorted(glob.glob(os.path.join(root, 'sibrar---single-branch-recommender_amd', 'csrc', '*.h*'))):
            h.update(os.path.basename(p_).encode())
            h.update(open(p_, 'rb').read())
        f.write(f'# csrc_sha16={h.hexdigest()[:16]} (bench.py csrc_sha16(): the kernel sources these counters were collected with)\n')
        f.write('# counters are KiB per dispatch; corrected_MB = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / 1e6 '
                '(gfx950: FETCH_SIZE tallies 128-byte requests of wide coalesced reads at 64 bytes)\n')
        f.write('kernel,grid_threads,launches,mean_FETCH_SIZE_KiB_raw,mean_WRITE_SIZE_KiB,corrected_HBM_MB_per_launch\n')
        for k in keys:
            fe, wr = fetch.get(k, []), write.get(k, [])
            mf = sum(fe) / len(fe) if fe else 0.0
            mw = sum(wr) / len(wr) if wr else 0.0
            f.write(f'"{k[0]}",{k[1]},{max(len(fe), len(wr))},{mf:.1f},{mw:.1f},{(2 * mf + mw) * 1024 / 1e6:.2f}\n')

# ---- matrix-core utilisation (third PMC pass): SQ_VALU_MFMA_BUSY_CYCLES counts busy cycles of the matrix pipe summed over the
# SIMDs; GRBM_GUI_ACTIVE is reported as the SUM over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back) -> elapsed cycles =
# GRBM_GUI_ACTIVE / 8; MfmaUtil = busy / (elapsed * 1024 SIMDs). MOPS counters * 512 = FLOP issued on the matrix cores.
def pmc_multi(path, counters):
    acc = {c: defaultdict(list) for c in counters}
    if not os.path.exists(path):
        return acc
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] in acc:
            acc[r['Counter_Name']][(short(r['Kernel_Name']), int(r['Grid_Size']))].append(float(r['Counter_Value']))
    return acc


names = ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_INSTS_VALU_MFMA_MOPS_F32', 'SQ_INSTS_VALU_MFMA_MOPS_F16', 'GRBM_GUI_ACTIVE']
mf = pmc_multi(os.path.join(src, 'mfma', 'bench_counter_collection.csv'), names)
if mf['GRBM_GUI_ACTIVE']:
    mean = lambda d, k: (sum(d[k]) / len(d[k])) if d.get(k) else 0.0
    keys = [k for k in mf['GRBM_GUI_ACTIVE'] if mean(mf['SQ_VALU_MFMA_BUSY_CYCLES'], k) > 0]
    keys.sort(key=lambda k: -mean(mf['GRBM_GUI_ACTIVE'], k) * len(mf['GRBM_GUI_ACTIVE'][k]))
    with open(os.path.join(out, f'{tag}_bench_mfma_util.csv'), 'w') as f:
        f.write('# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 '
                'SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-b256 --no-c1\n')
        f.write('# per dispatch means. elapsed_cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs); '
                'mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (elapsed_cycles * 1024 SIMDs); mfma_tflop = MOPS * 512 / 1e12; '
                'clock_GHz needs the kernel trace duration of the same dispatch (profiles/<tag>_bench_kernel_by_shape.csv)\n')
        f.write('kernel,grid_threads,launches,GRBM_GUI_ACTIVE,SQ_VALU_MFMA_BUSY_CYCLES,mfma_util,MOPS_F32,MOPS_F16,mfma_gflop_per_launch\n')
        for k in keys:
            ga, busy = mean(mf['GRBM_GUI_ACTIVE'], k), mean(mf['SQ_VALU_MFMA_BUSY_CYCLES'], k)
            m32, m16 = mean(mf['SQ_INSTS_VALU_MFMA_MOPS_F32'], k), mean(mf['SQ_INSTS_VALU_MFMA_MOPS_F16'], k)
            util = busy / (ga / 8.0 * 1024.0) if ga else 0.0
            f.write(f'"{k[0]}",{k[1]},{len(mf["GRBM_GUI_ACTIVE"][k])},{ga:.0f},{busy:.0f},{util:.4f},{m32:.0f},{m16:.0f},{(m32 + m16) * 512 / 1e9:.3f}\n')
line = os.path.join(src, 'bench_line.json')
if os.path.exists(line) and os.path.getsize(line) > 0:
    shutil.copy(line, os.path.join(out, f'{tag}_bench_line_profiled.json'))
print('wrote', sorted(p for p in os.listdir(out) if p.startswith(tag)))
