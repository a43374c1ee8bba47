#!/bin/bash
# Lab: time the complete scorer (c2 shape, SBR_ST_PRE=16) with every variant library under tools/lab/bin + the product library, then a
# kernel trace of the product library (main kernel vs final-selection kernel).   usage (GPU box): bash tools/lab/run_scorer_variants.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SBR_ST_PRE=${SBR_ST_PRE:-16}
cat > /tmp/one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch, sibrar_amd as S
ops = S.ops
U, I, D = 100_000, 50_000, 128
g = torch.Generator().manual_seed(1)
u = (torch.randn(U, D, generator=g) / 8).half().cuda(); it = (torch.randn(I, D, generator=g) / 8).half().cuda()
fn = lambda: ops.score_topk_f16(u, it, 20)
for _ in range(6): fn()
evs = []
for _ in range(12):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); evs.append((a, b))
torch.cuda.synchronize()
ts = sorted(x.elapsed_time(y) for x, y in evs)
print(os.environ.get('SBR_LAB_LIB', 'product'), f'{ts[len(ts) // 2]:.3f} ms', flush=True)
PY
python3 /tmp/one.py
for l in tools/lab/bin/libsibrar_*.so; do SBR_LAB_LIB=$l python3 /tmp/one.py; done
rocprofv3 --kernel-trace --stats -d gpurun_out/sc_trace -o sc --output-format csv -- python3 /tmp/one.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/sc_trace/**/sc_kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
