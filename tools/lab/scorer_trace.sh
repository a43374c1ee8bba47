#!/bin/bash
# Kernel trace of the in-process A/B (tools/lab/scorer_ab.py): per kernel name and grid size, launches and mean duration — tells the
# main kernel of two scorer builds (different grids) and their final-selection kernels apart.   usage: bash tools/lab/scorer_trace.sh [D] [excl]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/scorer_trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/t -o t --output-format csv -- python3 tools/lab/scorer_ab.py ${1:-128} ${2:-0} 5 > $O/ab.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/scorer_trace/t/**/t_kernel_trace.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'score_topk' in n or 's5_' in n:
            acc[(n[:60], r['Grid_Size_X'], r['Workgroup_Size_X'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for k, v in sorted(acc.items()):
        v = sorted(v)
        print(f'{k[0]:60s} grid {k[1]:>8s} wg {k[2]:>5s}  n {len(v):4d}  median {v[len(v) // 2] / 1e3:9.1f} us  min {v[0] / 1e3:9.1f}')
PY
tail -4 $O/ab.log
