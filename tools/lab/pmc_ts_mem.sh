#!/bin/bash
# on the GPU box: HBM / L2 counters of the TN kernels over tools/lab/ts_bench.py (one shape). usage: bash tools/lab/pmc_ts_mem.sh 128x768
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SH=${1:-128x768}
O=gpurun_out/pmc_tsmem_$SH
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p1 -o s --output-format csv -- python3 tools/lab/ts_bench.py $SH > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/p2 -o s --output-format csv -- python3 tools/lab/ts_bench.py $SH > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum -d $O/p3 -o s --output-format csv -- python3 tools/lab/ts_bench.py $SH > $O/p3.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ('p1','p2','p3'):
    for f in glob.glob('$O/'+p+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: [0.0,0])
        for r in csv.DictReader(open(f)):
            if 'split_tn' not in r['Kernel_Name'] and 'gemm_ring' not in r['Kernel_Name']: continue
            a=acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
        for k,(v,n) in sorted(acc.items()): print(f'{p} {k:32s} per launch {v/n:.5g}  ({n} launches)')
PY
