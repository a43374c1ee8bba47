#!/bin/bash
# on the GPU box: L2 / HBM counters of every GEMM kernel of tools/bench_kernels.py gemm, per kernel symbol. usage: bash tools/lab/pmc_gemm_mem.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_gemm_mem
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p1 -o s --output-format csv -- python3 tools/bench_kernels.py gemm > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum -d $O/p3 -o s --output-format csv -- python3 tools/bench_kernels.py gemm > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum -d $O/p4 -o s --output-format csv -- python3 tools/bench_kernels.py gemm > $O/p4.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ('p1','p3','p4'):
    for f in glob.glob('$O/'+p+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: [0.0,0])
        for r in csv.DictReader(open(f)):
            n=r['Kernel_Name']
            if 'gemm' not in n: continue
            a=acc[(n[:48], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X','?'), r['Counter_Name'])]; a[0]+=float(r['Counter_Value']); a[1]+=1
        for k,(v,n) in sorted(acc.items()): print(f'{p} {k[0]:48s} grid {k[1]:>8s} {k[2]:36s} per launch {v/n:.5g}  ({n})')
PY
