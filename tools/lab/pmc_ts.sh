#!/bin/bash
# on the GPU box: SQ counter passes over tools/lab/ts_bench.py (one shape: "128x128" or "128x768"), summarised for the TN kernels.
# usage: bash tools/lab/pmc_ts.sh 128x768      (SBR_LAB_LIB / SBR_TN_SPLIT as for ts_bench.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SH=${1:-128x768}
O=gpurun_out/pmc_ts_$SH
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/p1 -o s --output-format csv -- python3 tools/lab/ts_bench.py $SH > $O/p1.log 2>&1
echo pass1 done
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/p2 -o s --output-format csv -- python3 tools/lab/ts_bench.py $SH > $O/p2.log 2>&1
echo pass2 done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVES -d $O/p3 -o s --output-format csv -- python3 tools/lab/ts_bench.py $SH > $O/p3.log 2>&1
echo pass3 done
python3 - <<PY
import csv, glob, collections
for p in ('p1','p2','p3'):
    for f in glob.glob('$O/'+p+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: [0.0,0])
        for r in csv.DictReader(open(f)):
            if 'split_tn' not in r['Kernel_Name'] and 'gemm_ring' not in r['Kernel_Name']: continue
            a=acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
        for k,(v,n) in sorted(acc.items()): print(f'{p} {k:32s} per launch {v/n:.4g}  ({n} launches)')
PY
