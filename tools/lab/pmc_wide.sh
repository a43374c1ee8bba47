#!/bin/bash
# PMC passes over the wide bf16-split kernel (GPU box).  usage: bash tools/lab/pmc_wide.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_wide; rm -rf $O; mkdir -p $O
ARGS="tools/lab/wide_one.py 90112 512 512 1"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES -d $O/a -o w --output-format csv -- python3 $ARGS > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/b -o w --output-format csv -- python3 $ARGS > $O/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for p in ('a', 'b'):
    for f in glob.glob(f'gpurun_out/pmc_wide/{p}/**/w_counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'gemm_split_wide' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(p, k, f'{sum(v) / len(v):.4g}', len(v))
PY
