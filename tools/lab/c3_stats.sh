#!/bin/bash
# Lab: kernel-trace statistics of the c3 configuration (tools/run_configs.py c3 --train-only --batch=4096). Output: gpurun_out/c3_stats.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c3_stats
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o b --output-format csv -- python3 tools/run_configs.py c3 --train-only --batch=4096 > $O/log.txt 2>&1
cp $O/b_kernel_stats.csv gpurun_out/c3_stats.csv
rm -f $O/b_kernel_trace.csv
