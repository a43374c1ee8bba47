#!/bin/bash
# Lab: kernel-trace stats of the bench's training step only (no scoring), written to gpurun_out/step_stats_<tag>.csv. usage: step_stats.sh tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-a}
O=gpurun_out/step_stats_$T
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o b --output-format csv -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-b256 --no-c1 --no-scoring --no-configs > $O/log.txt 2>&1
cp $O/b_kernel_stats.csv gpurun_out/step_stats_$T.csv
grep '^{' $O/log.txt | tail -1 > gpurun_out/step_line_$T.json
rm -f $O/b_kernel_trace.csv
