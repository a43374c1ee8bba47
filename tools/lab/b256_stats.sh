#!/bin/bash
# Lab: kernel-trace statistics of the c2 step at the reference's default batch 256. Output: gpurun_out/b256_stats.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/b256_stats
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o b --output-format csv -- python3 bench.py --batch-size 256 --steps 200 --warmup 5 --no-cpu-baseline --no-b256 --no-c1 --no-scoring --no-configs > $O/log.txt 2>&1
cp $O/b_kernel_stats.csv gpurun_out/b256_stats.csv
grep '^{' $O/log.txt | tail -1 > gpurun_out/b256_line.json
rm -f $O/b_kernel_trace.csv
