#!/bin/bash
# Lab: rocprofv3 kernel durations of the bf16-split TN kernel for the product library and the variants named on the command line
# (e.g. prof_ts.sh 3 4 -> tools/lab/bin/libsibrar_ts3.so, ..ts4.so; build them first). Output: gpurun_out/ts_prof.txt
cd "$(dirname "$0")/../.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/prof_ts
mkdir -p $O
: > gpurun_out/ts_prof.txt
for v in product "$@"; do
  if [ "$v" = product ]; then unset SBR_LAB_LIB; else export SBR_LAB_LIB=tools/lab/bin/libsibrar_ts$v.so; fi
  rocprofv3 --kernel-trace --stats -d $O/$v -o t --output-format csv -- python3 tools/lab/ts_bench.py > $O/$v.log 2>&1 || { echo "rocprofv3 failed for $v" >> gpurun_out/ts_prof.txt; exit 1; }
  python3 tools/lab/prof_ts_summary.py $O/$v $v >> gpurun_out/ts_prof.txt
done
cat gpurun_out/ts_prof.txt
