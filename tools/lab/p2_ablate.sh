#!/bin/bash
# kernel times of the two-pass scorer's launches for the product library and every tools/lab/bin/libsibrar_abl*.so (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in ${LIBS:-product abl2 abl4 abl5 abl6}; do
  if [ "$lib" = product ]; then unset SBR_LAB_LIB; else export SBR_LAB_LIB=tools/lab/bin/libsibrar_$lib.so; fi
  rm -rf gpurun_out/p2abl_$lib
  rocprofv3 --kernel-trace --stats -d gpurun_out/p2abl_$lib -o p --output-format csv -- python3 tools/lab/scorer_routes_once.py ${1:-128} 1 > gpurun_out/p2abl_$lib.log 2>&1
  echo "== $lib"; grep -E "rescore|finalize2|select|score_max" gpurun_out/p2abl_$lib/p_kernel_stats.csv | cut -d, -f1,2,4 | cut -c1-40,100-200
done
