#!/bin/bash
# Lab: ablations of the bf16-split TN kernel (TS_ABL: 1 no loads in the loop, 2 no split / plane stores, 3 both) as separate
# libraries under tools/lab/bin/ (TS_TAG=x appends x to the library name); extra -D flags after "--" (e.g. build_ts_variants.sh 0 -- -DTS_FOO=1 builds libsibrar_ts0.so with it)
set -e
cd "$(dirname "$0")/../.."
C=sibrar---single-branch-recommender_amd/csrc
mkdir -p tools/lab/bin
make -C $C -j8 > /dev/null
vs=(); extra=()
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; extra=("$@"); break; fi
  vs+=("$1"); shift
done
for v in "${vs[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTS_ABL=$v "${extra[@]}" -c $C/gemm_split_tn_f32.hip -o tools/lab/bin/ts_$v.o
  objs=$(ls $C/build/*.o | grep -v gemm_split_tn_f32.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/lab/bin/ts_$v.o -o tools/lab/bin/libsibrar_ts$v$TS_TAG.so
  echo built tools/lab/bin/libsibrar_ts$v$TS_TAG.so
done
