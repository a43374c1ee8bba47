#!/bin/bash
# Lab: ablations of the direct TN kernel (TD_ABL) as separate libraries under tools/lab/bin/
set -e
cd "$(dirname "$0")/../.."
C=sibrar---single-branch-recommender_amd/csrc
mkdir -p tools/lab/bin
make -C $C -j8 > /dev/null
for v in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTD_ABL=$v -c $C/gemm_tn_direct_f32.hip -o tools/lab/bin/td_$v.o
  objs=$(ls $C/build/*.o | grep -v gemm_tn_direct_f32.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/lab/bin/td_$v.o -o tools/lab/bin/libsibrar_td$v.so
  echo built tools/lab/bin/libsibrar_td$v.so
done
