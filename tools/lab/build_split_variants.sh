#!/bin/bash
# Lab: variants of the K = N = 128 bf16-split GEMM kernel (-D switches of csrc/gemm_split_f32.hip) as libraries under tools/lab/bin/.
#   usage: bash tools/lab/build_split_variants.sh "tag -DSP_ABL_COALESCED" ...
set -e
cd "$(dirname "$0")/../.."
C=sibrar---single-branch-recommender_amd/csrc
mkdir -p tools/lab/bin
make -C $C -j8 > /dev/null
for v in "$@"; do
  set -- $v
  tag=$1; shift
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSBR_LAB "$@" -c $C/gemm_split_f32.hip -o tools/lab/bin/sp_$tag.o
  objs=$(ls $C/build/*.o | grep -v gemm_split_f32.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/lab/bin/sp_$tag.o -o tools/lab/bin/libsibrar_$tag.so
  echo built tools/lab/bin/libsibrar_$tag.so
done
