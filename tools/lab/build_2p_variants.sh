#!/bin/bash
# Lab: variants of the two-pass scorer (-D switches of csrc/score_topk_f16_2p.hip, e.g. S2_ABL=1) as separate libraries under
# tools/lab/bin/, selected through SBR_LAB_LIB.   usage: bash tools/lab/build_2p_variants.sh "tag -DS2_ABL=1" "tag2 ..."
set -e
cd "$(dirname "$0")/../.."
C=sibrar---single-branch-recommender_amd/csrc
mkdir -p tools/lab/bin
make -C $C -j8 > /dev/null
for v in "$@"; do
  set -- $v
  tag=$1; shift
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSBR_LAB "$@" -c $C/score_topk_f16_2p.hip -o tools/lab/bin/p2_$tag.o
  objs=$(ls $C/build/*.o | grep -v score_topk_f16_2p.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/lab/bin/p2_$tag.o -o tools/lab/bin/libsibrar_$tag.so
  echo built tools/lab/bin/libsibrar_$tag.so
done
