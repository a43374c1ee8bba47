#!/bin/bash
# Lab: variants of the narrow-wave scorer (prefetch distances S5_PF1 / S5_PF2, loader waves S5_NL) as separate libraries under
# tools/lab/bin/, selected by tools/scorer_lab.py through SBR_LAB_LIB.   usage: bash tools/lab/build_scorer_variants.sh "1 1 2" "3 2 2" ...
set -e
cd "$(dirname "$0")/../.."
C=sibrar---single-branch-recommender_amd/csrc
mkdir -p tools/lab/bin
make -C $C -j8 > /dev/null
for v in "$@"; do
  set -- $v
  tag="pf$1_$2_nl$3${4:+_$4}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DS5_PF1=$1 -DS5_PF2=$2 -DS5_NL=$3 $5 -c $C/score_topk_f16_n.hip -o tools/lab/bin/n_$tag.o
  objs=$(ls $C/build/*.o | grep -v score_topk_f16_n.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/lab/bin/n_$tag.o -o tools/lab/bin/libsibrar_$tag.so
  echo built tools/lab/bin/libsibrar_$tag.so
done
