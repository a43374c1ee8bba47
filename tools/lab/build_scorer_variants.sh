#!/bin/bash
# Lab: variants of the fused scorer, always built with -DSBR_LAB (the SBR_ST_DEBUG / SBR_ST_PRE environment switches and the ablation
# instantiations exist only in these libraries, never in the product library): variants of the fused scorer (any -D switches of csrc/score_topk_f16_n.hip: S5_PF1 / S5_PF2 prefetch distances, S5_NL loader
# waves, S5_CAPH, S5_RF, S5_NOSTORE ...) as separate libraries under tools/lab/bin/, selected through SBR_LAB_LIB.
#   usage: bash tools/lab/build_scorer_variants.sh "tag -DS5_RF=16 ..." "tag2 ..."
set -e
cd "$(dirname "$0")/../.."
C=sibrar---single-branch-recommender_amd/csrc
mkdir -p tools/lab/bin
make -C $C -j8 > /dev/null
for v in "$@"; do
  set -- $v
  tag=$1; shift
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSBR_LAB "$@" -c $C/score_topk_f16_n.hip -o tools/lab/bin/n_$tag.o
  objs=$(ls $C/build/*.o | grep -v score_topk_f16_n.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/lab/bin/n_$tag.o -o tools/lab/bin/libsibrar_$tag.so
  echo built tools/lab/bin/libsibrar_$tag.so
done
