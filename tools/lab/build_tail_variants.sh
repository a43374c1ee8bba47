#!/bin/bash
# Lab: variants of csrc/fused_tail.hip as separate libraries under tools/lab/bin/: build_tail_variants.sh <tag> -- <-D flags>
set -e
cd "$(dirname "$0")/../.."
C=sibrar---single-branch-recommender_amd/csrc
mkdir -p tools/lab/bin
make -C $C -j8 > /dev/null
tag=$1; shift; [ "$1" = "--" ] && shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $C/fused_tail.hip -o tools/lab/bin/ft_$tag.o
objs=$(ls $C/build/*.o | grep -v fused_tail.o)
hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/lab/bin/ft_$tag.o -o tools/lab/bin/libsibrar_ft$tag.so
echo built tools/lab/bin/libsibrar_ft$tag.so
