#!/bin/bash
# on the GPU box: time every variant library under tools/lab/bin for both benchmark shapes
for lib in tools/lab/bin/libsibrar_*.so; do
  nl=$(echo $lib | sed 's/.*_nl\([0-9]\).*/\1/')
  for shape in "128 50000" "256 25000"; do
    echo "== $lib D,I = $shape"
    SBR_LAB_LIB=$lib SBR_LAB_MAXW=$((16 - nl)) timeout -k 10 120 python tools/scorer_lab.py $shape 100000 ${@:-time ablate} 2>&1 | grep -v amdgpu.ids
  done
done
