#!/bin/bash
# kernel micro-benchmarks (tools/bench_kernels.py gemm) for the product library and every tools/lab/bin/libsibrar_*.so
cd "$GRAFT_REPO_ROOT"
unset SBR_LAB_LIB
echo "== product"; python tools/bench_kernels.py gemm 2>/dev/null | grep "bf16x3"
for so in tools/lab/bin/libsibrar_*.so; do
  echo "== $so"; SBR_LAB_LIB=$so python tools/bench_kernels.py gemm 2>/dev/null | grep "bf16x3"
done
