#!/bin/bash
# Knock-out variants of the ring GEMM (one phase removed each) for phase-overlap analysis: bin/ring_{nomfma,nodma,nolds,nobar}
set -e
cd "$(dirname "$0")"
SRC="../../sibrar---single-branch-recommender_amd/csrc"
mkdir -p bin gen
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -I$SRC -Wno-unused-value -Wno-unused-result"
python3 - <<'PY'
import re
src = open('../../sibrar---single-branch-recommender_amd/csrc/gemm_ring_f32.hip').read()
v = re.sub(r"acc\[i\]\[j\] = __builtin_amdgcn_mfma_f32_32x32x2f32\(fa\[i\]\.(\w), fb\[j\]\.(\w), acc\[i\]\[j\], 0, 0, 0\);",
           r"acc[i][j][0] += fa[i].\1 * fb[j].\2;", src)
open('gen/ring_nomfma.hip', 'w').write(v)
v = src.replace("    if (issued < q_total) {\n      issue(issued % NS);", "    if (issued < q_total && issued < NS) {\n      issue(issued % NS);")
v = v.replace("if (q + NS - 1 <= q_total) ring_wait_vmcnt<(NS - 2) * PER_T>();\n    else ring_wait_vmcnt<0>();", "ring_wait_vmcnt<0>();")
open('gen/ring_nodma.hip', 'w').write(v)
# fragments from registers instead of LDS (k-major reads knocked out)
v = src.replace("fa[i] = make_float4(p[0], p[BM], p[2 * BM], p[3 * BM]);", "fa[i] = make_float4(1.f + kq, 2.f, 3.f + i, 4.f); (void)p;")
v = v.replace("fb[j] = make_float4(p[0], p[BN], p[2 * BN], p[3 * BN]);", "fb[j] = make_float4(1.f + kq, 2.f, 3.f + j, 4.f); (void)p;")
open('gen/ring_nolds.hip', 'w').write(v)
v = src.replace("    __builtin_amdgcn_s_barrier();                              // every wave's part of slab q landed; slab q - 1 consumed", "")
open('gen/ring_nobar.hip', 'w').write(v)
PY
for v in nomfma nodma nolds nobar; do
  hipcc $FLAGS $SRC/gemm_f32.hip gen/ring_$v.hip gemm_lab_main.cpp -o bin/ring_$v &
done
wait
ls bin
