#!/bin/bash
# Profiles of one round, collected on the GPU box (gpurun): kernel trace + the PMC passes of MI355X_MICROARCH.md (HBM section:
# FETCH_SIZE and WRITE_SIZE in separate passes; SQ / GRBM counters for the matrix-core utilisation in a third) of ONE command —
# bench.py's training step and its fused scoring pass. Raw output lands under gpurun_out/<tag>/, tools/make_profiles.py condenses
# it into profiles/<tag>_*.  usage: bash tools/profile_round.sh r02
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
ARGS="bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-b256 --no-c1 --no-configs"
rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 $ARGS > $O/stats.log 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o bench --output-format csv -- python3 $ARGS > $O/fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o bench --output-format csv -- python3 $ARGS > $O/write.log 2>&1
echo "write pass done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE -d $O/mfma -o bench --output-format csv -- python3 $ARGS > $O/mfma.log 2>&1
echo "mfma pass done"
grep '^{' $O/stats.log | tail -1 > $O/bench_line.json || true
python3 tools/make_profiles.py $O $TAG
# ---- the other one-GPU configurations at their own shapes: the c5 shard (100k x 25k x 256 scoring) and the c3 step (B = 4096)
for CFG in c5 c3; do
  P=$O/$CFG
  mkdir -p $P
  CARGS="bench.py --only $CFG --steps 30 --warmup 5"
  rocprofv3 --kernel-trace --stats -d $P/stats -o bench --output-format csv -- python3 $CARGS > $P/stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/fetch -o bench --output-format csv -- python3 $CARGS > $P/fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/write -o bench --output-format csv -- python3 $CARGS > $P/write.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE -d $P/mfma -o bench --output-format csv -- python3 $CARGS > $P/mfma.log 2>&1
  grep '^{' $P/stats.log | tail -1 > $P/bench_line.json || true
  python3 tools/make_profiles.py $P ${TAG}_$CFG
  echo "$CFG passes done"
done
