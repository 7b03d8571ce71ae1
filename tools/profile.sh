#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box.  Usage: tools/profile.sh <tag> [bench args]
# Writes gpurun_out/prof_<tag>/{trace,fetch,write,sq}/ ; the summaries are copied to profiles/ by hand.
# For the headline workload pass --no-asnorm-stage: the appended AS-norm stage launches the same score
# kernel on small shapes, which would be averaged into the per-dispatch means of the counter passes.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-check $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $OUT/fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py $ARGS > $OUT/write.log 2>&1 || { echo "write pass failed"; tail -5 $OUT/write.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $ROOT/bench.py $ARGS > $OUT/sq.log 2>&1 || { echo "sq pass failed"; tail -5 $OUT/sq.log; exit 1; }
find $OUT -name "*.csv" | head -40
