#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box.  Usage: tools/profile.sh <tag> [bench args]
# Writes gpurun_out/prof_<tag>/{trace,fetch,write,sq,sqg}/ ; tools/summarize_prof.py turns that into the
# summary kept under profiles/.  Counter passes are separate runs (one --pmc set each, no trace domains
# beside them); the program follows `--` directly.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-check --no-stages $*"
run() {  # <dir> <rocprofv3 options...>
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py $ARGS > $OUT/$name.log 2>&1 \
    || { echo "$name pass failed"; tail -5 $OUT/$name.log; exit 1; }
}
run trace --kernel-trace --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
# what the cohort-statistics gather issues: vector and LDS instruction counts and busy cycles
run sqg --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
find $OUT -name "*.csv" | head -40
