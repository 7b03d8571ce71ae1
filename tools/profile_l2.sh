#!/bin/bash
# L2 view of the cohort-statistics gather: requests the vector L1s send to L2, L2 hits and misses, per
# dispatch.  Usage: tools/profile_l2.sh <tag> [bench args]; writes gpurun_out/prof_<tag>/l2/.
# One counter set, no trace domain beside it; the program follows `--` directly.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-check --no-stages $*"
timeout -s KILL 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv \
  -d $OUT/l2 -- python3 $ROOT/bench.py $ARGS > $OUT/l2.log 2>&1 || { echo "l2 pass failed"; tail -5 $OUT/l2.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv \
  -d $OUT/fetch -- python3 $ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $OUT/fetch.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
for tag in ("l2", "fetch"):
    cc = glob.glob(os.path.join(sys.argv[1], tag, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(max(cc, key=os.path.getmtime))):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "gather" in k or "topk" in k:
            print(tag, k[:60], {c: "%.4g (n=%d)" % (sum(v) / len(v), len(v)) for c, v in cs.items()})
PY
