#!/bin/bash
# Instruction mix of the top-k kernel alone: tools/profile_topk.sh <tag> [rows cohort ...]; writes
# gpurun_out/prof_<tag>/.  One counter set per run, no trace domain beside it; the program follows `--`.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${*:-65536 20000}"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
  --output-format csv -d $OUT/sq -- python3 $ROOT/tools/topk_probe.py $ARGS > $OUT/sq.log 2>&1 || { echo "sq pass failed"; tail -5 $OUT/sq.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/sq2 -- python3 $ROOT/tools/topk_probe.py $ARGS > $OUT/sq2.log 2>&1 || { echo "sq2 pass failed"; tail -5 $OUT/sq2.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
for tag in ("sq", "sq2"):
    cc = glob.glob(os.path.join(sys.argv[1], tag, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(max(cc, key=os.path.getmtime))):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "topk" in k:
            print(tag, k[:70], {c: "%.4g (n=%d)" % (sum(v) / len(v), len(v)) for c, v in cs.items()})
PY
