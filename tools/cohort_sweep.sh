#!/bin/bash
# Step time of the AS-norm pipeline against the cohort size (10 000 enrol x 131 072 test, nbest 300), with the
# kernels that ran: tools/cohort_sweep.sh <out file> [cohort sizes...]
OUT=$1; shift
SIZES=${@:-"20000 24000 40960 50000"}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
: > $OUT
for NC in $SIZES; do
  echo "== cohort $NC ==" >> $OUT
  timeout -k 10 280 python $ROOT/bench.py --workload asnorm_10k_128k --test-chunk 131072 --n-coh $NC --steps 2 --warmup 1 \
      --no-cpu-baseline --no-stages 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('ms_per_step %.1f  parity %s' % (d['ms_per_step'], d.get('parity', {}).get('max_err_over_max_abs_1')))
print('stages', {k: round(v, 1) for k, v in d['stages_ms_per_step'].items()})" >> $OUT || { echo "cohort $NC failed" >> $OUT; exit 1; }
  (cd /tmp && export TMPDIR=/tmp && timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cs_$NC -- \
      python3 $ROOT/bench.py --workload asnorm_10k_128k --test-chunk 131072 --n-coh $NC --steps 1 --warmup 0 --no-cpu-baseline \
      --no-check --no-stages > /tmp/cs_$NC.log 2>&1) || { echo "trace of cohort $NC failed" >> $OUT; exit 1; }
  python - /tmp/cs_$NC >> $OUT <<'PY'
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:6]:
    print("  %-64s calls %3s avg %10.1f us  %5s %%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
cat $OUT
