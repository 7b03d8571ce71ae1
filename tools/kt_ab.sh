#!/bin/bash
# kernel-trace of tools/ab_lib.py for each given library: average duration of the top-k kernel
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for l in "$@"; do
  rm -rf /tmp/kt_$l
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$l -- python3 $ROOT/tools/ab_lib.py $ROOT/tools/ab/$l.so 196608 > /tmp/kt_$l.log 2>&1 || { echo fail $l; tail -5 /tmp/kt_$l.log; exit 1; }
  f=$(find /tmp/kt_$l -name "*kernel_stats.csv" | head -1)
  echo "== $l"; grep -E "topk|Name" $f | cut -c1-200
done
