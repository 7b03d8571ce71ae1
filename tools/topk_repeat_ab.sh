#!/bin/bash
# the many-row repeat check of tools/topk_repeat.py with the in-tree library and with tools/ab/old.so
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for lib in hyperion_amd/libhypmi.so tools/ab/old.so; do
  for c in "20000 900 50 0 3.0 0.0" "4000 20000 300 2 5.0 -8.0"; do
    HYPMI_LIB=$lib timeout -k 10 120 python3 tools/topk_repeat.py $c 4 || exit 1
  done
done
