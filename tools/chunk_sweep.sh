#!/bin/bash
# Step time of the default AS-norm workload for several test-chunk sizes (run from the repository root on
# a GPU box).  Measured on one box: 32768 -> 643 ms, 65536 -> 641, 131072 -> 635, 262144 -> 637 (734 while
# one block of the Z part walked all 262 144 lists: its index lists, 600 B per test, did not stay in the
# caches that the workgroups of a launch share; blocks are now capped at 65 536 lists).
mkdir -p gpurun_out
for c in 32768 65536 131072 262144; do
  timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-stages --no-cpu-baseline --no-check --test-chunk $c \
    > gpurun_out/chunk_$c.json 2>> gpurun_out/chunk.err || exit 1
  python -c "
import json
d = json.loads(open('gpurun_out/chunk_$c.json').read().strip().splitlines()[-1])
print($c, round(d['ms_per_step'], 1), {k: round(v, 1) for k, v in d['stages_ms_per_step'].items()})
"
done
