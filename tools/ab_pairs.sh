#!/bin/bash
# paired against separate T-part statistics, alternating inside one GPU call
for i in 1 2 3; do
  HYPMI_PAIR_STATS=0 timeout -k 10 150 python tools/ab_lib.py hyperion_amd/libhypmi.so 2>/dev/null | tail -n 1
  HYPMI_PAIR_STATS=1 timeout -k 10 150 python tools/ab_lib.py hyperion_amd/libhypmi.so 2>/dev/null | tail -n 1
done
