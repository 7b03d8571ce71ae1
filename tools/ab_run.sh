#!/bin/bash
# Alternates two builds of the library inside one GPU call: tools/ab_run.sh <lib A> <lib B> [rounds [n_test]]
# (HYPMI_TEST_CHUNK in the environment sets the tests per chunk)
A=$1; B=$2; N=${3:-2}; NT=${4:-196608}
for i in $(seq $N); do
  timeout -k 10 150 python tools/ab_lib.py $A $NT 2>/dev/null | tail -n 1 || exit 1
  timeout -k 10 150 python tools/ab_lib.py $B $NT 2>/dev/null | tail -n 1 || exit 1
done
