#!/bin/bash
# tools/gather_zshape.py with each of the given builds of the library, in turn, inside one GPU call
for L in "$@"; do
  echo "== $L"
  timeout -k 10 150 python tools/gather_zshape.py $L 2>/dev/null | tail -n 3 || exit 1
done
