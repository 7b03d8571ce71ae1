#!/usr/bin/env python
"""Runs the top-k selection several times over one panel of many rows and reports the rows whose lists
differ between runs or from a stable descending argsort, with what is known about each such row
(ties at the boundary, how many entries are wrong).  HYPMI_LIB names another build of the library.

    python tools/topk_repeat.py n nc k discard scale shift [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from hyperion_amd import _lib  # noqa: E402

if os.environ.get("HYPMI_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["HYPMI_LIB"])

from hyperion_amd.score_norm.adapt_s_norm import lists_to_numpy, topk_rows  # noqa: E402

n, nc, k, disc = (int(a) for a in sys.argv[1:5])
scale, shift = float(sys.argv[5]), float(sys.argv[6])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 4
rng = np.random.default_rng(n + nc)
A = rng.standard_normal((n, nc), dtype=np.float32) * np.float32(scale) + np.float32(shift)
Ad = torch.from_numpy(A).cuda()
runs = [lists_to_numpy(topk_rows(Ad, k, disc), k) for _ in range(reps)]
torch.cuda.synchronize()
order = np.argsort(-A, axis=1, kind="stable")[:, disc:disc + k]
want = np.sort(order, axis=1)
print("lib", _lib.LIB_PATH, "case", sys.argv[1:7])
for r, got in enumerate(runs):
    bad = np.nonzero((np.sort(got, axis=1) != want).any(axis=1))[0]
    unstable = np.nonzero((got != runs[0]).any(axis=1))[0]
    print("run %d: %d rows differ from the argsort, %d rows differ from run 0 (as stored)" % (r, bad.size, unstable.size))
    for row in bad[:6]:
        g, w = set(got[row].tolist()), set(want[row].tolist())
        vals = np.sort(A[row])[::-1]
        print("   row %d (mod 4 = %d): missing %s extra %s dup %d; boundary values %s" % (
            row, row % 4, sorted(w - g)[:5], sorted(g - w)[:5], k - len(g),
            vals[max(disc + k - 2, 0):disc + k + 2].tolist()))
    for row in unstable[:4]:
        d = np.nonzero(got[row] != runs[0][row])[0]
        print("   unstable row %d: positions %s now %s was %s" % (row, d[:6].tolist(), got[row][d[:6]].tolist(),
                                                                   runs[0][row][d[:6]].tolist()))
