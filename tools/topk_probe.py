#!/usr/bin/env python
"""Time of the top-k selection alone: rows x cohort fp32 panel, top 300, ms per launch (median of 5).
HYPMI_LIB in the environment names another build of the library (for A/B runs inside one GPU call),
HYPMI_PROBE_DIST="scale,shift" the normal distribution of the scores (default 5,-8: the boundary of the
selection is positive and found in two full passes; 25,-60 puts it among the negative scores, eight passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hyperion_amd import _lib  # noqa: E402

if os.environ.get("HYPMI_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["HYPMI_LIB"])

from hyperion_amd.score_norm.adapt_s_norm import topk_rows  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for nc in [int(a) for a in sys.argv[2:]] or [20000]:
    scale, shift = (float(v) for v in os.environ.get("HYPMI_PROBE_DIST", "5,-8").split(","))  # N(shift, scale) scores
    A = torch.randn((rows, nc), device="cuda") * scale + shift
    ts = []
    for rep in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        topk_rows(A, 300, 0)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts[1:])[2]
    print("rows %d cohort %d: %.3f ms per launch = %.2f TB/s of row reads (HBM floor at 8 TB/s: %.3f ms)"
          % (rows, nc, ms, rows * nc * 4 / ms / 1e9, rows * nc * 4 / 8e12 * 1e3), flush=True)
    del A
