#!/usr/bin/env python
"""The Z-part shape of the cohort-statistics gather (few rows, 65 536 lists = 64 rounds per lane):
time per workgroup when every CU has exactly one workgroup (512 rows), ten (5120 rows), and the
benchmark's 10 000 rows (19.53 workgroups per CU), against the T-part shape's model of
11 us + 12.7 us per round."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hyperion_amd import _lib  # noqa: E402

if len(sys.argv) > 1:  # another build of the library (A/B comparisons inside one GPU call)
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from hyperion_amd.score_norm.adapt_s_norm import gather_stats, topk_rows  # noqa: E402

torch.manual_seed(0)
nc, k, nl = 20000, 300, 65536
cus = torch.cuda.get_device_properties(0).multi_processor_count
idx = topk_rows(torch.randn((nl, nc), device="cuda"), k, 0)
for rows in (2 * cus, 20 * cus, 10000, 40 * cus):
    X = torch.randn((rows, nc), device="cuda") * 5 - 8
    best = 1e9
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gather_stats(X, None, idx, nl, 1e-5, k)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    wg_per_cu = rows / 2 / cus
    print("%6d rows (%.2f workgroups per CU): %.3f ms per launch, %.1f us per workgroup slot, %.2f us per round"
          % (rows, wg_per_cu, best, best * 1e3 / wg_per_cu, best * 1e3 / wg_per_cu / 64), flush=True)

# the same shape with the normalisation inside the launch (hypmi_gather_apply)
from hyperion_amd.score_norm.adapt_s_norm import gather_apply  # noqa: E402

for rows in (10000, 40 * cus):
    X = torch.randn((rows, nc), device="cuda") * 5 - 8
    S = torch.randn((rows, nl), device="cuda")
    ldt = rows + 32 if rows % 512 == 0 else (rows + 3) // 4 * 4
    mt = torch.randn((nl, ldt), device="cuda")
    it = torch.rand((nl, ldt), device="cuda") + 0.5
    out = torch.empty((rows, nl), device="cuda")
    best = 1e9
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gather_apply(X, idx, nl, 1e-5, k, S, mt, it, ldt, out)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print("%6d rows, normalisation inside: %.3f ms per launch" % (rows, best), flush=True)
