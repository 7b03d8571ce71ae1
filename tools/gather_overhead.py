#!/usr/bin/env python
"""Fixed cost per workgroup and cost per round of lists of the cohort-statistics gather: launches of
32 768 rows x 20 000 cohort with 1, 2, 4, 8, 16 rounds of 1024 lists (300 members), linear fit of the
time per workgroup.  The slope is the rate of the walk alone, the intercept what a workgroup pays for its
row fill, the ramp and the uneven finish of its waves."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hyperion_amd import _lib  # noqa: E402

if len(sys.argv) > 1:  # another build of the library (A/B comparisons inside one GPU call)
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])

from hyperion_amd.score_norm.adapt_s_norm import gather_stats, topk_rows  # noqa: E402

torch.manual_seed(0)
rows, nc, k = 32768, 20000, 300
X = torch.randn((rows, nc), device="cuda") * 5 - 8
idx = topk_rows(torch.randn((16384, nc), device="cuda"), k, 0)
cus = torch.cuda.get_device_properties(0).multi_processor_count
res = []
for rounds in (1, 2, 4, 8, 16):
    nl = 1024 * rounds
    sub = idx[:, :nl].contiguous()
    best = 1e9
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gather_stats(X, None, sub, nl, 1e-5, k)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    per_wg = best * 1e3 / (rows / 2 / cus)
    res.append((rounds, per_wg))
    print("%2d rounds per lane: %.3f ms per launch, %.1f us per workgroup" % (rounds, best, per_wg), flush=True)
b, a = np.polyfit([r[0] for r in res], [r[1] for r in res], 1)
rate = 1024 * 2 * k * 4 / (b * 1e-6) * cus / 1e12  # gathered TB/s of the walk alone
print("per workgroup: %.1f us fixed + %.2f us per round of 1024 lists (fixed = %.2f rounds)" % (a, b, a / b))
print("the walk alone: %.1f TB/s gathered = %.3f of the %.0f TB/s LDS read roof" % (rate, rate / 157.2864, 157.2864))
