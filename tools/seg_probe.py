#!/usr/bin/env python
"""Where the segmented gather loses time against the undivided one: the same launch (32 768 rows, 8192 lists of
300 members) on random lists and on lists that take the same number of members from every segment (no lane
waits for a longer part of another list of its wave), per cohort size."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hyperion_amd.score_norm.adapt_s_norm import gather_stats, seg_plan, topk_rows  # noqa: E402

rows, lists, k = 32768, 8192, 300
for nc in [int(a) for a in sys.argv[1:]] or [20000, 24000, 50000]:
    X = torch.randn((rows, nc), device="cuda") * 5 - 8
    plan = seg_plan(nc, k)
    S, seg_len = (plan[0], plan[1]) if plan else (1, nc)
    for kind in ("random", "balanced"):
        sel = torch.randn((lists, nc), device="cuda")
        if kind == "balanced":  # exactly k / S members from every segment: lift them above everything else
            for s in range(S):
                part = sel[:, s * seg_len:min(nc, (s + 1) * seg_len)]
                kk = k // S + (1 if s < k % S else 0)
                thr = torch.topk(part, kk, dim=1).values[:, -1:]
                part += 100.0 * (part >= thr)
        idx = topk_rows(sel, k, 0)
        del sel
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gather_stats(X, None, idx, lists, 1e-5, k)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = sorted(ts[1:])[1]
        print("cohort %d (%d segment(s)) %s lists: %.3f ms per launch, %.1f us per workgroup of 8 rounds"
              % (nc, S, kind, ms, ms * 1e3 / (rows / 2 / 256)), flush=True)
