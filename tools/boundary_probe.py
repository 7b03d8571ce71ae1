#!/usr/bin/env python
"""Where the selection boundary of the benchmark's test-vs-cohort score rows lies: sign and size of the
score at rank nbest, the row's extreme scores, and how many counting passes a bit-by-bit search from the
sign bit down needs before at most 512 scores are left in its bracket (thresholds outside the row's range
of scores cost no pass).  Workload: bench.py's asnorm chain + SPLDA, a few thousand test rows."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from hyperion_amd.backend import PLDABackend  # noqa: E402
from hyperion_amd.pdfs.plda import SPLDA  # noqa: E402
from hyperion_amd.transforms import LDA, CentWhiten, LNorm, TransformList  # noqa: E402

cfg = dict(bench.WORKLOADS["asnorm_10k_128k"])
chain, model = bench.build_workload(cfg)
plda = SPLDA(mu=model["mu"], V=model["V"], W=model["W"])
preproc = TransformList([CentWhiten(mu=chain[0][1], T=chain[0][2], name="cw"), LNorm(name="ln"),
                         LDA(mu=chain[2][1], T=chain[2][2], name="lda")])
be = PLDABackend(plda, preproc, nbest=cfg["nbest"])
n_test = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
xt, _ = bench.device_xvectors(n_test, cfg["d"], seed=3)
xc, _ = bench.device_xvectors(cfg["n_coh"], cfg["d"], seed=2)
A = be.score(xt, xc)  # [n_test, n_cohort] raw scores
A = A if isinstance(A, torch.Tensor) else torch.from_numpy(A).cuda()
k = cfg["nbest"]
top = torch.topk(A, k, dim=1).values
kth = top[:, -1].cpu().numpy()
mx, mn = A.max(dim=1).values.cpu().numpy(), A.min(dim=1).values.cpu().numpy()
print("rows %d cohort %d nbest %d" % (A.shape[0], A.shape[1], k))
print("score at rank nbest: min %.3g  5%% %.3g  median %.3g  95%% %.3g  max %.3g; positive in %.1f %% of rows" % (
    kth.min(), np.quantile(kth, 0.05), np.median(kth), np.quantile(kth, 0.95), kth.max(), 100 * (kth > 0).mean()))
print("row maximum median %.3g, row minimum median %.3g, row mean %.3g, row std %.3g" % (
    np.median(mx), np.median(mn), float(A.mean()), float(A.std(dim=1).mean())))


def key_of(x):
    b = x.view(np.uint32)
    return np.where(b >> 31 != 0, ~b, b | np.uint32(0x80000000)).astype(np.uint64)


passes = []
Ah = A[:256].cpu().numpy()
for r in range(Ah.shape[0]):
    keys = key_of(Ah[r])
    kmin, kmax = keys.min(), keys.max()
    prefix, c_lo, c_hi, n = 0, keys.size, 0, 0
    for b in range(31, -1, -1):
        if c_lo - c_hi <= 512 and b < 31:
            break
        cand = prefix | (1 << b)
        if cand > kmax:
            c = 0
        elif cand <= kmin:
            c = keys.size
        else:
            c = int((keys >= cand).sum())
            n += 1
        if c > k - 1:
            prefix, c_lo = cand, c
        else:
            c_hi = c
    passes.append(n)
print("counting passes per row over %d rows: mean %.2f, histogram %s" % (len(passes), np.mean(passes),
                                                                      np.bincount(passes).tolist()))
