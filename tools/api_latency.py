#!/usr/bin/env python
"""Wall time of the numpy-in / numpy-out drop-in call for the VoxCeleb1-O size case (4874 x 4874,
512-d, y_dim 150) next to the float64 oracle on the host, and where the time goes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import synth  # noqa: E402
from hyperion_amd import engine  # noqa: E402
from hyperion_amd.pdfs.plda import SPLDA  # noqa: E402
from oracle import plda_ref as O  # noqa: E402

n, d, K = 4874, 512, 150
m = synth.make_splda(d, K)
x = synth.length_norm(synth.sample_splda(m, 40, 122, 7))[:n]
plda = SPLDA(mu=m["mu"], V=m["V"], W=m["W"])


def wall(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, r


plda.llr_1vs1(x[:64], x[:64])  # plan creation, first-touch
t_ref, ref = wall(lambda: O.llr_1vs1(m, x, x), 2)
t_api, got = wall(lambda: plda.llr_1vs1(x, x))
t_norefine, _ = wall(lambda: plda.llr_1vs1(x, x, refine=False))
xd = engine.upload(x)
t_dev, S = wall(lambda: plda.llr_1vs1(xd, xd))
t_d2h, h = wall(lambda: S.cpu())
t_cast, _ = wall(lambda: h.numpy().astype(np.float64))
err = np.abs(got - ref) / np.abs(ref)
print("oracle (host float64)      %8.1f ms" % t_ref)
print("numpy API, refine (default)%8.1f ms   max relative error %.2e" % (t_api, err.max()))
print("numpy API, refine=False    %8.1f ms" % t_norefine)
print("device tensors in/out      %8.1f ms" % t_dev)
print("  D2H of the fp32 matrix   %8.1f ms   host fp32 -> float64 %8.1f ms" % (t_d2h, t_cast))
