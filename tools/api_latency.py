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

# ---- adaptive S-norm through the numpy API at an evaluation-set size --------------------------------
from hyperion_amd.score_norm import AdaptSNorm  # noqa: E402

ne, nt, nc, nbest = 2000, 20000, 5000, 200
m2 = synth.make_splda(200, K)
xe, xt, xc = (synth.length_norm(synth.sample_splda(m2, cnt, 1, seed)) for cnt, seed in ((ne, 1), (nt, 2), (nc, 3)))
p2 = SPLDA(mu=m2["mu"], V=m2["V"], W=m2["W"])
s_et, s_ct, s_ec = p2.llr_1vs1(xe, xt), p2.llr_1vs1(xc, xt), p2.llr_1vs1(xe, xc)
AdaptSNorm(nbest=nbest).predict(s_et[:8, :8], s_ct[:, :8], s_ec[:8])  # first touch
t_api, got = wall(lambda: AdaptSNorm(nbest=nbest).predict(s_et, s_ct, s_ec))
sub = slice(0, 200)
t_ref, ref = wall(lambda: O.adapt_s_norm(s_et[sub, :2000], s_ct[:, :2000], s_ec[sub], nbest, 0), 1)
err = np.abs(got[sub, :2000] - ref) / np.maximum(np.abs(ref), 1)
print("AdaptSNorm.predict(numpy float64) %d x %d, cohort %d, nbest %d: %8.1f ms = %.1f M scores/s   max error %.2e"
      % (ne, nt, nc, nbest, t_api, ne * nt / t_api / 1e3, err.max()))
print("  oracle on a 200 x 2000 slice: %8.1f ms = %.2f M scores/s" % (t_ref, 200 * 2000 / t_ref / 1e3))
dev = [engine.upload(a) for a in (s_et, s_ec, np.ascontiguousarray(s_ct.T))]
from hyperion_amd.score_norm.adapt_s_norm import adapt_s_norm_device  # noqa: E402
t_dev, _ = wall(lambda: adapt_s_norm_device(dev[0], dev[1], dev[2], nbest, 0))
print("  device tensors in/out (fp32):  %8.1f ms" % t_dev)

# ---- the other host-array entry points at evaluation-set sizes ---------------------------------------
from hyperion_amd.score_norm import SNorm  # noqa: E402
from hyperion_amd.transforms import CentWhiten, LDA, LNorm, TransformList  # noqa: E402

chain = synth.chain_cfg2(d=512, lda_dim=200)
(_, mu_cw, T_cw), _, (_, mu_lda, T_lda) = chain
pre = TransformList([CentWhiten(mu=mu_cw, T=T_cw), LNorm(), LDA(mu=mu_lda, T=T_lda)])
xr = synth.raw_xvectors(100000, d=512, seed=5)[0]
pre.predict(xr[:64])
t_api, y = wall(lambda: pre.predict(xr))
xd512 = engine.upload(xr)
t_dev, _ = wall(lambda: pre.predict(xd512))
print("TransformList.predict(numpy float64 100000 x 512 -> 200): %7.1f ms   device tensors %6.2f ms" % (t_api, t_dev))

ids = np.repeat(np.arange(5000), 3)
x_en = synth.length_norm(synth.sample_splda(m2, 5000, 3, 11))
x_te = synth.length_norm(synth.sample_splda(m2, 20000, 1, 12))
for method in ("vavg-lnorm", "savg", "book"):
    p2.llr_Nvs1(x_en[:30], x_te[:8], ids1=ids[:30], method=method) if method != "book" else None
    if method == "book":
        D1 = p2.compute_stats_hard(x_en, ids)
        D2 = p2.compute_stats_hard(x_te, np.arange(20000))
        t_api, _ = wall(lambda: p2.llr_NvsM_book(D1, D2))
    else:
        t_api, _ = wall(lambda: p2.llr_Nvs1(x_en, x_te, ids1=ids, method=method))
    print("llr_Nvs1 %-10s numpy 15000 segments of 5000 models x 20000: %7.1f ms" % (method, t_api))
t_api, _ = wall(lambda: p2.llr_1vs1(xe, xt))
print("llr_1vs1 numpy 2000 x 20000 (200-d):                         %7.1f ms" % t_api)
t_api, _ = wall(lambda: SNorm().predict(s_et, s_ct, s_ec))
print("SNorm.predict numpy 2000 x 20000, cohort 5000:                %7.1f ms" % t_api)
