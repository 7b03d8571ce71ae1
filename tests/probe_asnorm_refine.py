"""How tight is the fp32 error estimate that decides which adaptively normalised scores are recomputed
in float64 (hypmi_asnorm_refine)?  For a few score distributions: the fp32 device result against the
float64 oracle, the largest ratio error / estimate (estimate with coefficient 2^-24), the share of
entries the default coefficient would recompute, the largest strict relative error with and without
the refinement, and the time of both.

Usage: python tests/probe_asnorm_refine.py [ne nt nc nbest]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def cases(ne, nt, nc, rng):
    yield "normal scores, cohort mean -14", rng.randn(ne, nt) * 6 - 5, rng.randn(nc, nt) * 8 - 14, rng.randn(ne, nc) * 8 - 14
    yield "narrow cohort far from zero", rng.randn(ne, nt) + 60, rng.randn(nc, nt) * 0.5 + 60, rng.randn(ne, nc) * 0.5 + 60
    yield "heavy tails", rng.standard_t(3, (ne, nt)) * 4, rng.standard_t(3, (nc, nt)) * 4 - 3, rng.standard_t(3, (ne, nc)) * 4 - 3
    # scores shaped like PLDA log-likelihood ratios: a common factor per side
    a, b, c = rng.randn(ne, 1) * 3, rng.randn(1, nt) * 3, rng.randn(nc) * 3
    yield "low rank plus noise", a + b + rng.randn(ne, nt) * 5 - 20, c[:, None] + b + rng.randn(nc, nt) * 5 - 20, \
        a + c[None, :] + rng.randn(ne, nc) * 5 - 20


def main():
    import torch

    import plda_ref as O
    from hyperion_amd.score_norm import AdaptSNorm

    ne, nt, nc, nbest = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (200, 900, 4000, 300)
    rng = np.random.RandomState(0)
    for name, sc, ct, ec in cases(ne, nt, nc, rng):
        ref = O.adapt_s_norm(sc, ct, ec, nbest, 0)
        plain = AdaptSNorm(nbest=nbest, refine=False)
        fine = AdaptSNorm(nbest=nbest)
        times = []
        for norm in (plain, fine):
            norm.predict(sc, ct, ec)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = norm.predict(sc, ct, ec)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            if norm is plain:
                got_plain = got
        # the estimate, from float64 statistics (what the kernel computes in fp32)
        best_t = np.argsort(-ct, axis=0, kind="stable")[:nbest]  # [nbest, nt]
        best_e = np.argsort(-ec, axis=1, kind="stable")[:, :nbest]
        mz = np.stack([ec[:, best_t[:, i]].mean(1) for i in range(nt)], 1)
        sz = np.stack([ec[:, best_t[:, i]].std(1) for i in range(nt)], 1)
        mt = np.stack([ct[best_e[i]].mean(0) for i in range(ne)], 0)
        st = np.stack([ct[best_e[i]].std(0) for i in range(ne)], 0)
        z, t = (sc - mz) / sz, (sc - mt) / st
        est = 2.0 ** -24 * ((np.abs(sc) + np.abs(mz)) / sz + (np.abs(sc) + np.abs(mt)) / st + 8 + 16 * (np.abs(z) + np.abs(t)))
        err = np.abs(got_plain - ref)
        rel_plain = err / np.abs(ref)
        rel_fine = np.abs(got - ref) / np.abs(ref)
        print("%-30s err/estimate max %.3f  median %.4f | refined %.2f %% | strict relative error: fp32 max %.2e "
              "(%.2f %% > 1e-4), refined max %.2e | %.1f ms -> %.1f ms"
              % (name, (err / est).max(), np.median(err / est), 100.0 * fine.last_refined / ref.size, rel_plain.max(),
                 100.0 * (rel_plain > 1e-4).mean(), rel_fine.max(), 1e3 * times[0], 1e3 * times[1]), flush=True)


if __name__ == "__main__":
    main()
