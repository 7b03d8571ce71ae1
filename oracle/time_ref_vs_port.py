"""Times the imported reference and the numpy restatement (oracle/plda_ref.py) on the same slice of the
adaptive S-norm workload, to confirm that the restatement, which `bench.py` times as `cpu_baseline`, is
neither faster nor slower than the reference by construction.

TEST INFRASTRUCTURE, in-container only (it imports the reference through oracle/load_ref.py; nothing run
on the GPU box reads /root/reference).

    python oracle/time_ref_vs_port.py [n_enrol n_test]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import load_ref, plda_ref as O  # noqa: E402
import synth  # noqa: E402


def main():
    ne = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    nt = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
    nc, nbest, d, lda_dim, K = 20000, 300, 512, 200, 150
    R = load_ref.load()
    chain = synth.chain_cfg2(d=d, lda_dim=lda_dim)
    model = synth.make_splda(lda_dim, K)
    xe, xt, xc = (synth.raw_xvectors(n, d=d, seed=s)[0].astype(np.float64) for n, s in ((ne, 41), (nt, 43), (nc, 42)))

    t0 = time.perf_counter()
    ye, yt, yc = (O.chain_predict(x, chain) for x in (xe, xt, xc))
    port = O.adapt_s_norm(O.llr_1vs1(model, ye, yt), O.llr_1vs1(model, yc, yt), O.llr_1vs1(model, ye, yc), nbest, 0)
    t_port = time.perf_counter() - t0

    (_, mu_cw, T_cw), _, (_, mu_lda, T_lda) = chain
    pre = R.TransformList([R.CentWhiten(mu=mu_cw, T=T_cw), R.LNorm(), R.LDA(mu=mu_lda, T=T_lda)])
    plda = R.SPLDA(y_dim=K, mu=model["mu"], V=model["V"], W=model["W"])
    t0 = time.perf_counter()
    ye, yt, yc = (pre.predict(x) for x in (xe, xt, xc))
    ref = R.AdaptSNorm(nbest=nbest).predict(plda.llr_1vs1(ye, yt), plda.llr_1vs1(yc, yt), plda.llr_1vs1(ye, yc))
    t_ref = time.perf_counter() - t0

    err = float(np.max(np.abs(ref - port)))
    print("slice %d enrol x %d test, cohort %d, nbest %d, %d threads" % (ne, nt, nc, nbest, os.cpu_count()))
    print("reference   %.2f s = %.0f scores/s" % (t_ref, ne * nt / t_ref))
    print("restatement %.2f s = %.0f scores/s (x %.2f of the reference's time)" % (t_port, ne * nt / t_port, t_port / t_ref))
    print("max |reference - restatement| = %.3g" % err)


if __name__ == "__main__":
    main()
