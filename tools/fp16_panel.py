#!/usr/bin/env python
"""What a 16-bit panel would cost the cohort-statistics gather in accuracy (host emulation, numpy only).

The rows of both cohort panels are shifted by their row mean and stored as fp32, fp16, bf16 or 16-bit fixed point with a scale per row; the moments of
every selection are then accumulated exactly, so that the only error is the storage format's.  Printed: the
error of the normalised scores against the float64 oracle on a 300 x 600 slice of the default workload
(chain + SPLDA, 20 000-member cohort, nbest 300), in the units of the 1e-4 parity bar.

    python tools/fp16_panel.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth
from oracle import plda_ref as O
ne, nt, nc, k = 300, 600, 20000, 300
chain = synth.chain_cfg2(d=512, lda_dim=200); model = synth.make_splda(200, 150)
xe, xt, xc = (synth.raw_xvectors(n, d=512, seed=s)[0] for n, s in ((ne, 41), (nt, 43), (nc, 42)))
# (raw_xvectors: class = row % 10000, so test i and enrol i share a class centre: the diagonal holds targets)
ye, yt, yc = (O.chain_predict(x, chain) for x in (xe, xt, xc))
S, CT, EC = O.llr_1vs1(model, ye, yt), O.llr_1vs1(model, yc, yt), O.llr_1vs1(model, ye, yc)
ref = O.adapt_s_norm(S, CT, EC, k, 0)
def part(rows, sel_by, store):
    # rows [n, nc]; lists from sel_by [m, nc] top-k; returns mean/std [n, m] from the stored panel
    idx = np.argsort(-sel_by, axis=1, kind="stable")[:, :k]            # [m, k]
    piv = rows.mean(axis=1, keepdims=True)
    p = store(rows - piv)
    g = p[:, idx]                                                        # [n, m, k]
    g = g.astype(np.float64)
    mu = g.mean(axis=2); var = (g * g).mean(axis=2) - mu * mu
    return piv + mu, np.sqrt(np.maximum(var, 0))
def int16_rows(a):
    """Fixed point with one scale per row: the largest magnitude of the row maps to 32767."""
    scale = np.abs(a).max(axis=1, keepdims=True) / 32767.0
    return np.rint(a / scale) * scale


for name, store in (("fp32", lambda a: a.astype(np.float32)), ("fp16", lambda a: a.astype(np.float16)), ("int16 per-row scale", int16_rows),
                    ("bf16", lambda a: (a.astype(np.float32).view(np.uint32) + 0x8000 & 0xffff0000).view(np.float32))):
    mz, sz = part(EC, CT.T, store)           # [ne, nt]
    mt, st = part(CT.T, EC, store)           # [nt, ne]
    got = ((S - mz) / np.maximum(sz, 1e-5) + (S - mt.T) / np.maximum(st.T, 1e-5)) / np.sqrt(2)
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1)
    print("%s panel (row-mean pivot, exact accumulation): max error %.2e, 99.9 %% quantile %.2e, share above 1e-4: %.2e, above 2e-5: %.2e"
          % (name, err.max(), np.quantile(err, 0.999), (err > 1e-4).mean(), (err > 2e-5).mean()))
