"""Host-side (float64) folding logic of hyperion_amd/fold.py, checked against the oracle by
evaluating the closed forms with numpy: this covers the algebra the device kernels rely on
(S = g1.g2 + q1 + q2 + c, chain composition, eigen-basis book scoring) without a GPU."""
import numpy as np
import pytest

from conftest import model_from
from hyperion_amd import fold
from oracle import plda_ref as O


def folded(m):
    if m["type"] == "splda":
        return fold.fold_from_WV(m["mu"], *fold.splda_WV(m["V"], m["W"]))
    if m["type"] == "plda":
        return fold.fold_from_WV(m["mu"], *fold.plda_WV(m["V"], m["U"], m["D"]))
    return fold.fold_frplda(m["mu"], m["B"], m["W"])


def closed_form(f, x1, x2):
    At, bt, An, bn, c = f
    g1, g2 = x1 @ At + bt, x2 @ At + bt
    h1, h2 = x1 @ An + bn, x2 @ An + bn
    q1 = 0.5 * (np.sum(g1 * g1, 1) - np.sum(h1 * h1, 1))
    q2 = 0.5 * (np.sum(g2 * g2, 1) - np.sum(h2 * h2, 1))
    return g1 @ g2.T + q1[:, None] + q2[None, :] + c


@pytest.mark.parametrize("kind", ["splda", "frplda", "plda"])
def test_closed_form_matches_golden(golden, kind):
    g = golden("llr_1vs1")
    for tag in "ab":
        m = model_from(g, "%s_%s_" % (kind, tag), kind)
        s = closed_form(folded(m), g["x1_" + tag], g["x2_" + tag])
        ref = g["%s_%s_scores" % (kind, tag)]
        assert np.max(np.abs(s - ref)) <= 1e-10 * max(1.0, np.max(np.abs(ref)))


def apply_plan(segments, tail, x):
    for A, b, d in segments:
        if A is not None:
            x = x @ A
        if b is not None:
            x = x + b
        x = np.sqrt(x.shape[1]) * x / (np.sqrt(np.sum(x ** 2, axis=1, keepdims=True)) + 1e-10)
    A, b = tail
    if A is not None:
        x = x @ A
    if b is not None:
        x = x + b
    return x


def test_chain_folding(golden):
    g = golden("transforms")
    x = g["x"]
    aff = fold.affine_from_mu_T
    cases = [
        ([aff(g["mu3"], g["T3"]) + (False,), (None, None, True)], "chain_lda_lnorm"),
        ([aff(g["mu1"], g["T1"]) + (False,), (None, None, True), aff(g["mu2"], g["T2"]) + (False,)],
         "chain_cw_lnorm_lda"),
        ([aff(g["mu1"], g["T1"]) + (False,), aff(g["mu2"], None) + (True,),
          aff(g["mu2"], g["T2"]) + (False,), (None, None, True)], "chain_two_norms"),
        ([aff(g["mu1"], g["T1"]) + (True,)], "lnorm"),
        ([aff(g["mu1"], None) + (False,)], "cw_noT"),
    ]
    for steps, key in cases:
        segs, tail, d_out = fold.fold_chain(steps, x.shape[1])
        y = apply_plan(segs, tail, x)
        assert y.shape[1] == d_out
        assert np.max(np.abs(y - g[key])) <= 1e-10 * max(1.0, np.max(np.abs(g[key])))


def test_diagonal_T():
    rng = np.random.RandomState(0)
    x, mu, T = rng.randn(5, 6), rng.randn(6), rng.rand(6) + 0.5
    A, b = fold.affine_from_mu_T(mu, T)
    assert np.allclose(x @ A + b, (x - mu) * T)


def book_scores(bf, N1, F1, N2, F2):
    """numpy model of hypmi_affine + hypmi_book_pack + hypmi_score for by-the-book scoring."""
    lam = bf["lam"]
    s1 = F1 @ bf["P"] + N1[:, None] * bf["row_bias"]
    s2 = F2 @ bf["P"] + N2[:, None] * bf["row_bias"]
    out = np.zeros((len(N1), len(N2)))
    logL = lambda n: np.sum(np.log1p(n * lam))  # noqa: E731
    for a in np.unique(N1):
        for b in np.unique(N2):
            i, j = np.where(N1 == a)[0], np.where(N2 == b)[0]
            st = 1 / np.sqrt(1 + (a + b) * lam)
            g1 = (s1[i] + bf["shift_t"]) * st
            g2 = (s2[j] + bf["shift_t"]) * st
            h1 = (s1[i] + bf["shift_n"]) / np.sqrt(1 + a * lam)
            h2 = (s2[j] + bf["shift_n"]) / np.sqrt(1 + b * lam)
            q1 = 0.5 * (np.sum(g1 ** 2, 1) - np.sum(h1 ** 2, 1))
            q2 = 0.5 * (np.sum(g2 ** 2, 1) - np.sum(h2 ** 2, 1))
            c = 0.5 * (logL(a) + logL(b) - logL(a + b)) + bf["c_extra"]
            out[np.ix_(i, j)] = g1 @ g2.T + q1[:, None] + q2[None, :] + c
    return out


@pytest.mark.parametrize("kind", ["splda", "frplda", "plda"])
def test_book_eigen_basis(golden, kind):
    g = golden("llr_NvsM")
    m = model_from(g, kind + "_", kind)
    if kind == "frplda":
        P, beta, lam, _, muBmu = fold.book_basis_frplda(m["mu"], m["B"], m["W"])
        bf = {"P": P, "row_bias": np.zeros(len(lam)), "shift_t": 0.5 * beta, "shift_n": beta,
              "lam": lam, "c_extra": 0.5 * muBmu}
    else:
        WV, VV = fold.splda_WV(m["V"], m["W"]) if kind == "splda" else fold.plda_WV(m["V"], m["U"], m["D"])
        P, lam = fold.book_basis_WV(WV, VV)
        bf = {"P": P, "row_bias": -(m["mu"] @ P), "shift_t": 0 * lam, "shift_n": 0 * lam, "lam": lam,
              "c_extra": 0.0}
    N1, F1, _ = O.compute_stats_hard(g["x1"], g["ids1"])
    N2, F2, _ = O.compute_stats_hard(g["x2"], g["ids2"])
    ref = g["%s_NvsM_book" % kind]
    s = book_scores(bf, N1, F1, N2, F2)
    assert np.max(np.abs(s - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("kind", ["splda", "frplda", "plda"])
def test_savg_is_bilinear(golden, kind):
    """Score averaging == scoring class means of the packed rows [g, q, 1] / [g, 1, q + c]."""
    g = golden("llr_NvsM")
    m = model_from(g, kind + "_", kind)
    At, bt, An, bn, c = folded(m)

    def packed(x, test_side):
        gg, hh = x @ At + bt, x @ An + bn
        q = 0.5 * (np.sum(gg ** 2, 1) - np.sum(hh ** 2, 1))
        one = np.ones_like(q)
        cols = [gg, one[:, None], (q + c)[:, None]] if test_side else [gg, q[:, None], one[:, None]]
        return np.hstack(cols)

    def mean_by(G, ids):
        return np.stack([G[ids == k].mean(0) for k in range(ids.max() + 1)])

    G1 = mean_by(packed(g["x1"], False), g["ids1"])
    G2 = mean_by(packed(g["x2"], True), g["ids2"])
    ref = g["%s_NvsM_savg" % kind]
    assert np.max(np.abs(G1 @ G2.T - ref)) <= 1e-10 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("kind", ["splda", "frplda", "plda"])
def test_shared_basis_projection_gives_the_same_scores(golden, kind):
    """The K-column form the device projects with: g and h as per-column maps of one projection."""
    from hyperion_amd.pdfs.plda import FRPLDA, PLDA, SPLDA

    g = golden("llr_1vs1")
    for tag in "ab":
        m = model_from(g, "%s_%s_" % (kind, tag), kind)
        model = {"splda": SPLDA, "frplda": FRPLDA, "plda": PLDA}[kind](**{k: v for k, v in m.items() if k != "type"})
        sh = model._shared_fold()
        c = model._fold()[4]

        def rows(x):
            s = x @ sh["P"] + sh["pb"]
            gg = (s + sh["shift_t"]) * sh["scale_t"]
            hh = (s + sh["shift_n"]) * sh["scale_n"]
            return gg, 0.5 * (np.sum(gg * gg, 1) - np.sum(hh * hh, 1))

        (g1, q1), (g2, q2) = rows(g["x1_" + tag]), rows(g["x2_" + tag])
        s = g1 @ g2.T + q1[:, None] + q2[None, :] + c
        ref = g["%s_%s_scores" % (kind, tag)]
        assert sh["P"].shape[1] == model._fold()[0].shape[1]
        assert np.max(np.abs(s - ref)) <= 1e-10 * max(1.0, np.max(np.abs(ref)))
