"""SPLDA EM training on the device against the reference's results (tests/golden/splda_training.npz)
and against the float64 oracle on a larger labelled set."""
import numpy as np
import pytest

from oracle import plda_ref as O

pytestmark = pytest.mark.gpu

SETUPS = {
    "a": (dict(epochs=6, ml_md="ml+md"), {}),
    "c": (dict(epochs=5, ml_md="ml+md", md_epochs=[1, 3]), dict(update_mu=False)),
    "d": (dict(epochs=3, ml_md="md"), dict(update_W=False)),
}


def _same_up_to_row_signs(V, ref, tol):
    """The rows of V are defined up to sign (x = mu + y V + e with y ~ N(0, I))."""
    sign = np.sign(np.sum(V * ref, axis=1))
    return np.allclose(V * sign[:, None], ref, rtol=tol, atol=tol * np.abs(ref).max())


def test_first_model_and_posterior_match_reference(golden):
    from hyperion_amd.pdfs.plda import SPLDA

    g = golden("splda_training")
    x, y, w = g["x"], g["y"], g["w"]
    K = g["init_V"].shape[0]
    for tag, kw in (("", {}), ("w_", dict(sample_weight=w))):
        D = SPLDA.compute_stats_hard(x, y, **kw)
        m = SPLDA(y_dim=K)
        m.initialize(D)
        assert np.allclose(m.mu, g["init_" + tag + "mu"], rtol=1e-10, atol=1e-12)
        assert _same_up_to_row_signs(m.V, g["init_" + tag + "V"], 1e-8)
        assert np.allclose(m.W, g["init_" + tag + "W"], rtol=1e-8, atol=1e-10)
        # the posterior of the reference's own first model (sign conventions included)
        m = SPLDA(mu=g["init_" + tag + "mu"], V=g["init_" + tag + "V"], W=g["init_" + tag + "W"])
        yy, cov, logpy, Ry, Py = m.compute_py_g_x(D, return_cov=True, return_logpy_0=True, return_acc=True)
        for a, name in zip((yy, cov, logpy, Ry, Py), ("y", "cov", "logpy", "Ry", "Py")):
            ref = g["post_" + tag + name]
            assert np.allclose(a, ref, rtol=1e-9, atol=1e-10 * np.abs(ref).max()), (tag, name)
        assert np.array_equal(m.compute_py_g_x(D), yy)
        assert np.isclose(m.elbo(m.Estep(D)), g["elbo0_" + tag[:1]], rtol=1e-10)


@pytest.mark.parametrize("tag", sorted(SETUPS))
def test_fit_matches_reference(golden, tag):
    from hyperion_amd.pdfs.plda import SPLDA

    g = golden("splda_training")
    kw, upd = SETUPS[tag]
    m = SPLDA(y_dim=g["init_V"].shape[0], **upd)
    elbo, elbo_norm = m.fit(g["x"], class_ids=g["y"], **kw)
    assert np.allclose(elbo, g["fit_%s_elbo" % tag], rtol=1e-8)
    assert np.allclose(elbo_norm, g["fit_%s_elbo_norm" % tag], rtol=1e-8)
    assert np.allclose(m.mu, g["fit_%s_mu" % tag], rtol=1e-6, atol=1e-8)
    assert np.allclose(m.W, g["fit_%s_W" % tag], rtol=1e-6, atol=1e-8)
    assert _same_up_to_row_signs(m.V, g["fit_%s_V" % tag], 1e-6)


def test_estep_on_weighted_statistics(golden):
    """Counts that are not whole numbers (the reference then visits the classes one by one)."""
    from hyperion_amd.pdfs.plda import SPLDA

    g = golden("splda_training")
    D = SPLDA.compute_stats_hard(g["x"], g["y"], sample_weight=g["w"])
    m = SPLDA(mu=g["init_w_mu"], V=g["init_w_V"], W=g["init_w_W"])
    elbo = []
    for _ in range(4):
        st = m.Estep(D)
        elbo.append(m.elbo(st))
        m.MstepML(st)
    assert np.allclose(elbo, g["fit_b_elbo"], rtol=1e-8)
    assert np.allclose(m.W, g["fit_b_W"], rtol=1e-6, atol=1e-8)


def test_training_at_size_against_oracle():
    """20 000 vectors of 3000 classes in 200 dimensions, y_dim 100, with a validation set: ELBO
    sequence, trained model and the scores it gives."""
    from hyperion_amd.pdfs.plda import SPLDA

    rng = np.random.RandomState(5)
    d, K, ncls, n = 200, 100, 3000, 20000
    Vt = rng.randn(K, d) * np.geomspace(1.0, 0.1, K)[:, None]
    y = rng.randint(0, ncls, n)
    y[:ncls] = np.arange(ncls)
    x = rng.randn(ncls, K)[y] @ Vt + rng.randn(n, d) + 0.3
    yv = np.arange(500).repeat(3)
    xv = rng.randn(500, K)[yv] @ Vt + rng.randn(1500, d) + 0.3
    m = SPLDA(y_dim=K)
    elbo, elbo_norm, elbo_val, elbo_val_norm = m.fit(x, class_ids=y, x_val=xv, class_ids_val=yv, epochs=5)
    D = O.compute_stats_hard(x, y)
    ref, ref_elbo = O.splda_fit(D, K, epochs=5)
    assert np.allclose(elbo, ref_elbo, rtol=1e-8)
    assert np.allclose(elbo_norm, ref_elbo / n, rtol=1e-8)
    assert np.all(np.isfinite(elbo_val)) and np.allclose(elbo_val_norm, elbo_val / 1500)
    assert np.allclose(m.V.T @ m.V, ref["V"].T @ ref["V"], rtol=1e-6, atol=1e-7)
    assert np.allclose(m.W, ref["W"], rtol=1e-6, atol=1e-8)
    s = m.llr_1vs1(x[:50], xv[:60])
    assert np.allclose(s, O.llr_1vs1(ref, x[:50], xv[:60]), rtol=1e-4, atol=1e-4)


def test_soft_statistics_match_numpy():
    from hyperion_amd.pdfs.plda import SPLDA

    rng = np.random.RandomState(8)
    x = rng.randn(300, 17)
    p = rng.dirichlet(np.ones(9), 300)
    w = rng.uniform(0.5, 1.5, 300)
    N, F, S = SPLDA.compute_stats_soft(x, p, sample_weight=w)
    pw = p * w[:, None]
    assert np.allclose(N, pw.sum(0)) and np.allclose(F, pw.T @ x, rtol=1e-12, atol=1e-12)
    assert np.allclose(S, x.T @ (pw.sum(1, keepdims=True) * x), rtol=1e-12, atol=1e-12)
