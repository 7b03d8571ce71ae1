"""FRPLDA EM training on the device against the reference's results
(tests/golden/frplda_training.npz), the reference's own FRPLDA test ported to this package
(tests/hyperion/pdfs/plda/test_frplda.py), adaptation loops, and the float64 oracle at size."""
import numpy as np
import pytest

from oracle import plda_ref as O

pytestmark = pytest.mark.gpu

SETUPS = {
    "a": (dict(epochs=6, ml_md="ml+md"), {}, 0.0),
    "c": (dict(epochs=5, ml_md="ml"), dict(update_mu=False), 0.02),
    "d": (dict(epochs=3, ml_md="ml"), dict(update_B=False, fullcov_W=False), 0.0),
}


def test_first_model_and_posterior_match_reference(golden):
    from hyperion_amd.pdfs.plda import FRPLDA

    g = golden("frplda_training")
    x, y, w = g["x"], g["y"], g["w"]
    for tag, kw in (("", {}), ("w_", dict(sample_weight=w))):
        D = FRPLDA.compute_stats_hard(x, y, **kw)
        m = FRPLDA()
        m.initialize(D)
        assert m.is_init and m.x_dim == m.y_dim == x.shape[1]
        for p in ("mu", "B", "W"):
            ref = g["init_" + tag + p]
            assert np.allclose(getattr(m, p), ref, rtol=1e-8, atol=1e-10 * np.abs(ref).max()), p
        m = FRPLDA(mu=g["init_" + tag + "mu"], B=g["init_" + tag + "B"], W=g["init_" + tag + "W"])
        yy, cov, logpy, Ry, Py = m.compute_py_g_x(D, return_cov=True, return_logpy_0=True, return_acc=True)
        for a, name in zip((yy, cov, logpy, Ry, Py), ("y", "cov", "logpy", "Ry", "Py")):
            ref = g["post_" + tag + name]
            assert np.allclose(a, ref, rtol=1e-9, atol=1e-10 * max(np.abs(ref).max(), 1e-300)), (tag, name)
        assert np.array_equal(m.compute_py_g_x(D), yy)
        assert np.isclose(m.elbo(m.Estep(D)), g["elbo0_" + tag[:1]], rtol=1e-10)


@pytest.mark.parametrize("tag", sorted(SETUPS))
def test_fit_matches_reference(golden, tag):
    from hyperion_amd.pdfs.plda import FRPLDA

    g = golden("frplda_training")
    kw, upd, shift = SETUPS[tag]
    m = FRPLDA(**upd)
    if shift:  # a model whose mean is kept
        m.initialize(FRPLDA.compute_stats_hard(g["x"], g["y"]))
        m.mu = m.mu + shift
    elbo, elbo_norm = m.fit(g["x"], class_ids=g["y"], **kw)
    assert np.allclose(elbo, g["fit_%s_elbo" % tag], rtol=1e-8)
    assert np.allclose(elbo_norm, g["fit_%s_elbo_norm" % tag], rtol=1e-8)
    for p in ("mu", "B", "W"):
        ref = g["fit_%s_%s" % (tag, p)]
        assert np.allclose(getattr(m, p), ref, rtol=1e-6, atol=1e-8 * np.abs(ref).max()), p


def test_estep_on_weighted_statistics(golden):
    """Counts that are not whole numbers (the reference then visits the classes one by one)."""
    from hyperion_amd.pdfs.plda import FRPLDA

    g = golden("frplda_training")
    D = FRPLDA.compute_stats_hard(g["x"], g["y"], sample_weight=g["w"])
    m = FRPLDA(mu=g["init_w_mu"], B=g["init_w_B"], W=g["init_w_W"])
    elbo = []
    for _ in range(4):
        st = m.Estep(D)
        elbo.append(m.elbo(st))
        m.MstepML(st)
    assert np.allclose(elbo, g["fit_b_elbo"], rtol=1e-8)
    assert np.allclose(m.W, g["fit_b_W"], rtol=1e-6, atol=1e-8 * np.abs(g["fit_b_W"]).max())


def test_reference_frplda_test_file(golden):
    """The numeric content of the reference's tests/hyperion/pdfs/plda/test_frplda.py (its plots
    left out): sample, initialize, posterior, fit, and that targets outscore non-targets for
    1-vs-1, N-vs-M and N-vs-1 trials with every averaging method."""
    from scipy import linalg as la

    from hyperion_amd.pdfs.plda import FRPLDA

    g = golden("frplda_training")
    num_classes, num_spc = 50, 10
    mu = np.array([0.5, 0])
    Sb = np.array([[1, 0.05], [0.05, 0.1]])
    Sw = np.array([[0.1, 0.05], [0.05, 1]])
    plda1 = FRPLDA(mu=mu, B=la.inv(Sb), W=la.inv(Sw))
    x_train = plda1.sample(num_classes, num_spc, seed=1024)
    class_ids = np.repeat(np.arange(num_classes), num_spc)
    plda2 = FRPLDA()
    plda2.initialize(plda2.compute_stats_hard(x_train, class_ids))
    assert np.allclose(la.inv(plda2.B), Sb, atol=0.5) and np.allclose(plda2.mu, mu, atol=0.5)

    x, y = plda1.sample(3, 3, seed=1024, return_y=True)
    yp, Cy = plda1.compute_py_g_x(plda1.compute_stats_hard(x, np.repeat(np.arange(3), 3)), return_cov=True)
    ref = O.frplda_compute_py_g_x({"mu": mu, "B": plda1.B, "W": plda1.W}, O.compute_stats_hard(x, np.repeat(np.arange(3), 3)),
                                  return_cov=True)
    assert np.allclose(yp, ref[0], rtol=1e-10) and np.allclose(Cy, ref[1], rtol=1e-10)

    plda3 = FRPLDA()
    elbo, elbo_norm = plda3.fit(x_train, class_ids)
    assert np.allclose(elbo, g["t2_elbo"], rtol=1e-8) and np.allclose(elbo_norm, g["t2_elbo_norm"], rtol=1e-8)
    for p in ("mu", "B", "W"):
        assert np.allclose(getattr(plda3, p), g["t2_fit_" + p], rtol=1e-6, atol=1e-8), p
    assert np.all(np.diff(elbo) > -1e-6 * np.abs(elbo[0]))

    tar = np.eye(num_classes, dtype=bool)
    x = plda1.sample(num_classes, 2, seed=1024)
    s = plda1.llr_1vs1(x[::2], x[1::2])
    assert np.mean(s[tar]) > np.mean(s[~tar])
    x = plda1.sample(num_classes, 6, seed=1024)
    ids3 = np.repeat(np.arange(num_classes), 3)
    for method in ("book", "savg", "vavg", "vavg-lnorm"):
        s = plda1.llr_NvsM(x[::2], x[1::2], ids1=ids3, ids2=ids3, method=method)
        assert s.shape == (num_classes, num_classes) and np.mean(s[tar]) > np.mean(s[~tar]), method
    x = plda1.sample(num_classes, 4, seed=1024)
    mask = np.zeros((len(x),), dtype=bool)
    mask[::4] = True
    for method in ("book", "savg", "vavg", "vavg-lnorm"):
        s = plda1.llr_Nvs1(x[~mask], x[mask], ids1=ids3, method=method)
        assert s.shape == (num_classes, num_classes) and np.mean(s[tar]) > np.mean(s[~tar]), method


@pytest.mark.parametrize("kind", ["splda", "frplda"])
def test_adaptation_loops(kind):
    """fit_adapt_weighted_avg_model = E-step, M-step, pull towards the prior model, per epoch; the
    same sequence driven by hand with the oracle's pieces.  fit_adapt runs EM from the current model."""
    from hyperion_amd.pdfs.plda import FRPLDA, SPLDA

    rng = np.random.RandomState(3)
    d, K, ncls = 10, 4, 60
    ids = np.repeat(np.arange(ncls), 5)
    make = lambda shift, seed: (np.random.RandomState(seed).randn(ncls, d)[ids] * 1.2  # noqa: E731
                                + np.random.RandomState(seed + 1).randn(len(ids), d) * 0.8 + shift)
    x_out, x_in = make(0.0, 10), make(0.7, 20)
    new = (lambda: SPLDA(y_dim=K)) if kind == "splda" else (lambda: FRPLDA())
    prior = new()
    prior.fit(x_out, class_ids=ids, epochs=4)
    m = prior.copy()
    elbo, elbo_norm = m.fit_adapt_weighted_avg_model(x_in, class_ids=ids, epochs=3, plda0=prior, w_mu=1, w_B=0.5,
                                                     w_W=0.25)
    # by hand, on the oracle
    keys = ("mu", "V", "W") if kind == "splda" else ("mu", "B", "W")
    ref0 = dict({k: np.array(getattr(prior, k)) for k in keys}, type=kind)
    cur = dict(ref0)
    D = O.compute_stats_hard(x_in, ids)
    ref_elbo = []
    for _ in range(3):
        if kind == "splda":
            st = O.splda_estep(cur, D)
            ref_elbo.append(O.splda_elbo(cur, st))
            cur = O.splda_mstep_md(O.splda_mstep_ml(cur, st), st)
        else:
            st = O.frplda_estep(cur, D)
            ref_elbo.append(O.frplda_elbo(cur, st))
            cur = O.frplda_mstep_ml(cur, st)
        cur = O.plda_weighted_avg_model(cur, ref0, 1, 0.5, 0.25)
    assert np.allclose(elbo, ref_elbo, rtol=1e-8) and np.allclose(elbo_norm, np.array(ref_elbo) / len(ids), rtol=1e-8)
    assert np.allclose(m.mu, prior.mu)  # w_mu = 1 puts the prior's mean back
    for k in keys[1:]:
        got, ref = getattr(m, k), cur[k]
        if k == "V":
            got, ref = got.T @ got, ref.T @ ref
        assert np.allclose(got, ref, rtol=1e-6, atol=1e-8 * np.abs(ref).max()), k

    m2 = prior.copy()
    e2, _ = m2.fit_adapt(x_in, class_ids=ids, x0=x_out, class_ids0=ids, epochs=3)
    m3 = prior.copy()
    e3, _ = m3.fit(x_in, class_ids=ids, epochs=3)  # an initialised model: fit continues from it
    assert np.allclose(e2, e3, rtol=1e-12) and np.allclose(m2.W, m3.W, rtol=1e-10)
    with pytest.raises(AssertionError):
        prior.copy().fit_adapt(x_in, class_ids=ids)


def test_training_at_size_against_oracle():
    """20 000 vectors of 3000 classes in 150 dimensions with ragged counts and a validation set."""
    from hyperion_amd.pdfs.plda import FRPLDA

    rng = np.random.RandomState(6)
    d, ncls, n = 150, 3000, 20000
    scale = np.geomspace(1.5, 0.2, d)
    y = rng.randint(0, ncls, n)
    y[:ncls] = np.arange(ncls)
    x = (rng.randn(ncls, d) * scale)[y] + rng.randn(n, d) + 0.3
    yv = np.arange(500).repeat(3)
    xv = (rng.randn(500, d) * scale)[yv] + rng.randn(1500, d) + 0.3
    m = FRPLDA()
    elbo, elbo_norm, elbo_val, elbo_val_norm = m.fit(x, class_ids=y, x_val=xv, class_ids_val=yv, epochs=5)
    ref, ref_elbo = O.frplda_fit(O.compute_stats_hard(x, y), epochs=5)
    assert np.allclose(elbo, ref_elbo, rtol=1e-8) and np.allclose(elbo_norm, ref_elbo / n, rtol=1e-8)
    assert np.all(np.isfinite(elbo_val)) and np.allclose(elbo_val_norm, elbo_val / 1500)
    for p in ("mu", "B", "W"):
        assert np.allclose(getattr(m, p), ref[p], rtol=1e-6, atol=1e-8 * np.abs(ref[p]).max()), p
    s = m.llr_1vs1(x[:50], xv[:60])
    assert np.allclose(s, O.llr_1vs1(ref, x[:50], xv[:60]), rtol=1e-4, atol=1e-4)
