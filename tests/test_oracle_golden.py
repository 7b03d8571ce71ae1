"""The CPU oracle (oracle/plda_ref.py) against the fixtures produced from the reference itself
(tests/golden/*.npz, made by oracle/gen_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from conftest import model_from
from oracle import plda_ref as O

TOL = 1e-10


def close(a, b, tol=TOL):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape
    assert np.array_equal(np.isnan(a), np.isnan(b))
    scale = max(1.0, float(np.nanmax(np.abs(b)))) if b.size else 1.0
    err = float(np.nanmax(np.abs(a - b))) if b.size else 0.0
    assert err <= tol * scale, (err, scale)


@pytest.mark.parametrize("kind", ["splda", "frplda", "plda"])
@pytest.mark.parametrize("tag", ["a", "b"])
def test_llr_1vs1(golden, kind, tag):
    g = golden("llr_1vs1")
    m = model_from(g, "%s_%s_" % (kind, tag), kind)
    close(O.llr_1vs1(m, g["x1_" + tag], g["x2_" + tag]), g["%s_%s_scores" % (kind, tag)])


def test_known_answers():
    """Values the reference printed for seeded toy models in an earlier run of it."""
    rng = np.random.RandomState(0)
    d, yd = 16, 4
    V = rng.randn(yd, d)
    A = rng.randn(d, d)
    W = A @ A.T / d + np.eye(d)
    m = {"type": "splda", "mu": rng.randn(d), "V": V, "W": W}
    # SPLDA.sample(5, 2, seed=1): noise first, then speaker factors
    import scipy.linalg as sla
    r1 = np.random.RandomState(1)
    z = r1.normal(size=(10, d)) @ sla.cholesky(O.inv_pdmat(W), lower=False)
    y = np.repeat(r1.normal(size=(5, yd)) @ V + m["mu"], 2, axis=0)
    x = y + z
    s = O.llr_1vs1(m, x[::2], x[1::2])
    assert np.allclose(s[0, :3], [2.97965075, -7.73765166, -77.20524113], atol=1e-7)
    fr = {"type": "frplda", "mu": rng.randn(d), "B": W, "W": 2 * W}
    assert np.allclose(O.llr_1vs1(fr, x[::2], x[1::2])[0, :2], [28.60123297, 1.69248982], atol=1e-7)
    pl = {"type": "plda", "mu": rng.randn(d), "V": V, "U": rng.randn(3, d), "D": np.abs(rng.randn(d)) + 1}
    assert np.allclose(O.llr_1vs1(pl, x[::2], x[1::2])[0, :2], [2.50270301, -8.85197729], atol=1e-7)
    sc, ct, ec = rng.randn(6, 7), rng.randn(30, 7), rng.randn(6, 30)
    assert np.allclose(O.adapt_s_norm(sc, ct, ec, nbest=10)[0, :3],
                       [1.04342288, 0.54618728, -1.02196569], atol=1e-7)


def test_transforms(golden):
    g = golden("transforms")
    x = g["x"]
    close(O.cent_whiten_predict(x, g["mu1"], g["T1"]), g["cw"])
    close(O.cent_whiten_predict(x, g["mu1"], None), g["cw_noT"])
    close(O.cent_whiten_predict(x, None, g["T1"]), g["cw_nomu"])
    close(O.lnorm_predict(x, g["mu1"], g["T1"]), g["lnorm"])
    close(O.lnorm_predict(x), g["lnorm_plain"])
    close(O.lda_predict(x, g["mu3"], g["T3"]), g["lda"])
    close(O.lda_predict(x, g["mu3"], g["T3"]), g["pca"])
    close(O.lda_predict(x, None, g["T3"]), g["pca_nomu"])
    close(O.chain_predict(x, [("lda", g["mu3"], g["T3"]), ("lnorm", None, None)]), g["chain_lda_lnorm"])
    close(O.chain_predict(x, [("cw", g["mu1"], g["T1"]), ("lnorm", None, None),
                              ("lda", g["mu2"], g["T2"])]), g["chain_cw_lnorm_lda"])
    close(O.chain_predict(x, [("pca", g["mu1"], g["T1"]), ("lnorm", g["mu2"], None),
                              ("lda", g["mu2"], g["T2"]), ("lnorm", None, None)]), g["chain_two_norms"])
    close(O.lnorm_predict(g["x_zero_row"]), g["lnorm_zero_row"])


@pytest.mark.parametrize("kind", ["splda", "frplda", "plda"])
@pytest.mark.parametrize("method", ["book", "vavg", "vavg-lnorm", "savg"])
def test_nvsm(golden, kind, method):
    g = golden("llr_NvsM")
    m = model_from(g, kind + "_", kind)
    key = method.replace("-", "_")
    close(O.llr_NvsM(m, g["x1"], g["x2"], g["ids1"], g["ids2"], method), g["%s_NvsM_%s" % (kind, key)])
    close(O.llr_Nvs1(m, g["x1"], g["x2"], g["ids1"], method), g["%s_Nvs1_%s" % (kind, key)])


def test_stats_hard(golden):
    g = golden("llr_NvsM")
    N, F, _ = O.compute_stats_hard(g["stats_x"], g["stats_ids"])
    close(N, g["stats_N"])
    close(F, g["stats_F"])
    assert N[2] == 0  # class id 2 never occurs


def test_score_norm(golden):
    g = golden("score_norm")
    sc, ct, ec = g["scores"], g["coh_test"], g["enr_coh"]
    close(O.adapt_s_norm(sc, ct, ec, 10, 0), g["asnorm_n10"])
    close(O.adapt_s_norm(sc, ct, ec, 20, 3), g["asnorm_n20d3"])
    close(O.adapt_s_norm(sc, ct, ec, 100, 0), g["asnorm_clamp"])
    close(O.adapt_s_norm(sc, ct, ec, 100, 60), g["asnorm_clampd"])
    close(O.adapt_s_norm(sc, ct, ec, 10, 0, g["mask_coh_test"], g["mask_enr_coh"]), g["asnorm_masked"])
    close(O.s_norm(sc, ct, ec), g["snorm"])
    close(O.s_norm(sc, ct, ec, g["mask_coh_test"], g["mask_enr_coh"]), g["snorm_masked"])
    close(O.t_norm(sc, ct), g["tnorm"])
    close(O.z_norm(sc, ec), g["znorm"])
    close(O.z_norm(sc, g["enr_coh_const"]), g["znorm_floor"])
    close(O.adapt_s_norm(sc, ct, g["enr_coh_const"], 10, 0), g["asnorm_floor"])


def test_metrics_and_math(golden):
    g = golden("metrics_math")
    # the reference's own unit-test inputs and bounds (tests/hyperion/metrics/test_eer.py:14-29,
    # test_dcf.py:14-44)
    tar = np.linspace(-2, 10, 1000) + 3
    non = np.linspace(-10, 2, 1000) + 3
    eer = O.compute_eer(tar, non)
    assert 0.166 < eer < 0.167
    close(eer, g["lin_eer"])
    dcf = O.compute_min_dcf(tar, non, 0.5)[0]
    assert 0.332 < dcf < 0.334
    close(dcf, g["lin_min_dcf"])
    close(O.compute_eer(g["tar"], g["non"]), g["eer"])
    close(O.compute_min_dcf(g["tar"], g["non"], 0.05)[0], g["min_dcf_005"])
    close(O.compute_min_dcf(g["tar"], g["non"], 0.01)[0], g["min_dcf_001"])
    close(O.logdet_pdmat(g["pd_A"]), g["pd_logdet"])
    close(O.inv_pdmat(g["pd_A"]), g["pd_inv"])
    close(O.right_solve_upper(g["tri_R"], g["pd_X"]), g["tri_XiR"])
    # same pins as the reference's tests/hyperion/utils/test_math.py: against scipy inv / det
    assert np.allclose(O.inv_pdmat(g["pd_A"]), np.linalg.inv(g["pd_A"]))
    assert np.allclose(O.logdet_pdmat(g["pd_A"]), np.log(np.linalg.det(g["pd_A"])))


def test_medium(golden):
    g = golden("splda_medium")
    m = model_from(g, "m_", "splda")
    s = O.llr_1vs1(m, g["x1"].astype(float), g["x2"].astype(float))
    close(s, g["scores"])
    # the calibrated synthetic model has scores on both sides of zero
    assert (s > 0).any() and (s < 0).any()


def test_splda_project_matches_reference(golden):
    """oracle and product SPLDA.project against the reference's projected model (host arithmetic)."""
    from hyperion_amd.pdfs.plda import SPLDA

    g = golden("splda_project")
    mu, V, W = O.splda_project(g["m_mu"], g["m_V"], g["m_W"], g["T"], g["delta_mu"])
    for got, want in ((mu, g["p_mu"]), (V, g["p_V"]), (W, g["p_W"])):
        assert np.allclose(got, want, rtol=1e-10, atol=1e-12)
    model = SPLDA(mu=g["m_mu"].copy(), V=g["m_V"], W=g["m_W"])
    p = model.project(g["T"], g["delta_mu"])
    assert np.allclose(p.mu, g["p_mu"], rtol=1e-10, atol=1e-12) and np.allclose(p.V, g["p_V"], rtol=1e-10)
    assert np.allclose(p.W, g["p_W"], rtol=1e-9, atol=1e-12)
    assert np.array_equal(model.mu, g["m_mu"])  # the model itself is not modified
    s = O.llr_1vs1({"type": "splda", "mu": mu, "V": V, "W": W}, g["x"], g["x"])
    assert np.allclose(s, g["scores"], rtol=1e-9, atol=1e-9)


def test_training_statistics_restatement(golden):
    """Whitening / LDA fits of the oracle against the reference's (fixtures made by
    oracle/gen_golden.py gen_training_stats)."""
    g = golden("training_stats")
    mu, T = O.cent_whiten_fit(g["x"])
    assert np.allclose(mu, g["cw_mu"], rtol=1e-12) and np.allclose(T, g["cw_T"], rtol=1e-8, atol=1e-10)
    mu, T = O.cent_whiten_fit(g["x"], g["w"])
    assert np.allclose(mu, g["cww_mu"], rtol=1e-12) and np.allclose(T, g["cww_T"], rtol=1e-8, atol=1e-10)
    mu, Sb, Sw = O.sb_sw(g["x"], g["y"])
    assert np.allclose(mu, g["sb_mu"]) and np.allclose(Sb, g["sb_Sb"]) and np.allclose(Sw, g["sb_Sw"])
    mu, T = O.lda_fit(g["x"], g["y"], 6)
    assert np.allclose(mu, g["lda_mu"]) and np.allclose(T, g["lda_T"], rtol=1e-8, atol=1e-10)
    for tag, kw in (("pca3_", dict(pca_dim=3)), ("pcar_", dict(pca_var_r=0.8, whiten=True))):
        mu, T = O.pca_fit(g["x"], **kw)
        assert T.shape == g[tag + "T"].shape and np.allclose(T, g[tag + "T"], rtol=1e-8, atol=1e-10)


_SPLDA_SETUPS = {
    "a": (False, dict(epochs=6, ml_md="ml+md"), {}),
    "b": (True, dict(epochs=4, ml_md="ml"), {}),
    "c": (False, dict(epochs=5, ml_md="ml+md", md_epochs=[1, 3]), dict(update_mu=False)),
    "d": (False, dict(epochs=3, ml_md="md"), dict(update_W=False)),
}


def _weighted_stats(x, y, w):
    ncls = int(y.max()) + 1
    N = np.bincount(y, weights=w, minlength=ncls)
    F = np.zeros((ncls, x.shape[1]))
    np.add.at(F, y, x * w[:, None])
    return N, F, (x * w[:, None]).T @ x


def test_splda_training_restatement(golden):
    """SPLDA EM of the oracle against the reference's first model, posterior, ELBO sequences and
    trained parameters (fixture made by oracle/gen_golden.py gen_splda_training)."""
    g = golden("splda_training")
    x, y, w = g["x"], g["y"], g["w"]
    K = g["init_V"].shape[0]
    for tag, D in (("", O.compute_stats_hard(x, y)), ("w_", _weighted_stats(x, y, w))):
        mu, V, W = O.splda_initialize(D, K)
        assert np.allclose(mu, g["init_" + tag + "mu"], rtol=1e-10, atol=1e-12)
        assert np.allclose(V, g["init_" + tag + "V"], rtol=1e-9, atol=1e-11)
        assert np.allclose(W, g["init_" + tag + "W"], rtol=1e-9, atol=1e-11)
        m0 = {"type": "splda", "mu": g["init_" + tag + "mu"], "V": g["init_" + tag + "V"],
              "W": g["init_" + tag + "W"]}
        got = O.splda_compute_py_g_x(m0, D, return_cov=True, return_logpy_0=True, return_acc=True)
        for a, name in zip(got, ("y", "cov", "logpy", "Ry", "Py")):
            assert np.allclose(a, g["post_" + tag + name], rtol=1e-9, atol=1e-11), name
        assert np.isclose(O.splda_elbo(m0, O.splda_estep(m0, D)), g["elbo0_" + tag[:1]], rtol=1e-11)
    for tag, (weighted, kw, upd) in _SPLDA_SETUPS.items():
        D = _weighted_stats(x, y, w) if weighted else O.compute_stats_hard(x, y)
        m, elbo = O.splda_fit(D, K, **kw, **upd)
        assert np.allclose(elbo, g["fit_%s_elbo" % tag], rtol=1e-9)
        for p in ("mu", "V", "W"):
            assert np.allclose(m[p], g["fit_%s_%s" % (tag, p)], rtol=1e-7, atol=1e-9), (tag, p)


def test_torch_layer_formula_agrees_with_numpy_model(golden):
    """The trainable SPLDA layer of the reference states the same score with W = U U^T
    (hyperion/torch/models/plda/splda.py:182-222): an independent statement of the closed form, checked
    against the reference's numpy scores of the 1-vs-1 fixture."""
    g = golden("llr_1vs1")
    for case in ("a", "b"):
        m = model_from(g, "splda_%s_" % case, "splda")
        x1, x2 = g["x1_" + case], g["x2_" + case]
        ev, E = np.linalg.eigh(m["W"])
        U = E * np.sqrt(ev)
        s = O.splda_torch_layer_llr(x1, x2, m["mu"], m["V"], U)
        assert np.allclose(s, g["splda_%s_scores" % case], rtol=1e-9, atol=1e-9)
        self_scores = O.splda_torch_layer_llr(x1, None, m["mu"], m["V"], U)
        assert np.allclose(self_scores, O.llr_1vs1(m, x1, x1), rtol=1e-9, atol=1e-9)
        assert np.allclose(self_scores, self_scores.T, rtol=1e-12, atol=1e-12)


_FRPLDA_SETUPS = {
    "a": (False, dict(epochs=6, ml_md="ml+md"), {}, 0.0),
    "b": (True, dict(epochs=4, ml_md="ml"), {}, 0.0),
    "c": (False, dict(epochs=5, ml_md="ml"), dict(update_mu=False), 0.02),
    "d": (False, dict(epochs=3, ml_md="ml"), dict(update_B=False, fullcov_W=False), 0.0),
}


def test_frplda_training_restatement(golden):
    """FRPLDA EM, sampler and model interpolation of the oracle against the reference's results
    (fixture made by oracle/gen_golden.py gen_frplda_training), including the set-up of the
    reference's own test (tests/hyperion/pdfs/plda/test_frplda.py:22-29, 137-145)."""
    g = golden("frplda_training")
    truth = {"type": "frplda", "mu": g["t2_mu"], "B": g["t2_B"], "W": g["t2_W"]}
    x2, y2 = O.plda_sample(truth, 50, 10, seed=1024, return_y=True)
    assert np.allclose(x2, g["t2_x"], rtol=1e-12, atol=1e-12) and np.allclose(y2, g["t2_y"], rtol=1e-12, atol=1e-12)
    m, elbo = O.frplda_fit(O.compute_stats_hard(g["t2_x"], np.repeat(np.arange(50), 10)))
    assert np.allclose(elbo, g["t2_elbo"], rtol=1e-9)
    for p in ("mu", "B", "W"):
        assert np.allclose(m[p], g["t2_fit_" + p], rtol=1e-7, atol=1e-9), p

    x, y, w = g["x"], g["y"], g["w"]
    for tag, D in (("", O.compute_stats_hard(x, y)), ("w_", _weighted_stats(x, y, w))):
        for a, name in zip(O.frplda_initialize(D), ("mu", "B", "W")):
            assert np.allclose(a, g["init_" + tag + name], rtol=1e-9, atol=1e-11), name
        m0 = {"type": "frplda", "mu": g["init_" + tag + "mu"], "B": g["init_" + tag + "B"], "W": g["init_" + tag + "W"]}
        got = O.frplda_compute_py_g_x(m0, D, return_cov=True, return_logpy_0=True, return_acc=True)
        for a, name in zip(got, ("y", "cov", "logpy", "Ry", "Py")):
            assert np.allclose(a, g["post_" + tag + name], rtol=1e-9, atol=1e-11), name
        assert np.isclose(O.frplda_elbo(m0, O.frplda_estep(m0, D)), g["elbo0_" + tag[:1]], rtol=1e-11)
    for tag, (weighted, kw, upd, shift) in _FRPLDA_SETUPS.items():
        D = _weighted_stats(x, y, w) if weighted else O.compute_stats_hard(x, y)
        start = None
        if shift:
            mu0, B0, W0 = O.frplda_initialize(D)
            start = {"type": "frplda", "mu": mu0 + shift, "B": B0, "W": W0}
        m, elbo = O.frplda_fit(D, m=start, **kw, **upd)
        assert np.allclose(elbo, g["fit_%s_elbo" % tag], rtol=1e-9)
        for p in ("mu", "B", "W"):
            assert np.allclose(m[p], g["fit_%s_%s" % (tag, p)], rtol=1e-7, atol=1e-9), (tag, p)

    for kind, keys in (("splda", "mu V W"), ("frplda", "mu B W"), ("plda", "mu V U D")):
        a = dict({k: g["adapt_%s_a_%s" % (kind, k)] for k in keys.split()}, type=kind)
        b = dict({k: g["adapt_%s_b_%s" % (kind, k)] for k in keys.split()}, type=kind)
        assert np.allclose(O.plda_sample(a, 4, 3, seed=11), g["sample_" + kind], rtol=1e-12, atol=1e-12)
        out = O.plda_weighted_avg_model(a, b, 0.75, 0.5, 0.25)
        for k in keys.split():
            got, ref = out[k], g["adapt_%s_out_%s" % (kind, k)]
            if k in ("V", "U"):  # eigenvector rows: defined up to sign
                got, ref = got.T @ got, ref.T @ ref
            assert np.allclose(got, ref, rtol=1e-9, atol=1e-11), (kind, k)


_PLDA_SETUPS = {
    "a": (False, dict(epochs=6, ml_md="ml+md"), {}),
    "b": (True, dict(epochs=2, ml_md="ml"), {}),
    "c": (False, dict(epochs=5, ml_md="ml+md", md_epochs=[1, 3]), dict(update_D=False)),
    "d": (False, dict(epochs=3, ml_md="md"), {}),
    "e": (False, dict(epochs=3, ml_md="ml"), dict(update_mu=False)),
}
_PLDA_STATS = ("N", "M", "F", "S", "logpy", "y_acc", "Ry1", "Ry", "Cy", "Py", "Rz1", "Rz", "Ryz", "Cz")


def test_plda_training_restatement(golden):
    """EM of the full PLDA model in the oracle against the reference's first model, posterior,
    E-step accumulators, ELBO sequences and trained parameters (oracle/gen_golden.py gen_plda_training)."""
    g = golden("plda_training")
    x, y, w = g["x"], g["y"], g["w"]
    K, Z = int(g["y_dim"]), int(g["z_dim"])
    for tag, D in (("", O.compute_stats_hard(x, y)), ("w_", _weighted_stats(x, y, w))):
        for a, name in zip(O.plda_initialize(D, K, Z), ("mu", "V", "U", "D")):
            assert np.allclose(a, g["init_" + tag + name], rtol=1e-8, atol=1e-10), name
        m0 = dict({k: g["init_" + tag + k] for k in ("mu", "V", "U", "D")}, type="plda")
        got = O.plda_compute_py_g_x(m0, D, return_cov=True, return_logpy_0=True, return_acc=True)
        for a, name in zip(got, ("y", "cov", "logpy", "Ry", "Py")):
            assert np.allclose(a, g["post_" + tag + name], rtol=1e-9, atol=1e-11), name
        st = O.plda_estep(m0, D)
        for a, name in zip(st, _PLDA_STATS):
            ref = g["estep_" + tag + name]
            assert np.allclose(a, ref, rtol=1e-8, atol=1e-10 * max(1.0, np.abs(ref).max())), name
        assert np.isclose(O.plda_elbo(m0, st), g["elbo0_" + tag[:1]], rtol=1e-11)
    for tag, (weighted, kw, upd) in _PLDA_SETUPS.items():
        D = _weighted_stats(x, y, w) if weighted else O.compute_stats_hard(x, y)
        m, elbo = O.plda_fit(D, K, Z, **kw, **upd)
        assert np.allclose(elbo, g["fit_%s_elbo" % tag], rtol=1e-9)
        for p in ("mu", "V", "U", "D"):
            assert np.allclose(m[p], g["fit_%s_%s" % (tag, p)], rtol=1e-6, atol=1e-8), (tag, p)


@pytest.mark.parametrize("tag", ["", "q_"])
def test_detection_metric_restatements(golden, tag):
    """Hull, minimum / actual DCF over several priors, PRBEP and the combined evaluation of the oracle
    against the reference's values, on continuous scores and on scores with many ties."""
    g = golden("metrics_math")
    tar, non = (g["tar"], g["non"]) if tag == "" else (g["tar_q"], g["non_q"])
    priors = g["priors"]
    pm, pf = O.compute_rocch(tar, non)
    assert np.allclose(pm, g[tag + "rocch_pmiss"], rtol=0, atol=1e-13) and np.allclose(pf, g[tag + "rocch_pfa"], rtol=0, atol=1e-13)
    assert np.allclose(np.stack(O.compute_min_dcf_multi(tar, non, priors)), g[tag + "min_dcf"], rtol=1e-12)
    assert np.allclose(np.stack(O.compute_act_dcf(tar, non, priors)), g[tag + "act_dcf"], rtol=1e-12)
    one = O.compute_act_dcf(tar, non, 0.05, normalize=False)
    assert np.allclose([float(np.ravel(v)[0]) for v in one], g[tag + "act_dcf_005_raw"], rtol=1e-12)
    assert np.isclose(O.compute_prbep(tar, non), g[tag + "prbep"], rtol=1e-12)
    mn, act, eer, prbep = O.fast_eval_dcf_eer(tar, non, priors)
    assert np.allclose(mn, g[tag + "fast_min"], rtol=1e-12) and np.allclose(act, g[tag + "fast_act"], rtol=1e-12)
    assert np.isclose(eer, g[tag + "fast_eer"], rtol=1e-12) and np.isclose(prbep, g[tag + "fast_prbep"], rtol=1e-12)


def test_two_stage_score_norm_restatements(golden):
    """ZT- and TZ-norm of the oracle against the reference's predict bodies run on its own ZNorm and
    TNorm objects (oracle/gen_golden.py gen_score_norm)."""
    g = golden("score_norm")
    sc, ct, ec, cc = g["scores"], g["coh_test"], g["enr_coh"], g["coh_coh"]
    masks = (g["mask_coh_test"], g["mask_enr_coh"], g["mask_coh_coh"])
    assert np.allclose(O.zt_norm(sc, ct, ec, cc), g["ztnorm"], rtol=1e-10, atol=1e-10)
    assert np.allclose(O.tz_norm(sc, ct, ec, cc), g["tznorm"], rtol=1e-10, atol=1e-10)
    assert np.allclose(O.zt_norm(sc, ct, ec, cc, *masks), g["ztnorm_masked"], rtol=1e-10, atol=1e-10)
    assert np.allclose(O.tz_norm(sc, ct, ec, cc, *masks), g["tznorm_masked"], rtol=1e-10, atol=1e-10)
