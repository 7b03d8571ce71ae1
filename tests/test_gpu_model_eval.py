"""Model-evaluation methods around the scoring path on the device against the reference's outputs
(tests/golden/model_eval.npz): conditional log-densities, the pooling variants under their own
names, the one-hot statistics, compute_aux and the PCA dimension chooser."""
import numpy as np
import pytest

from conftest import model_from

pytestmark = pytest.mark.gpu


def _model(g, kind):
    from hyperion_amd.pdfs.plda import FRPLDA, PLDA, SPLDA

    m = model_from(g, kind + "_", kind)
    cls = {"splda": SPLDA, "frplda": FRPLDA, "plda": PLDA}[kind]
    return cls(**{k: v for k, v in m.items() if k != "type"})


def test_log_densities_match_reference(golden):
    g = golden("model_eval")
    x, y, z = g["x"], g["y"], g["z"]
    got = _model(g, "splda").log_probx_g_y(x, y)
    assert got.shape == (x.shape[0],) and got.dtype == np.float64
    assert np.allclose(got, g["splda_logp_y"], rtol=1e-11, atol=1e-11)
    pl = _model(g, "plda")
    assert np.allclose(pl.log_probx_g_y(x, y), g["plda_logp_y"], rtol=1e-10, atol=1e-10)
    assert np.allclose(pl.log_probx_g_yz(x, y, z), g["plda_logp_yz"], rtol=1e-11, atol=1e-11)
    # diagonal within-class precision given as a vector
    from hyperion_amd.pdfs.plda import SPLDA

    w = np.diag(g["splda_W"]).copy()
    d = SPLDA(mu=g["splda_mu"], V=g["splda_V"], W=w, fullcov_W=False).log_probx_g_y(x, y)
    f = SPLDA(mu=g["splda_mu"], V=g["splda_V"], W=np.diag(w)).log_probx_g_y(x, y)
    assert np.allclose(d, f, rtol=1e-12)


def test_compute_aux_sets_the_reference_attributes(golden):
    g = golden("model_eval")
    pl = _model(g, "plda")
    pl.compute_aux()
    for k in ("_DU", "_J", "_Lz", "_log_Lz", "_W", "_VW", "_VWV"):
        ref = g["plda_aux" + k]
        assert np.allclose(np.asarray(getattr(pl, k)), ref, rtol=1e-10, atol=1e-11 * max(1.0, np.abs(ref).max())), k
    v = np.arange(3.0 * 2).reshape(2, 3)
    assert np.allclose(pl._mult_iLz(v) @ pl._Lz, v)


@pytest.mark.parametrize("kind", ["splda", "frplda", "plda"])
def test_named_pooling_variants_match_reference(golden, kind):
    g = golden("model_eval")
    m = _model(g, kind)
    x, ids, x2, ids2 = g["x"], g["ids"], g["x2"], g["ids2"]
    tol = dict(rtol=1e-4, atol=1e-4)  # the scoring path computes in fp32: 1e-4 relative, the bound its parity tests use
    D1, D2 = m.compute_stats_hard(x, ids), m.compute_stats_hard(x2, ids2)
    for ln in (True, False):
        tag = "%s_%s" % (kind, "ln" if ln else "raw")
        assert np.allclose(m.llr_NvsM_vavg(D1, D2, do_lnorm=ln), g["nvsm_vavg_" + tag], **tol)
        assert np.allclose(m.llr_Nvs1_vavg(D1, x2, do_lnorm=ln), g["nvs1_vavg_" + tag], **tol)
    assert np.allclose(m.llr_NvsM_savg(x, ids, x2, ids2), g["nvsm_savg_" + kind], **tol)
    assert np.allclose(m.llr_Nvs1_savg(x, ids, x2), g["nvs1_savg_" + kind], **tol)


def test_one_hot_statistics_and_pca_dimension(golden):
    from hyperion_amd.pdfs.plda import SPLDA
    from hyperion_amd.transforms import PCA

    g = golden("model_eval")
    N, F, S = SPLDA.compute_stats_hard_v0(g["x"], g["ids"])
    assert np.array_equal(N, g["v0_N"]) and np.allclose(F, g["v0_F"], rtol=1e-12, atol=1e-12)
    assert np.allclose(S, g["v0_S"], rtol=1e-12, atol=1e-11)
    for tag, var_r, min_dim, want in g["pca_dim_cases"]:
        xx = g["pca_x_full"] if tag == 0 else g["pca_x_def"]
        assert PCA.get_pca_dim_for_var_ratio(xx, var_r if var_r != 1 else 1, int(min_dim)) == int(want), (tag, var_r)


def test_observed_roc_and_calibration_match_reference(golden):
    from hyperion_amd.metrics import compute_roc, opt_loglr

    g = golden("roc_cal_cos")
    tar, non = g["tar"], g["non"]
    pm, pf = compute_roc(tar, non)
    assert np.array_equal(pm, g["roc_pmiss"]) and np.array_equal(pf, g["roc_pfa"])  # counts: exact
    for method in ("laplace", "raw"):
        t, n = opt_loglr(tar, non, method)
        assert np.array_equal(t, g["cal_tar_" + method]) and np.array_equal(n, g["cal_non_" + method])
    import torch

    pm2, pf2 = compute_roc(torch.from_numpy(tar).cuda(), torch.from_numpy(non).cuda())  # device arrays in
    assert np.array_equal(pm2, pm) and np.array_equal(pf2, pf)


@pytest.mark.parametrize("d", [512, 37])
def test_cosine_scoring_matches_reference(golden, d):
    import torch

    from hyperion_amd.utils.math import cosine_scoring

    g = golden("roc_cal_cos")
    x1, x2, ref = g["cos%d_x1" % d], g["cos%d_x2" % d], g["cos%d" % d]
    got = cosine_scoring(x1, x2)
    assert got.dtype == np.float64 and got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 1e-5  # |cos| <= 1: absolute = the path's mixed tolerance
    dev = cosine_scoring(torch.from_numpy(x1).float().cuda(), torch.from_numpy(x2).float().cuda())
    assert dev.is_cuda and np.max(np.abs(dev.cpu().numpy() - ref)) <= 1e-5
    with pytest.raises(ValueError):
        cosine_scoring(x1, x2[:, :-1])
