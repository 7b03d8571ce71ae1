"""Host side of the diarization consumer: the oracle's restatement and the package's AHC against the
reference's linkage and cluster ids (tests/golden/diarization.npz, made by oracle/gen_golden.py
gen_diarization)."""
import numpy as np
import pytest

from conftest import model_from
from oracle import plda_ref as O

CASES = (("full", 1, 4.0), ("pca", 0.8, 2.0))


@pytest.mark.parametrize("tag,var_r,thr", CASES)
def test_oracle_restates_the_consumer(golden, tag, var_r, thr):
    g = golden("diarization")
    ids, Z = O.diar_ahc_plda_cluster(model_from(g, "m_", "splda"), g["x"], thr, var_r)
    assert np.array_equal(ids, g["ids_" + tag])
    assert np.allclose(Z, g["Z_" + tag], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("tag,var_r,thr", CASES)
def test_ahc_class_matches_reference_tree_and_labels(golden, tag, var_r, thr):
    from hyperion_amd.clustering import AHC

    g = golden("diarization")
    m = model_from(g, "m_", "splda")
    x = O.lnorm_predict(g["x"])
    mu, V, W = m["mu"], m["V"], m["W"]
    if var_r < 1:
        pmu, T = O.pca_fit(x, pca_var_r=var_r, whiten=True)
        x = O.lnorm_predict((x - pmu) @ T)
        mu, V, W = O.splda_project(mu, V, W, T, pmu)
    ahc = AHC()
    ahc.fit(O.splda_llr_1vs1(x, x, mu, V, W))
    assert np.allclose(ahc.Z, g["Z_" + tag], rtol=1e-9, atol=1e-9)
    assert np.array_equal(ahc.get_flat_clusters(thr), g["ids_" + tag])
    n = len(g["x"])
    for k in (1, 2, 5, n, n + 3):
        assert np.array_equal(ahc.get_flat_clusters(k, criterion="num_clusters"), O.ahc_flat_clusters_n(ahc.Z, k))


def test_ahc_mask_and_other_metrics():
    from hyperion_amd.clustering import AHC

    rng = np.random.RandomState(0)
    pts = np.concatenate([rng.randn(5, 2), rng.randn(5, 2) + 6.0])
    dist = np.linalg.norm(pts[:, None] - pts[None], axis=-1)
    a = AHC(metric="prob")
    a.fit(np.exp(-dist))
    ids = a.get_flat_clusters(2, criterion="num_clusters")
    assert len(set(ids[:5])) == 1 and len(set(ids[5:])) == 1 and ids[0] != ids[5]
    b = AHC()
    mask = np.ones((10, 10), dtype=bool)
    mask[0, 1] = mask[1, 0] = False
    b.fit(-dist, mask=mask)
    two = b.get_flat_clusters(2, criterion="num_clusters")
    assert two[0] != two[1]  # the forbidden pair stays apart down to the last merge
    free = AHC()
    free.fit(-dist)
    assert free.get_flat_clusters(2, criterion="num_clusters")[0] == free.get_flat_clusters(2, criterion="num_clusters")[1]


def test_oracle_restates_the_unsupervised_calibration(golden):
    """Mixture, BIC and the BIC decision of the oracle against the reference's
    (hyperion/diarization/diar_ahc_plda.py:100-120, gmm_tied_diag_cov.py)."""
    g = golden("diarization")
    m = {"mu": g["m_mu"], "V": g["m_V"], "W": g["m_W"]}
    for tag, key in (("three", "x"), ("one", "x_one")):
        xl = O.lnorm_predict(g[key])
        scale, bias, bic, pi, mu, lam = O.unsup_gmm_calibration(O.splda_llr_1vs1(xl, xl, m["mu"], m["V"], m["W"]))
        assert np.isclose(bic, g["cal_%s_bic" % tag], rtol=1e-9)
        assert np.allclose(pi, g["cal_%s_pi" % tag], rtol=1e-9) and np.allclose(mu, g["cal_%s_mu" % tag], rtol=1e-9)
        assert np.isclose(lam, g["cal_%s_Lambda" % tag][0], rtol=1e-9)
        assert np.array_equal(O.diar_ahc_plda_cluster(m, g[key], 4.0, 1, use_bic=True)[0], g["ids_bic_" + tag])


def test_every_level_of_the_tree_matches_reference(golden):
    """compute_flat_clusters and the homogeneity / completeness curve against the reference's
    (tests/golden/ahc_levels.npz, hyperion/clustering/ahc.py:115-158)."""
    from hyperion_amd.clustering import AHC

    g = golden("ahc_levels")
    a = AHC(method="average", metric="llr")
    a.fit(g["scores"])
    assert np.allclose(a.Z, g["Z"], rtol=1e-12, atol=1e-12)
    h, c = a.evaluate_homogeneity_completeness_tradeoff(g["true"])  # builds the table on demand
    assert np.array_equal(a.flat_clusters, g["flat_clusters"])
    assert np.allclose(h, g["homogeneity"], atol=1e-12) and np.allclose(c, g["completeness"], atol=1e-12)
    n = g["scores"].shape[0]
    for i in (0, 1, n // 2, n - 1):
        assert np.array_equal(a.flat_clusters[i], a.get_flat_clusters_from_num_clusters(n - i))
    a.fit(g["scores"][:5, :5])  # a new tree drops the old table
    assert a.flat_clusters is None
