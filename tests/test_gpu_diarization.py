"""The diarization consumer on the device (pre-processing, recording-level PCA, projected SPLDA,
self-scoring) against the reference's cluster ids and linkage (tests/golden/diarization.npz)."""
import numpy as np
import pytest

from conftest import model_from
from oracle import plda_ref as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,var_r,thr", (("full", 1, 4.0), ("pca", 0.8, 2.0)))
def test_cluster_matches_reference(golden, tag, var_r, thr):
    from hyperion_amd.diarization import DiarAHCPLDA
    from hyperion_amd.pdfs.plda import SPLDA
    from hyperion_amd.transforms import LNorm

    g = golden("diarization")
    m = model_from(g, "m_", "splda")
    diar = DiarAHCPLDA(SPLDA(mu=m["mu"], V=m["V"], W=m["W"]), preproc=LNorm(), threshold=thr, pca_var_r=var_r)
    ids = diar.cluster(g["x"])
    assert np.array_equal(ids, g["ids_" + tag])
    ref = g["Z_" + tag]
    assert np.array_equal(diar._ahc.Z[:, :2], ref[:, :2]) and np.array_equal(diar._ahc.Z[:, 3], ref[:, 3])
    assert np.allclose(diar._ahc.Z[:, 2], ref[:, 2], rtol=1e-4, atol=1e-3)


def test_recording_with_many_segments():
    """2000 segments of 6 speakers in 128 dimensions: same partition as the float64 oracle."""
    from hyperion_amd.diarization import DiarAHCPLDA
    from hyperion_amd.pdfs.plda import SPLDA
    from hyperion_amd.transforms import LNorm

    import synth

    model = synth.make_splda(128, 40)
    rng = np.random.RandomState(4)
    spk = rng.randint(0, 6, 2000)
    x = model["mu"] + (1.5 * rng.randn(6, 40))[spk] @ model["V"] + rng.randn(2000, 128) @ np.linalg.cholesky(
        np.linalg.inv(model["W"])).T
    diar = DiarAHCPLDA(SPLDA(mu=model["mu"], V=model["V"], W=model["W"]), preproc=LNorm(), threshold=0.0,
                       pca_var_r=0.9)
    ids = diar.cluster(x)
    want, _ = O.diar_ahc_plda_cluster(model, x, 0.0, 0.9)
    assert np.array_equal(ids, want)


@pytest.mark.parametrize("tag,key", [("three", "x"), ("one", "x_one")])
def test_unsupervised_calibration_matches_reference(golden, tag, key):
    """The two-Gaussian calibration of the self-scores and the BIC decision (diar_ahc_plda.py:100-120,
    156-159): mixture, BIC and cluster ids of the reference for a three-speaker recording (BIC > 0,
    clustered) and a short single-speaker one (BIC < 0, one cluster)."""
    from hyperion_amd.diarization import DiarAHCPLDA
    from hyperion_amd.pdfs.plda import SPLDA
    from hyperion_amd.transforms import LNorm

    g = golden("diarization")
    m = {k: g["m_" + k] for k in ("mu", "V", "W")}
    x = g[key]
    diar = DiarAHCPLDA(SPLDA(**m), preproc=LNorm(), threshold=4.0, do_unsup_cal=True, use_bic=True)
    scores = diar.scores(x)
    cal, bic, (pi, mu, lam) = diar._unsup_gmm_calibration(scores)
    # the device scores carry the score path's 1e-4 tolerance into the mixture (the sharp mixture of the
    # single-speaker recording, shared deviation 0.024, amplifies it most); exact scores are checked below
    assert np.isclose(bic, g["cal_%s_bic" % tag], rtol=1e-4, atol=1e-6)
    assert np.allclose(pi, g["cal_%s_pi" % tag], rtol=1e-3) and np.allclose(mu, g["cal_%s_mu" % tag], rtol=1e-4)
    assert np.isclose(lam, g["cal_%s_Lambda" % tag][0], rtol=1e-3)
    assert cal.is_cuda and cal.shape == scores.shape
    # host scores in, host scores out; the linear map is the log-likelihood ratio of the components
    xl = O.lnorm_predict(x)
    ref_scores = O.splda_llr_1vs1(xl, xl, m["mu"], m["V"], m["W"])
    scale, bias, ref_bic = O.unsup_gmm_calibration(ref_scores)[:3]
    cal_h, bic_h, _ = diar._unsup_gmm_calibration(ref_scores)
    assert isinstance(cal_h, np.ndarray) and np.allclose(cal_h, scale * ref_scores + bias, rtol=1e-8, atol=1e-8)
    assert np.isclose(bic_h, ref_bic, rtol=1e-9, atol=1e-11)
    ids = diar.cluster(x)
    assert np.array_equal(ids, g["ids_bic_" + tag])
    assert (len(np.unique(ids)) == 1) == (bic < 0)
    assert DiarAHCPLDA(SPLDA(**m), use_bic=True).use_bic is False  # needs do_unsup_cal, as in the reference
