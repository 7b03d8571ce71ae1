"""Detection metrics of explicit score arrays (hyperion_amd.metrics) against the reference's values
(tests/golden/metrics_math.npz), the bounds of the reference's own tests
(tests/hyperion/metrics/test_eer.py:14-29, test_dcf.py:14-44) and the oracle on larger sets."""
import numpy as np
import pytest

from oracle import plda_ref as O

pytestmark = pytest.mark.gpu


def _hull_equal(got, ref):
    """Same hull: every vertex of `got` is a vertex of `ref`, and the points `ref` has in addition
    lie on segments of `got` (the reference can keep a point that lies exactly on a segment)."""
    gm, gf = got
    rm, rf = ref
    pts = {(round(a, 12), round(b, 12)) for a, b in zip(rm, rf)}
    assert all((round(a, 12), round(b, 12)) in pts for a, b in zip(gm, gf))
    on = np.interp(rf[::-1], gf[::-1], gm[::-1])[::-1]
    assert np.allclose(on, rm, rtol=0, atol=1e-12)


def test_reference_test_bounds():
    from hyperion_amd import metrics as M

    tar = np.linspace(-2, 10, 1000) + 3
    non = np.linspace(-10, 2, 1000) + 3
    assert 0.166 < M.compute_eer(tar, non) < 0.167
    assert 0.332 < M.compute_min_dcf(tar, non, 0.5)[0] < 0.334
    assert np.isclose(M.effective_prior(0.01, 10, 1), 0.1 / 1.09)


@pytest.mark.parametrize("tag", ["", "q_"])
def test_metrics_match_reference(golden, tag):
    from hyperion_amd import metrics as M

    g = golden("metrics_math")
    tar, non = (g["tar"], g["non"]) if tag == "" else (g["tar_q"], g["non_q"])
    priors = g["priors"]
    _hull_equal(M.compute_rocch(tar, non), (g[tag + "rocch_pmiss"], g[tag + "rocch_pfa"]))
    if tag == "":
        assert np.isclose(M.compute_eer(tar, non), g["eer"], rtol=1e-12)
        assert np.isclose(M.compute_min_dcf(tar, non, 0.05)[0], g["min_dcf_005"], rtol=1e-12)
    assert np.allclose(np.stack(M.compute_min_dcf(tar, non, priors)), g[tag + "min_dcf"], rtol=1e-12)
    assert np.allclose(np.stack(M.compute_act_dcf(tar, non, priors)), g[tag + "act_dcf"], rtol=1e-12)
    one = M.compute_act_dcf(tar, non, 0.05, normalize=False)
    assert np.allclose([float(np.ravel(v)[0]) for v in one], g[tag + "act_dcf_005_raw"], rtol=1e-12)
    assert np.ndim(one[0]) == 0
    assert np.isclose(M.compute_prbep(tar, non), g[tag + "prbep"], rtol=1e-12)
    mn, act, eer, prbep = M.fast_eval_dcf_eer(tar, non, priors)
    assert np.allclose(mn, g[tag + "fast_min"], rtol=1e-12) and np.allclose(act, g[tag + "fast_act"], rtol=1e-12)
    assert np.isclose(eer, g[tag + "fast_eer"], rtol=1e-12) and np.isclose(prbep, g[tag + "fast_prbep"], rtol=1e-12)
    full = M.fast_eval_dcf_eer(tar, non, priors, return_probs=True)
    assert len(full) == 8 and np.allclose(full[4], g[tag + "min_dcf"][1]) and np.allclose(full[6], g[tag + "act_dcf"][1])
    pm, pf = M.compute_rocch(tar, non)
    assert np.allclose(M.compute_dcf(pm, pf, priors).min(axis=1), g[tag + "min_dcf"][0], rtol=1e-12)
    with pytest.raises(AssertionError):
        M.compute_act_dcf(tar, non, priors[::-1])


def test_large_set_and_device_scores_against_oracle():
    """200 000 targets and 2 000 000 non-targets, given as a device float32 tensor and as host
    float64 arrays; scores straight from llr_1vs1 with a TrialKey-style mask."""
    import torch

    from hyperion_amd import metrics as M

    rng = np.random.RandomState(12)
    tar = (rng.randn(200_000) * 4 + 8).astype(np.float32)
    non = (rng.randn(2_000_000) * 8 - 14).astype(np.float32)
    ref_eer = O.compute_eer(tar.astype(np.float64)[::20], non.astype(np.float64)[::20])
    assert np.isclose(M.compute_eer(tar[::20], non[::20]), ref_eer, rtol=1e-12)
    e_host = M.compute_eer(tar, non)
    e_dev = M.compute_eer(torch.from_numpy(tar).cuda(), torch.from_numpy(non).cuda())
    assert e_host == e_dev and abs(e_host - ref_eer) < 3e-3
    d = M.compute_min_dcf(tar, non, 0.05)
    assert 0 < d[0] < 1 and np.isclose(d[0], (0.05 * d[1] + 0.95 * d[2]) / 0.05)
    with pytest.raises(AssertionError):
        M.compute_eer(tar, np.zeros(0))
