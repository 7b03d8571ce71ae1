"""ROC convex hull metrics from binned scores (host logic of hyperion_amd.metrics) against the
oracle's restatement of the reference metrics evaluated on the same scores rounded to their bins."""
import numpy as np
import pytest

from hyperion_amd.metrics import eer_from_rocch, min_dcf_from_rocch, rocch_from_counts
from oracle import plda_ref as O


def _bins(s, lo, hi, nbins):
    scale = np.float32(nbins) / (np.float32(hi) - np.float32(lo))
    f = np.floor((s.astype(np.float32) - np.float32(lo)) * scale)
    return np.clip(f, 0, nbins - 1).astype(np.int64)


@pytest.mark.parametrize("seed,n_tar,n_non,nbins", [(0, 300, 5000, 64), (1, 50, 400, 16), (2, 1000, 20000, 1024),
                                                    (3, 7, 9, 4), (4, 200, 200, 4096)])
def test_hull_from_counts_equals_reference_on_binned_scores(seed, n_tar, n_non, nbins):
    rng = np.random.RandomState(seed)
    tar = (8 + 4 * rng.randn(n_tar)).astype(np.float32)
    non = (-14 + 8 * rng.randn(n_non)).astype(np.float32)
    lo, hi = -60.0, 40.0
    bt, bn = _bins(tar, lo, hi, nbins), _bins(non, lo, hi, nbins)
    ct, cn = np.bincount(bt, minlength=nbins), np.bincount(bn, minlength=nbins)
    p_miss, p_fa = rocch_from_counts(ct, cn)
    rm, rf = O.compute_rocch(bt.astype(float), bn.astype(float))  # ties inside a bin: targets first
    assert p_miss.shape == rm.shape
    assert np.allclose(p_miss, rm, atol=1e-14) and np.allclose(p_fa, rf, atol=1e-14)
    assert abs(eer_from_rocch(p_miss, p_fa) - O.compute_eer(bt.astype(float), bn.astype(float))) <= 1e-12
    for prior in (0.05, 0.01, 0.5):
        got = min_dcf_from_rocch(p_miss, p_fa, prior)
        ref = O.compute_min_dcf(bt.astype(float), bn.astype(float), prior)
        assert np.allclose(got, ref, atol=1e-12)


def test_binned_metrics_converge_to_unbinned():
    rng = np.random.RandomState(9)
    tar = 8 + 4 * rng.randn(4000)
    non = -14 + 8 * rng.randn(40000)
    eer = O.compute_eer(tar, non)
    for nbins, tol in ((256, 4e-3), (4096, 4e-4)):
        ct = np.bincount(_bins(tar, -60, 40, nbins), minlength=nbins)
        cn = np.bincount(_bins(non, -60, 40, nbins), minlength=nbins)
        assert abs(eer_from_rocch(*rocch_from_counts(ct, cn)) - eer) <= tol


def test_degenerate_hulls():
    # perfectly separated: EER 0; fully overlapping single bin: EER 0.5
    assert eer_from_rocch(*rocch_from_counts([0, 0, 5], [7, 0, 0])) == 0.0
    assert abs(eer_from_rocch(*rocch_from_counts([5], [7])) - 0.5) <= 1e-15
    with pytest.raises(AssertionError):
        rocch_from_counts([0, 0], [1, 2])
