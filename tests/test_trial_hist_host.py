"""ROC convex hull metrics from binned scores (host logic of hyperion_amd.metrics) against the
oracle's restatement of the reference metrics evaluated on the same scores rounded to their bins."""
import numpy as np
import pytest

from hyperion_amd.metrics import eer_from_rocch, min_dcf_from_rocch, rocch_from_counts
from oracle import plda_ref as O


def _strict_vertices(p_miss, p_fa, n_tar, n_non):
    """Hull vertices as integer (missed targets, accepted non-targets) pairs without the points that
    lie exactly on the segment between their neighbours.  (The reference pools blocks of equal level
    with a floating-point comparison of incrementally updated means, which now and then keeps such
    a point; exact arithmetic never does.  The hull as a curve, EER and DCF are the same.)"""
    pts = [(int(round(m * n_tar)), int(round(f * n_non))) for m, f in zip(p_miss, p_fa)]
    keep = [pts[0]]
    for i in range(1, len(pts) - 1):
        (m0, f0), (m1, f1), (m2, f2) = keep[-1], pts[i], pts[i + 1]
        if (m1 - m0) * (f2 - f1) != (m2 - m1) * (f1 - f0):
            keep.append(pts[i])
    return keep + [pts[-1]]


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


@pytest.mark.parametrize("seed,n_tar,n_non,nbins", [(0, 2000, 100000, 256), (1, 300, 5000, 64), (3, 50, 400, 8),
                                                    (4, 2000, 2000, 32)])
def test_one_refinement_of_the_open_bins_gives_the_exact_hull(seed, n_tar, n_non, nbins):
    """Host logic of the second pass: the bins whose best ordering beats the worst-ordering hull are
    the only ones that need their individual scores; with those known the hull is the oracle's."""
    from hyperion_amd.metrics.trial_hist import bins_that_can_hold_a_vertex

    rng = np.random.RandomState(seed)
    tar = (8 + 4 * rng.randn(n_tar)).astype(np.float32)
    non = (-14 + 8 * rng.randn(n_non)).astype(np.float32)
    if seed == 3:
        tar, non = np.round(tar), np.round(non)  # many exact ties between the classes
    bt, bn = _bins(tar, -60, 40, nbins), _bins(non, -60, 40, nbins)
    ct, cn = np.bincount(bt, minlength=nbins), np.bincount(bn, minlength=nbins)
    todo = bins_that_can_hold_a_vertex(ct, cn)
    assert 0 < len(todo) < nbins
    refined = {int(b): (np.sort(tar[bt == b]), np.sort(non[bn == b])) for b in todo}
    assert len(bins_that_can_hold_a_vertex(ct, cn, refined)) == 0
    p_miss, p_fa, still_open = rocch_from_counts(ct, cn, refined, optimistic=True, return_open=True)
    assert len(still_open) == 0
    rm, rf = O.compute_rocch(tar.astype(float), non.astype(float))
    assert _strict_vertices(p_miss, p_fa, n_tar, n_non) == _strict_vertices(rm, rf, n_tar, n_non)
    assert abs(eer_from_rocch(p_miss, p_fa) - O.compute_eer(tar.astype(float), non.astype(float))) <= 1e-14
    assert np.allclose(min_dcf_from_rocch(p_miss, p_fa, 0.05), O.compute_min_dcf(tar.astype(float), non.astype(float), 0.05),
                       rtol=0, atol=1e-14)
    pm2, pf2 = rocch_from_counts(ct, cn, refined)  # the ordering of the other bins no longer matters
    assert np.array_equal(pm2, p_miss) and np.array_equal(pf2, p_fa)
