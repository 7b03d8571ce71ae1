"""Score sink on the device (hypmi_score_hist through hyperion_amd.metrics.TrialHistogram): bin counts
bit-exact against numpy on the same float32 scores, moments against float64 sums, metrics against
the oracle on the binned scores."""
import numpy as np
import pytest

from oracle import plda_ref as O

pytestmark = pytest.mark.gpu


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


def _ref(s, cls, lo, hi, nbins):
    """cls: 1 target, 0 non-target, -1 no trial; returns hist [2, nbins] and moments [2, 4]."""
    hist = np.zeros((2, nbins), dtype=np.int64)
    mom = np.zeros((2, 4))
    for c in (0, 1):
        v = s[cls == c]
        fin = np.isfinite(v)
        hist[c] = np.bincount(_bins(v[fin], lo, hi, nbins), minlength=nbins)
        w = v[fin].astype(np.float64)
        mom[c] = [fin.sum(), w.sum(), (w * w).sum(), (~fin).sum()]
    return hist, mom


def _check(h, hist, mom):
    assert np.array_equal(h.hist.cpu().numpy(), hist)
    got = h.mom.cpu().numpy()
    assert np.array_equal(got[:, [0, 3]], mom[:, [0, 3]])
    assert np.allclose(got[:, 1:3], mom[:, 1:3], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("ne,nt,nbins", [(1, 1, 1), (3, 5, 7), (37, 1023, 64), (64, 1024, 4096), (130, 4099, 8192),
                                         (1000, 3000, 4096), (5, 70001, 512)])
def test_hist_with_keys_is_exact(ne, nt, nbins):
    import torch

    from hyperion_amd.metrics import TrialHistogram

    rng = np.random.RandomState(ne + nt)
    s = (-5 + 12 * rng.randn(ne, nt)).astype(np.float32)
    ke = rng.randint(-1, 6, ne).astype(np.int32)
    kt = rng.randint(-1, 6, nt).astype(np.int32)
    cls = np.where((ke[:, None] < 0) | (kt[None, :] < 0), -1, (ke[:, None] == kt[None, :]).astype(int))
    lo, hi = -30.0, 25.0  # some scores fall outside and are clamped into the end bins
    h = TrialHistogram(lo, hi, nbins).add(torch.from_numpy(s).cuda(), ke, kt)
    _check(h, *_ref(s, cls, lo, hi, nbins))


def test_hist_masks_pitched_blocks_and_accumulation():
    """TrialKey-style masks (tar + non, tar only), no labels, a block that is a column slice of a wider
    matrix (unaligned rows), and two blocks added to the same histogram."""
    import torch

    from hyperion_amd.metrics import TrialHistogram

    rng = np.random.RandomState(3)
    ne, nt, nbins = 77, 2050, 1000
    wide = (4 * rng.randn(ne, nt + 9)).astype(np.float32)
    wide[5, 17] = np.nan
    wide[6, 18] = np.inf
    wide[7, 2000] = -np.inf
    S = torch.from_numpy(wide).cuda()[:, 3:3 + nt]
    s = wide[:, 3:3 + nt]
    tar = rng.rand(ne, nt) < 0.02
    non = (rng.rand(ne, nt) < 0.7) & ~tar
    lo, hi = -16.0, 16.0
    cls = np.where(tar, 1, np.where(non, 0, -1))
    _check(TrialHistogram(lo, hi, nbins).add(S, tar=tar, non=non), *_ref(s, cls, lo, hi, nbins))
    _check(TrialHistogram(lo, hi, nbins).add(S, tar=torch.from_numpy(tar).cuda()),
           *_ref(s, tar.astype(int), lo, hi, nbins))
    _check(TrialHistogram(lo, hi, nbins).add(S), *_ref(s, np.zeros_like(cls), lo, hi, nbins))
    h = TrialHistogram(lo, hi, nbins)
    h.add(S[:, :1000], tar=tar[:, :1000], non=non[:, :1000]).add(S[:, 1000:], tar=tar[:, 1000:], non=non[:, 1000:])
    _check(h, *_ref(s, cls, lo, hi, nbins))
    m = h.moments()
    assert m["tar"]["count"] == int((np.isfinite(s) & tar).sum())
    assert m["non"]["nonfinite"] + m["tar"]["nonfinite"] == int((~np.isfinite(s) & (tar | non)).sum())


def test_sink_metrics_match_oracle_on_binned_scores():
    """VoxCeleb-size SPLDA matrix reduced on the device: EER / minDCF from the histograms equal the
    oracle's on the same scores rounded to their bin, and are close to the unbinned values."""
    import synth
    from hyperion_amd.metrics import TrialHistogram
    from hyperion_amd.pdfs.plda import SPLDA

    m, x, cls = synth.voxceleb_like()
    import torch

    S = SPLDA(mu=m["mu"], V=m["V"], W=m["W"]).llr_1vs1(torch.from_numpy(x.astype(np.float32)).cuda(),
                                                      torch.from_numpy(x.astype(np.float32)).cuda())
    lo, hi, nbins = -120.0, 60.0, 4096
    h = TrialHistogram(lo, hi, nbins).add(S, cls, cls)
    s = S.cpu().numpy()
    is_tar = cls[:, None] == cls[None, :]
    bt, bn = _bins(s[is_tar], lo, hi, nbins), _bins(s[~is_tar], lo, hi, nbins)
    non, tar = h.counts()
    assert np.array_equal(tar, np.bincount(bt, minlength=nbins))
    assert np.array_equal(non, np.bincount(bn, minlength=nbins))
    # the oracle on every binned score would sort 23.8 M values; sub-sample as the known answer does
    ct = np.bincount(bt[::20], minlength=nbins)
    cn = np.bincount(bn[::100], minlength=nbins)
    from hyperion_amd.metrics import eer_from_rocch, min_dcf_from_rocch, rocch_from_counts

    hull = rocch_from_counts(ct, cn)
    assert abs(eer_from_rocch(*hull) - O.compute_eer(bt[::20].astype(float), bn[::100].astype(float))) <= 1e-12
    ref_eer = O.compute_eer(s[is_tar][::20].astype(float), s[~is_tar][::100].astype(float))
    assert abs(eer_from_rocch(*hull) - ref_eer) <= 2e-4
    ref_dcf = O.compute_min_dcf(s[is_tar][::20].astype(float), s[~is_tar][::100].astype(float), 0.05)[0]
    assert abs(min_dcf_from_rocch(*hull, 0.05)[0] - ref_dcf) <= 2e-3
    assert abs(h.eer() - ref_eer) <= 2e-3  # all trials against the sub-sample


def test_snorm_pipeline_streams_into_the_sink():
    """score_snorm(reduce=hist.sink(...)) equals the histogram of the materialised matrix."""
    import synth
    from hyperion_amd.backend import PLDABackend
    from hyperion_amd.metrics import TrialHistogram
    from hyperion_amd.pdfs.plda import SPLDA

    m = synth.make_splda(40, 12)
    rng = np.random.RandomState(1)
    xe, xt, xc = (synth.length_norm(rng.randn(n, 40)) for n in (60, 517, 300))
    ke, kt = rng.randint(0, 20, 60), rng.randint(0, 20, 517)
    be = PLDABackend(SPLDA(mu=m["mu"], V=m["V"], W=m["W"]), None, nbest=40, test_chunk=128)
    coh = be.prepare_cohort(xe, xc)
    full = be.score_snorm(xt, coh)
    a = TrialHistogram(-12, 12, 2048).add(full, ke, kt)
    b = TrialHistogram(-12, 12, 2048)
    assert be.score_snorm(xt, coh, reduce=b.sink(ke, kt)) is None
    assert np.array_equal(a.hist.cpu().numpy(), b.hist.cpu().numpy())
    assert np.allclose(a.mom.cpu().numpy(), b.mom.cpu().numpy(), rtol=1e-12)
    assert int(a.hist.sum()) == 60 * 517
    # second streamed pass: exact hull without ever holding the matrix
    second = b.exact_pass()
    assert be.score_snorm(xt, coh, reduce=second.sink(ke, kt)) is None
    second.finish()
    s = full.cpu().numpy().astype(np.float64)
    is_tar = ke[:, None] == kt[None, :]
    assert abs(b.eer() - O.compute_eer(s[is_tar], s[~is_tar])) <= 1e-14


@pytest.mark.parametrize("ne,nt,nbins,ncls", [(300, 4000, 256, 40), (64, 1500, 32, 8), (1000, 6000, 4096, 200)])
def test_second_pass_makes_the_hull_exact(ne, nt, nbins, ncls):
    """Histogram pass, then ONE extraction pass over the bins that can hold a hull vertex: the ROC
    convex hull, EER and minDCF equal the oracle's on all individual scores."""
    import torch

    from hyperion_amd.metrics import TrialHistogram

    rng = np.random.RandomState(ne)
    ke, kt = rng.randint(0, ncls, ne).astype(np.int32), rng.randint(0, ncls, nt).astype(np.int32)
    is_tar = ke[:, None] == kt[None, :]
    s = np.where(is_tar, 6 + 4 * rng.randn(ne, nt), -10 + 7 * rng.randn(ne, nt)).astype(np.float32)
    s[3, 5] = np.nan  # skipped by both passes
    S = torch.from_numpy(s).cuda()
    h = TrialHistogram(-45.0, 30.0, nbins).add(S, ke, kt)
    assert not h.is_exact()
    coarse = h.eer()
    second = h.exact_pass()
    second.add(S[:, :nt // 2], ke, kt[:nt // 2]).add(S[:, nt // 2:], ke, kt[nt // 2:])  # blocks may differ in shape
    second.finish()
    assert h.is_exact() and h.exact_pass() is None
    ok = np.isfinite(s)
    tar, non = s[is_tar & ok].astype(np.float64), s[~is_tar & ok].astype(np.float64)
    rm, rf = O.compute_rocch(tar, non)
    pm, pf = h.rocch()
    assert _strict_vertices(pm, pf, len(tar), len(non)) == _strict_vertices(rm, rf, len(tar), len(non))
    assert abs(h.eer() - O.compute_eer(tar, non)) <= 1e-14 and abs(coarse - h.eer()) > 0
    assert np.allclose(h.min_dcf(0.05), O.compute_min_dcf(tar, non, 0.05), rtol=0, atol=1e-14)
    # the fraction of scores the second pass had to keep
    assert sum(second.expect) < 0.6 * ne * nt


def test_second_pass_detects_different_scores():
    import torch

    from hyperion_amd.metrics import TrialHistogram

    rng = np.random.RandomState(0)
    ke, kt = rng.randint(0, 5, 50).astype(np.int32), rng.randint(0, 5, 700).astype(np.int32)
    S = torch.from_numpy((4 * rng.randn(50, 700)).astype(np.float32)).cuda()
    h = TrialHistogram(-20.0, 20.0, 64).add(S, ke, kt)
    second = h.exact_pass().add(S + 0.5, ke, kt)
    with pytest.raises(RuntimeError, match="did not see the same scores"):
        second.finish()


def test_full_extraction_is_get_tar_non():
    """Every cell flagged: the device lists are TrialScores.get_tar_non up to order."""
    import torch

    from hyperion_amd.metrics.trial_hist import ExactPass, TrialHistogram

    rng = np.random.RandomState(2)
    tar = rng.rand(40, 900) < 0.05
    non = (rng.rand(40, 900) < 0.6) & ~tar
    s = (3 * rng.randn(40, 900)).astype(np.float32)
    S = torch.from_numpy(s).cuda()
    h = TrialHistogram(-15.0, 15.0, 128).add(S, tar=tar, non=non)
    p = ExactPass(h, np.arange(128)).add(S, tar=tar, non=non)
    n_non, n_tar = (int(c) for c in p.count.cpu())
    assert (n_non, n_tar) == (int(non.sum()), int(tar.sum()))
    assert np.array_equal(np.sort(p.out[1, :n_tar].cpu().numpy()), np.sort(s[tar]))
    assert np.array_equal(np.sort(p.out[0, :n_non].cpu().numpy()), np.sort(s[non]))
