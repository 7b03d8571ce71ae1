"""Score sink on the device (hypmi_score_hist through hyperion_amd.metrics.TrialHistogram): bin counts
bit-exact against numpy on the same float32 scores, moments against float64 sums, metrics against
the oracle on the binned scores."""
import numpy as np
import pytest

from oracle import plda_ref as O

pytestmark = pytest.mark.gpu


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
