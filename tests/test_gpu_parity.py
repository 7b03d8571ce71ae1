"""GPU parity: the HIP path (through the C ABI, via the host classes) against the reference's
golden outputs and against the CPU oracle on seeded inputs.

Tolerance.  The device computes in fp32 with fp32 MFMA accumulation, the reference in float64.
The bar written here is  |s_gpu - s_ref| <= 1e-4 * max(|s_ref|, 1)  per score: 1e-4 relative for
every score of magnitude >= 1 and 1e-4 absolute below that (fp32 accumulation over K terms cannot
give a relative bound for scores that cancel to ~0).  Decision metrics (EER, minDCF) must be
identical."""
import numpy as np
import pytest

from conftest import model_from
from oracle import plda_ref as O

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def assert_scores(got, ref, rtol=RTOL):
    got, ref = np.asarray(got, float), np.asarray(ref, float)
    assert got.shape == ref.shape
    assert got.dtype == np.float64
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)
    assert np.nanmax(err) <= rtol if err.size else True, float(np.nanmax(err))


def make(kind, m):
    from hyperion_amd.pdfs.plda import FRPLDA, PLDA, SPLDA

    if kind == "splda":
        return SPLDA(mu=m["mu"], V=m["V"], W=m["W"])
    if kind == "frplda":
        return FRPLDA(mu=m["mu"], B=m["B"], W=m["W"])
    return PLDA(mu=m["mu"], V=m["V"], U=m["U"], D=m["D"])


@pytest.mark.parametrize("kind", ["splda", "frplda", "plda"])
@pytest.mark.parametrize("tag", ["a", "b"])
def test_llr_1vs1_golden(golden, kind, tag):
    g = golden("llr_1vs1")
    m = model_from(g, "%s_%s_" % (kind, tag), kind)
    s = make(kind, m).llr_1vs1(g["x1_" + tag], g["x2_" + tag])
    assert_scores(s, g["%s_%s_scores" % (kind, tag)])


def test_splda_medium_golden_and_metrics(golden):
    g = golden("splda_medium")
    m = model_from(g, "m_", "splda")
    s = make("splda", m).llr_1vs1(g["x1"], g["x2"])
    ref = g["scores"]
    assert_scores(s, ref)
    # rows/cols come from 32 classes x 8: x1 = even rows, x2 = odd rows of the same ordering
    cls1 = (np.arange(96) * 2) // 8
    cls2 = (np.arange(ref.shape[1]) * 2 + 1) // 8
    key = cls1[:, None] == cls2[None, :]
    assert O.compute_eer(s[key], s[~key]) == O.compute_eer(ref[key], ref[~key])
    assert O.compute_min_dcf(s[key], s[~key], 0.05)[0] == O.compute_min_dcf(ref[key], ref[~key], 0.05)[0]


def test_transforms_golden(golden):
    from hyperion_amd.transforms import LDA, PCA, CentWhiten, LNorm, TransformList

    g = golden("transforms")
    x = g["x"]

    def chk(y, key, tol=2e-6):
        ref = g[key]
        assert y.dtype == np.float64 and y.shape == ref.shape
        assert np.max(np.abs(y - ref)) <= tol * max(1.0, np.max(np.abs(ref))), key

    chk(CentWhiten(mu=g["mu1"], T=g["T1"]).predict(x), "cw")
    chk(CentWhiten(mu=g["mu1"], T=None).predict(x), "cw_noT")
    chk(CentWhiten(mu=None, T=g["T1"]).predict(x), "cw_nomu")
    chk(LNorm(mu=g["mu1"], T=g["T1"]).predict(x), "lnorm")
    chk(LNorm().predict(x), "lnorm_plain")
    chk(LDA(mu=g["mu3"], T=g["T3"]).predict(x), "lda")
    chk(PCA(mu=g["mu3"], T=g["T3"]).predict(x), "pca")
    chk(PCA(mu=None, T=g["T3"]).predict(x), "pca_nomu")
    chk(TransformList([LDA(mu=g["mu3"], T=g["T3"], name="lda"), LNorm(name="ln")]).predict(x),
        "chain_lda_lnorm")
    chk(TransformList([CentWhiten(mu=g["mu1"], T=g["T1"], name="cw"), LNorm(name="ln"),
                       LDA(mu=g["mu2"], T=g["T2"], name="lda")]).predict(x), "chain_cw_lnorm_lda")
    chk(TransformList([PCA(mu=g["mu1"], T=g["T1"], name="pca"), LNorm(mu=g["mu2"], T=None, name="l1"),
                       LDA(mu=g["mu2"], T=g["T2"], name="lda"), LNorm(name="l2")]).predict(x),
        "chain_two_norms")
    chk(LNorm().predict(g["x_zero_row"]), "lnorm_zero_row")
    # 1-D T: the reference raises NameError here; the element-wise product is the evident intent
    d = np.linspace(0.5, 2.0, x.shape[1])
    y = CentWhiten(mu=g["mu1"], T=d).predict(x)
    assert np.max(np.abs(y - (x - g["mu1"]) * d)) <= 2e-6 * np.max(np.abs(y))
    # the input is not modified
    x0 = x.copy()
    LNorm().predict(x)
    assert np.array_equal(x, x0)


@pytest.mark.parametrize("kind", ["splda", "frplda", "plda"])
@pytest.mark.parametrize("method", ["book", "vavg", "vavg-lnorm", "savg"])
def test_nvsm_golden(golden, kind, method):
    g = golden("llr_NvsM")
    model = make(kind, model_from(g, kind + "_", kind))
    key = method.replace("-", "_")
    s = model.llr_NvsM(g["x1"], g["x2"], ids1=g["ids1"], ids2=g["ids2"], method=method)
    assert_scores(s, g["%s_NvsM_%s" % (kind, key)])
    s = model.llr_Nvs1(g["x1"], g["x2"], ids1=g["ids1"], method=method)
    assert_scores(s, g["%s_Nvs1_%s" % (kind, key)])


@pytest.mark.parametrize("kind", ["splda", "frplda"])
def test_nvsm_from_stats(golden, kind):
    """ids=None: the first arguments already are (N, F, S) statistics; F must not be modified."""
    g = golden("llr_NvsM")
    model = make(kind, model_from(g, kind + "_", kind))
    N1, F1, _ = O.compute_stats_hard(g["x1"], g["ids1"])
    N2, F2, _ = O.compute_stats_hard(g["x2"], g["ids2"])
    F1c = F1.copy()
    for method in ["book", "vavg", "vavg-lnorm"]:
        s = model.llr_NvsM((N1, F1, None), (N2, F2, None), method=method)
        assert_scores(s, g["%s_NvsM_%s" % (kind, method.replace("-", "_"))])
    assert np.array_equal(F1, F1c)


def test_book_large_counts_and_offset_mean():
    """By-the-book scoring of classes with about a thousand segments and a model mean far from the
    origin (|mu| = 40 x the spread): the centring F - N mu is done in float64 and the per-count maps
    are applied in float64 as in the reference (splda.py:326-375).  With a thousand segments a side
    |q| reaches several thousand while the score is of order ten, which an fp32 packed row cannot
    carry to 1e-4 of the score; host statistics and host vectors therefore go through the float64
    refinement pass (every score of such a block is re-evaluated from float64 packed rows) and meet
    1e-4 of max(|score|, 1); fp32 device statistics keep the fp32 rows and 1e-3.
    Against the oracle (the reference's test data hold no such case: parity pinned by the oracle only)."""
    import torch

    import synth

    m = synth.make_splda(48, 10)
    m = dict(m, mu=np.full(48, 40.0))
    rng = np.random.RandomState(8)
    n1, n2 = 6, 9
    cnt1 = rng.randint(900, 1100, n1)
    cnt2 = np.array([1, 1, 2, 1, 3, 1, 1, 2, 1])
    ids1, ids2 = np.repeat(np.arange(n1), cnt1), np.repeat(np.arange(n2), cnt2)
    x1 = synth.sample_splda(m, n1 + n2, 1, seed=9)[:n1][ids1] + 0.7 * rng.randn(len(ids1), 48)
    x2 = synth.sample_splda(m, n1 + n2, 1, seed=9)[n1:][ids2] + 0.7 * rng.randn(len(ids2), 48)
    N1, F1, _ = O.compute_stats_hard(x1, ids1)
    N2, F2, _ = O.compute_stats_hard(x2, ids2)
    ref = O.llr_NvsM_book(m, (N1.copy(), F1.copy(), None), (N2.copy(), F2.copy(), None))
    model = make("splda", m)
    F1c, F2c = F1.copy(), F2.copy()
    s = model.llr_NvsM((N1, F1, None), (N2, F2, None), method="book")
    assert_scores(s, ref, rtol=1e-4)
    assert np.array_equal(F1, F1c) and np.array_equal(F2, F2c)
    # from the vectors themselves (float64 class sums on the device), both entry points
    assert_scores(model.llr_NvsM(x1, x2, ids1=ids1, ids2=ids2, method="book"), ref, rtol=1e-4)
    ref1 = O.llr_NvsM_book(m, (N1.copy(), F1.copy(), None), (np.ones(len(x2)), x2.copy(), None))
    assert_scores(model.llr_Nvs1(x1, x2, ids1=ids1, method="book"), ref1, rtol=1e-4)
    # without the pass: the fp32 rows alone (what device fp32 statistics get)
    s32 = model.llr_NvsM_book((N1, F1, None), (N2, F2, None), refine=False)
    assert_scores(s32, ref, rtol=1e-3)
    sd = model.llr_NvsM_book((N1, torch.from_numpy(F1).float().cuda(), None),
                             (N2, torch.from_numpy(F2).float().cuda(), None))
    assert isinstance(sd, torch.Tensor) and sd.dtype == torch.float32
    with pytest.raises(ValueError):
        model.llr_NvsM_book((N1, torch.from_numpy(F1).float().cuda(), None), (N2, F2, None), refine=True)


def test_compute_stats_hard(golden):
    from hyperion_amd.pdfs.plda import SPLDA

    g = golden("llr_NvsM")
    N, F, S = SPLDA.compute_stats_hard(g["stats_x"], g["stats_ids"])
    assert np.array_equal(N, g["stats_N"])
    assert np.max(np.abs(F - g["stats_F"])) <= 1e-6 * np.max(np.abs(g["stats_F"]))
    assert N[2] == 0 and np.all(F[2] == 0)  # a class id that never occurs


def test_missing_class_gives_nan(golden):
    """An enrolment id that never occurs: N = 0 and the class mean is 0/0.  The reference then
    fails inside scipy (solve_triangular rejects non-finite input with ValueError); this
    implementation returns a NaN row for that class and correct scores for all others."""
    g = golden("llr_NvsM")
    m = model_from(g, "splda_", "splda")
    ids1 = g["ids1"].copy()
    ids1[ids1 == 2] = 8  # class 2 disappears, class 8 appears
    with pytest.raises(ValueError), np.errstate(all="ignore"):
        O.llr_NvsM(m, g["x1"], g["x2"], ids1, g["ids2"], "vavg")
    s = make("splda", m).llr_NvsM(g["x1"], g["x2"], ids1=ids1, ids2=g["ids2"], method="vavg")
    assert s.shape[0] == 9 and np.all(np.isnan(s[2]))
    # the other classes: same as scoring with contiguous ids (class 8 renamed back to 2)
    ref = O.llr_NvsM(m, g["x1"], g["x2"], g["ids1"], g["ids2"], "vavg")
    assert_scores(s[[0, 1, 3, 4, 5, 6, 7]], ref[[0, 1, 3, 4, 5, 6, 7]])
    assert_scores(s[8:9], ref[2:3])


def test_score_norm_golden(golden):
    from hyperion_amd.score_norm import AdaptSNorm, SNorm, TNorm, ZNorm

    g = golden("score_norm")
    sc, ct, ec = g["scores"], g["coh_test"], g["enr_coh"]
    keep = [a.copy() for a in (sc, ct, ec)]

    def chk(y, key, tol=2e-5):
        ref = g[key]
        assert y.dtype == np.float64 and y.shape == ref.shape
        err = np.abs(y - ref) / np.maximum(np.abs(ref), 1.0)
        assert err.max() <= tol, (key, err.max())

    chk(AdaptSNorm(nbest=10).predict(sc, ct, ec), "asnorm_n10")
    chk(AdaptSNorm(nbest=20, nbest_discard=3).predict(sc, ct, ec), "asnorm_n20d3")
    chk(AdaptSNorm(nbest=100).predict(sc, ct, ec), "asnorm_clamp")
    chk(AdaptSNorm(nbest=100, nbest_discard=60).predict(sc, ct, ec), "asnorm_clampd")
    chk(SNorm().predict(sc, ct, ec), "snorm")
    chk(SNorm().predict(sc, ct, ec, g["mask_coh_test"], g["mask_enr_coh"]), "snorm_masked")
    chk(TNorm().predict(sc, ct), "tnorm")
    chk(ZNorm().predict(sc, ec), "znorm")
    # constant cohort: std 0 -> floor 1e-5, so the output is 1e5 * (s - 1.5): the fp32 rounding of
    # the input score s (6e-8 relative) is amplified wherever s is close to the cohort mean
    chk(ZNorm().predict(sc, g["enr_coh_const"]), "znorm_floor", tol=1e-3)
    chk(AdaptSNorm(nbest=10).predict(sc, ct, g["enr_coh_const"]), "asnorm_floor", tol=1e-3)
    # two-stage norms: the reference's predict bodies run on its own ZNorm / TNorm objects (its
    # ZTNorm / TZNorm classes cannot be constructed)
    from hyperion_amd.score_norm import TZNorm, ZTNorm

    cc, masks = g["coh_coh"], (g["mask_coh_test"], g["mask_enr_coh"], g["mask_coh_coh"])
    chk(ZTNorm().predict(sc, ct, ec, cc), "ztnorm")
    chk(TZNorm().predict(sc, ct, ec, cc), "tznorm")
    chk(ZTNorm().predict(sc, ct, ec, cc, *masks), "ztnorm_masked")
    chk(TZNorm().predict(sc, ct, ec, cc, *masks), "tznorm_masked")
    for a, b in zip((sc, ct, ec), keep):
        assert np.array_equal(a, b)  # caller arrays untouched


def test_adapt_s_norm_masked():
    """Masked cohort entries count as zero when ranking and are excluded from the moments.  The
    cohort scores are positive here so that the zeros rank last and no tie straddles the top-k
    boundary (numpy's order among ties is unspecified)."""
    from hyperion_amd.score_norm import AdaptSNorm

    rng = np.random.RandomState(5)
    ne, nt, nc = 9, 14, 50
    sc = rng.randn(ne, nt) * 5
    ct = rng.rand(nc, nt) * 4 + 2
    ec = rng.rand(ne, nc) * 4 + 2
    mct = rng.rand(nc, nt) > 0.3
    mec = rng.rand(ne, nc) > 0.3
    ref = O.adapt_s_norm(sc, ct, ec, 8, 0, mct, mec)
    got = AdaptSNorm(nbest=8).predict(sc, ct, ec, mct, mec)
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)
    assert err.max() <= 2e-5


def test_adapt_s_norm_vs_oracle_large():
    """A cohort large enough to exercise several radix passes and many lists per row."""
    from hyperion_amd.score_norm import AdaptSNorm

    rng = np.random.RandomState(9)
    ne, nt, nc = 70, 333, 3000
    sc = rng.randn(ne, nt) * 6 - 5
    ct = (rng.randn(nc, nt) * 8 - 14).astype(np.float32).astype(float)
    ec = (rng.randn(ne, nc) * 8 - 14).astype(np.float32).astype(float)
    for nbest, disc in [(300, 0), (200, 17)]:
        ref = O.adapt_s_norm(sc, ct, ec, nbest, disc)
        for refine in (False, None):  # fp32 throughout (the fused kernels) / float64 where fp32 is not enough
            got = AdaptSNorm(nbest=nbest, nbest_discard=disc, refine=refine).predict(sc, ct, ec)
            err = np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)
            assert err.max() <= 2e-5, err.max()


def _refine_cases(ne, nt, nc, rng):
    yield rng.randn(ne, nt) * 6 - 5, rng.randn(nc, nt) * 8 - 14, rng.randn(ne, nc) * 8 - 14
    yield rng.randn(ne, nt) + 60, rng.randn(nc, nt) * 0.5 + 60, rng.randn(ne, nc) * 0.5 + 60  # narrow, far from zero
    yield rng.standard_t(3, (ne, nt)) * 4, rng.standard_t(3, (nc, nt)) * 4 - 3, rng.standard_t(3, (ne, nc)) * 4 - 3
    a, b, c = rng.randn(ne, 1) * 3, rng.randn(1, nt) * 3, rng.randn(nc) * 3  # a common factor per side, as PLDA scores have
    yield (a + b + rng.randn(ne, nt) * 5 - 20, c[:, None] + b + rng.randn(nc, nt) * 5 - 20,
           a + c[None, :] + rng.randn(ne, nc) * 5 - 20)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(60, 700, 2500, 300, 0), (33, 130, 22_000, 150, 4)])
def test_adapt_s_norm_strict_relative(shape):
    """float64 host arrays in, every normalised score within 1e-4 RELATIVE of the reference's float64
    value, the ones that cancel to nearly zero included: entries whose fp32 error estimate is too large for
    their magnitude are recomputed from the float64 matrices (hypmi_asnorm_refine).  Without the
    refinement the same calls miss that bar on some entries, which is what makes the check meaningful.
    The second shape takes the cohort in two segments."""
    from hyperion_amd.score_norm import AdaptSNorm

    ne, nt, nc, nbest, disc = shape
    rng = np.random.RandomState(31)
    missed = 0
    for sc, ct, ec in _refine_cases(ne, nt, nc, rng):
        ref = O.adapt_s_norm(sc, ct, ec, nbest, disc)
        norm = AdaptSNorm(nbest=nbest, nbest_discard=disc)
        got = norm.predict(sc, ct, ec)
        rel = np.abs(got - ref) / np.abs(ref)
        assert rel.max() <= 1e-4, float(rel.max())
        assert 0 < norm.last_refined < ref.size // 2, norm.last_refined
        plain = AdaptSNorm(nbest=nbest, nbest_discard=disc, refine=False)
        rel32 = np.abs(plain.predict(sc, ct, ec) - ref) / np.abs(ref)
        assert plain.last_refined is None
        missed += int((rel32 > 1e-4).sum())
    assert missed > 0


@pytest.mark.gpu
def test_adapt_s_norm_refine_needs_float64_host_arrays():
    import torch

    from hyperion_amd.score_norm import AdaptSNorm

    rng = np.random.RandomState(1)
    sc, ct, ec = rng.randn(5, 7), rng.randn(40, 7), rng.randn(5, 40)
    norm = AdaptSNorm(nbest=10, refine=True)
    with pytest.raises(ValueError):
        norm.predict(sc.astype(np.float32), ct, ec)
    with pytest.raises(ValueError):
        norm.predict(torch.from_numpy(sc).cuda(), ct, ec)
    with pytest.raises(ValueError):
        norm.predict(sc, ct, ec, mask_coh_test=rng.rand(40, 7) > 0.1)
    assert np.isfinite(norm.predict(sc, ct, ec)).all()


@pytest.mark.parametrize("shape", [(37, 91, 640, 50, 0), (130, 70, 2100, 300, 3), (1100, 1200, 1030, 40, 0)])
def test_adapt_s_norm_predict_every_input_kind(shape):
    """float64 / float32, host arrays / device tensors, either mask alone, both or none (the last shape is
    large enough for the staged upload of wide host arrays): the same result as the oracle, a float64 array
    for host input, and the caller's arrays left as they were (the reference zeroes them in place)."""
    import itertools

    import torch

    from hyperion_amd.score_norm import AdaptSNorm

    ne, nt, nc, nbest, disc = shape
    rng = np.random.RandomState(5)
    S, CT, EC = rng.randn(ne, nt) * 3, rng.randn(nc, nt) * 3 + 1, rng.randn(ne, nc) * 3 - 1
    mct, mec = rng.rand(nc, nt) > 0.2, rng.rand(ne, nc) > 0.2
    for dt, use_ct, use_ec in itertools.product((np.float64, np.float32), (False, True), (False, True)):
        host = [a.astype(dt) for a in (S, CT, EC)]
        m1, m2 = (mct if use_ct else None), (mec if use_ec else None)
        ref = O.adapt_s_norm(*[a.astype(np.float64) for a in host], nbest, disc, m1, m2)
        for dev in (False, True):
            args = [torch.from_numpy(a).cuda() for a in host] if dev else [a.copy() for a in host]
            got = AdaptSNorm(nbest=nbest, nbest_discard=disc).predict(*args, mask_coh_test=m1, mask_enr_coh=m2)
            for a, b in zip(args, host):
                assert np.array_equal(a.cpu().numpy() if dev else a, b), "the caller's array was modified"
            if dev:
                got = got.cpu().numpy()
            else:
                assert got.dtype == np.float64
            err = np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)
            assert err.max() <= 2e-5, (dt.__name__, use_ct, use_ec, dev, err.max())


def test_topk_ties_and_duplicates():
    """Exact ties: the selected set must have the right size and every selected value must be >=
    every unselected one; ties at the boundary go to the lower index."""
    import torch

    from hyperion_amd.score_norm.adapt_s_norm import lists_to_numpy, topk_rows

    rng = np.random.RandomState(3)
    A = rng.randint(0, 6, size=(37, 500)).astype(np.float32)  # heavy ties
    A[5] = 1.0
    for k, disc in [(50, 0), (17, 9), (500, 0), (1, 499)]:
        idx = lists_to_numpy(topk_rows(torch.from_numpy(A).cuda(), k, disc), k)  # [rows, k]
        for r in range(A.shape[0]):
            order = np.lexsort((np.arange(500), -A[r]))  # descending value, ascending index
            want = np.sort(order[disc:disc + k])
            assert np.array_equal(np.sort(idx[r]), want), (k, disc, r)


@pytest.mark.parametrize("n1,n2", [(0, 5), (5, 0), (1, 1), (1, 33), (257, 31), (129, 65), (300, 1000)])
def test_ragged_shapes(n1, n2):
    import synth

    m = synth.make_splda(48, 10)
    x = synth.length_norm(synth.sample_splda(m, 200, 8, seed=3))
    x1, x2 = x[:n1], x[1600 - n2:1600]
    s = make("splda", m).llr_1vs1(x1, x2)
    assert s.shape == (n1, n2)
    if n1 and n2:
        assert_scores(s, O.llr_1vs1(m, x1, x2))


@pytest.mark.parametrize("d,K", [(30, 5), (64, 62), (150, 100), (200, 150), (320, 180), (330, 250)])
def test_packed_widths(d, K):
    """Every register-stationary width and the K-loop kernel (wide models)."""
    import synth

    m = synth.make_splda(d, K)
    x = synth.length_norm(synth.sample_splda(m, 40, 6, seed=2))
    s = make("splda", m).llr_1vs1(x[:150], x[80:])
    assert_scores(s, O.llr_1vs1(m, x[:150], x[80:]))


def test_frplda_wide():
    """FRPLDA has K = x_dim: 400 dims exceeds the register-stationary widths."""
    import synth

    m = synth.make_frplda(400, 60)
    xs = synth.length_norm(synth.sample_splda(synth.make_splda(400, 60), 20, 5, seed=4))
    s = make("frplda", m).llr_1vs1(xs[:60], xs[40:])
    assert_scores(s, O.llr_1vs1(m, xs[:60], xs[40:]))


def test_fused_preproc_equals_two_step(golden):
    """llr_1vs1(x, preproc=chain) == llr_1vs1(chain.predict(x)) == oracle."""
    import synth
    from hyperion_amd.transforms import LDA, CentWhiten, LNorm, TransformList

    chain = synth.chain_cfg2(d=96, lda_dim=40, seed=3)
    m = synth.make_splda(40, 12)
    x1, _ = synth.raw_xvectors(70, d=96, seed=21, num_centers=30)
    x2, _ = synth.raw_xvectors(90, d=96, seed=22, num_centers=30)
    tl = TransformList([CentWhiten(mu=chain[0][1], T=chain[0][2], name="cw"), LNorm(name="ln"),
                        LDA(mu=chain[2][1], T=chain[2][2], name="lda")])
    model = make("splda", m)
    ref = O.llr_1vs1(m, O.chain_predict(x1, chain), O.chain_predict(x2, chain))
    assert_scores(model.llr_1vs1(x1, x2, preproc=tl), ref)
    assert_scores(model.llr_1vs1(tl.predict(x1), tl.predict(x2)), ref)


def test_voxceleb_size_full_matrix():
    """Config 1: 4874 x 4874 SPLDA scores, 512-d, y_dim 150, against the oracle (full matrix),
    plus EER / minDCF identity on the sub-sampled trial lists."""
    import synth

    m, x, cls = synth.voxceleb_like()
    s = make("splda", m).llr_1vs1(x, x)
    ref = O.llr_1vs1(m, x, x)
    err = np.abs(s - ref) / np.maximum(np.abs(ref), 1.0)
    assert err.max() <= RTOL, err.max()
    key = cls[:, None] == cls[None, :]
    tar_g, non_g = s[key][::20], s[~key][::100]
    tar_r, non_r = ref[key][::20], ref[~key][::100]
    assert O.compute_eer(tar_g, non_g) == O.compute_eer(tar_r, non_r)
    assert O.compute_min_dcf(tar_g, non_g, 0.05)[0] == O.compute_min_dcf(tar_r, non_r, 0.05)[0]
    # size-independent property: the matrix of a set against itself is symmetric up to rounding
    assert np.max(np.abs(s - s.T)) <= 2e-4


def test_backend_snorm_pipeline_chunked_and_sharded():
    """Device-resident chain -> SPLDA -> AS-norm pipeline against the oracle, with the test set
    cut into several chunks and into two shards (the multi-GPU partition, run back to back)."""
    import synth
    from hyperion_amd.backend import PLDABackend, shard_range
    from hyperion_amd.pdfs.plda import SPLDA
    from hyperion_amd.transforms import LDA, CentWhiten, LNorm, TransformList

    chain = synth.chain_cfg2(d=96, lda_dim=40, seed=3)
    m = synth.make_splda(40, 12)
    xe, _ = synth.raw_xvectors(50, d=96, seed=21, num_centers=30)
    xt, _ = synth.raw_xvectors(333, d=96, seed=22, num_centers=30)
    xc, _ = synth.raw_xvectors(400, d=96, seed=23, num_centers=200)
    tl = TransformList([CentWhiten(mu=chain[0][1], T=chain[0][2], name="cw"), LNorm(name="ln"),
                        LDA(mu=chain[2][1], T=chain[2][2], name="lda")])
    be = PLDABackend(SPLDA(mu=m["mu"], V=m["V"], W=m["W"]), tl, nbest=50, nbest_discard=2, test_chunk=100)
    ye, yt, yc = (O.chain_predict(a, chain) for a in (xe, xt, xc))
    ref = O.adapt_s_norm(O.llr_1vs1(m, ye, yt), O.llr_1vs1(m, yc, yt), O.llr_1vs1(m, ye, yc), 50, 2)
    coh = be.prepare_cohort(xe, xc)
    got = be.score_snorm(xt, coh).cpu().numpy().astype(float)
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)
    assert err.max() <= 1e-4, err.max()
    # two test shards, as two ranks would own them
    parts = []
    for r in range(2):
        lo, hi = shard_range(len(xt), r, 2)
        parts.append(be.score_snorm(xt[lo:hi], coh).cpu().numpy())
    # the members of a cohort list are summed in an order that depends on the position of the test in
    # its chunk (bank-class rotation of the top-k lists): shards agree to fp32 rounding, not bitwise
    assert np.max(np.abs(np.hstack(parts) - got) / np.maximum(np.abs(got), 1.0)) <= 5e-6
    # streaming reduction instead of a materialised matrix
    acc = []
    be.score_snorm(xt, coh, reduce=lambda c0, blk: acc.append((c0, float(blk.sum(dtype=__import__("torch").float64)))))
    assert [c for c, _ in acc] == [0, 100, 200, 300]
    assert abs(sum(v for _, v in acc) - got.sum()) <= 1e-3 * abs(got).sum()


@pytest.mark.gpu
def test_backend_snorm_enrol_rows_walked_in_blocks():
    """With more enrolment rows than the tile budget allows per chunk of tests, the rows are walked in
    blocks (row ranges of the packed rows, of the enrol-vs-cohort panel and of the per-enrolment
    lists): same result as one tile, for the matrix and for the streamed blocks."""
    import torch

    import synth
    from hyperion_amd.backend import PLDABackend
    from hyperion_amd.metrics import TrialHistogram
    from hyperion_amd.pdfs.plda import SPLDA

    m = synth.make_splda(40, 12)
    xe, ce = synth.raw_xvectors(2500, d=40, seed=21, num_centers=30)
    xt, ct = synth.raw_xvectors(700, d=40, seed=22, num_centers=30)
    xc, _ = synth.raw_xvectors(400, d=40, seed=23, num_centers=200)
    be = PLDABackend(SPLDA(mu=m["mu"], V=m["V"], W=m["W"]), None, nbest=50, test_chunk=512)
    coh = be.prepare_cohort(xe, xc)
    whole = be.score_snorm(xt, coh).cpu().numpy()
    be.tile_bytes = 24 * 512 * 1024  # 1024 enrolment rows per tile: blocks of 1024, 1024 and 452
    tiled = be.score_snorm(xt, coh).cpu().numpy()
    assert np.array_equal(tiled, whole)
    seen = []
    h = TrialHistogram(-60.0, 60.0, 512)
    sink = h.sink(ce, ct)

    def reduce(c0, blk, r0):
        seen.append((c0, r0, tuple(blk.shape)))
        assert torch.equal(blk, torch.from_numpy(whole[r0:r0 + blk.shape[0], c0:c0 + blk.shape[1]]).cuda())
        sink(c0, blk, r0)

    assert be.score_snorm(xt, coh, reduce=reduce) is None
    assert seen == [(c0, r0, (min(1024, 2500 - r0), min(512, 700 - c0))) for c0 in (0, 512) for r0 in (0, 1024, 2048)]
    ref = TrialHistogram(-60.0, 60.0, 512).add(torch.from_numpy(whole).cuda(), ce, ct)
    assert all(np.array_equal(a, b) for a, b in zip(h.counts(), ref.counts()))


def test_strict_relative_parity_with_refinement():
    """With the fp64 refinement pass (default for numpy inputs) every score of the VoxCeleb-size
    matrix matches the float64 oracle to 1e-4 RELATIVE, including those that cancel to ~0; without
    it only the mixed bound holds.  Also checks the fraction of refined entries is small."""
    import torch

    import synth
    from hyperion_amd import engine
    from hyperion_amd.engine import SIDE_ENROL, SIDE_TEST

    m, x, cls = synth.voxceleb_like()
    model = make("splda", m)
    ref = O.llr_1vs1(m, x, x)
    s = model.llr_1vs1(x, x)  # numpy in -> refined
    rel = np.abs(s - ref) / np.abs(ref)
    assert rel.max() <= 1e-4, rel.max()
    raw = model.llr_1vs1(x, x, refine=False)
    rel_raw = np.abs(raw - ref) / np.abs(ref)
    assert (rel_raw > 1e-4).mean() > 1e-5  # the fp32 path alone does miss the strict bound somewhere
    assert np.max(np.abs(raw - ref) / np.maximum(np.abs(ref), 1.0)) <= 1e-4
    # how many entries the refinement touches
    plan = model.score_plan()
    xd = engine.upload(x)
    S = engine.score(plan.prologue(xd, SIDE_ENROL), plan.prologue(xd, SIDE_TEST))
    D1, b1 = plan.prologue_f64(engine.upload64(x), SIDE_ENROL)
    D2, b2 = plan.prologue_f64(engine.upload64(x), SIDE_TEST)
    n = engine.refine_scores(S, D1, b1, D2, b2, rel_tol=0.5e-4, count=True)
    assert 0 < n < 0.05 * S.numel(), n
    # the fp64 packed rows themselves: g.g + q1 + q2(+c) reproduces the oracle to 1e-9
    s64 = (D1[:50, :-2] @ D2[:60, :-2].T + D1[:50, -2:-1] + D2[:60, -2].unsqueeze(0)).cpu().numpy()
    assert np.max(np.abs(s64 - ref[:50, :60])) <= 1e-9 * np.max(np.abs(ref))
    assert isinstance(model.llr_1vs1(xd, xd), torch.Tensor)  # device in -> device out, unrefined


@pytest.mark.parametrize("kind", ["splda", "frplda", "plda"])
def test_refinement_golden_strict(golden, kind):
    """Golden cases under the strict relative bound (numpy path, refinement on), incl. the chain."""
    g = golden("llr_1vs1")
    for tag in "ab":
        m = model_from(g, "%s_%s_" % (kind, tag), kind)
        s = make(kind, m).llr_1vs1(g["x1_" + tag], g["x2_" + tag])
        ref = g["%s_%s_scores" % (kind, tag)]
        assert np.max(np.abs(s - ref) / np.abs(ref)) <= 1e-4


def test_topk_boundary_redecided_from_float64_matrix():
    """numpy AdaptSNorm path: two cohort scores that differ by 1e-9 collide in fp32 exactly at the
    top-k boundary; the member float64 argsort keeps must be kept (the fp32 tie rule alone would keep
    the other one, whose enrol-vs-cohort scores are very different)."""
    from hyperion_amd.score_norm import AdaptSNorm

    rng = np.random.RandomState(12)
    ne, nt, nc, k = 6, 40, 300, 25
    sc = rng.randn(ne, nt) * 5
    ct = rng.randn(nc, nt) * 4 - 6
    ec = rng.randn(ne, nc) * 4 - 6
    for t in range(nt):
        order = np.argsort(-ct[:, t])
        a, b = sorted((order[k - 1], order[k]))      # the two members around the boundary
        v = ct[order[k - 1], t]
        ct[a, t] = np.float32(v)                     # lower index: fp32 tie rule would keep it
        ct[b, t] = np.float64(np.float32(v)) + 1e-9  # float64 says this one ranks higher
    ec[:, ::2] += 40.0                               # make a wrong member visible in the statistics
    ref = O.adapt_s_norm(sc, ct, ec, k, 0)
    got = AdaptSNorm(nbest=k).predict(sc, ct, ec)
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)
    assert err.max() <= 2e-5, err.max()


def test_topk_boundary_redecided_from_packed_rows():
    """Device pipeline: a cohort of near-duplicate vectors puts pairs of fp32-indistinguishable
    scores at the selection boundary of most rows.  With the fp64 packed rows as exact source the
    selected sets equal those of float64 argsort in every row; without it they do not."""
    import synth
    from hyperion_amd import engine
    from hyperion_amd.engine import SIDE_ENROL, SIDE_TEST
    from hyperion_amd.score_norm.adapt_s_norm import lists_to_numpy, topk_rows

    m = synth.make_splda(64, 20)
    model = make("splda", m)
    rng = np.random.RandomState(1)
    base = synth.length_norm(synth.sample_splda(m, 500, 1, seed=5))
    xc = np.vstack([base, base * (1 + 2e-8 * rng.randn(500, 1)) + 1e-8 * rng.randn(500, 64)])
    xt = synth.length_norm(synth.sample_splda(m, 300, 1, seed=6))
    plan = model.score_plan()
    k, disc = 120, 7
    A = engine.score(plan.prologue(engine.upload(xt), SIDE_ENROL), plan.prologue(engine.upload(xc), SIDE_TEST))
    ref_scores = O.llr_1vs1(m, xt, xc)
    order = np.lexsort((np.arange(1000)[None, :].repeat(300, 0), -ref_scores), axis=1)
    want = np.sort(order[:, disc:disc + k], axis=1)
    Gt, bt = plan.prologue_f64(engine.upload64(xt), SIDE_ENROL)
    Gc, bc = plan.prologue_f64(engine.upload64(xc), SIDE_TEST)
    exact = {"G_rows": Gt, "G_coh": Gc, "bnd_rows": bt, "coh_gmax": float(bc[:, 0].max()),
             "coh_qmax": float(bc[:, 1].max()), "count": True}
    got = np.sort(lists_to_numpy(topk_rows(A, k, disc, exact), k), axis=1)
    plain = np.sort(lists_to_numpy(topk_rows(A, k, disc), k), axis=1)
    bad_exact = int((got != want).any(axis=1).sum())
    bad_plain = int((plain != want).any(axis=1).sum())
    assert bad_plain > 10, bad_plain      # the test bites: fp32 alone picks wrong members
    assert bad_exact == 0, (bad_exact, exact["counters"])
    assert exact["counters"][0] > 0 and exact["counters"][1] == 0


def _decode_split(H, n, kp):
    """Split rows (int32 words) -> float64 [3, n, 16 * ks16] of the three bf16 parts.  Row layout:
    [HS slots: even 8-chunks][3 * ks16 slots: odd 8-chunks][pad], slot 3 * j + s of a half = part s
    of chunk pair j; HS = 3 * ks16 rounded up to 16, ks16 = 2 * ceil(kp / 32)."""
    ks16 = 2 * ((kp + 31) // 32)
    hs = (3 * ks16 + 15) // 16 * 16
    row_slots = (hs + 3 * ks16) | 1
    assert H.numel() == n * row_slots * 4 + 256
    rows = H[:n * row_slots * 4].cpu().numpy().view(np.uint32).reshape(n, row_slots, 4)
    out = np.zeros((3, n, 16 * ks16), dtype=np.float64)
    used = np.zeros(row_slots, dtype=bool)
    for j in range(ks16):
        for s in range(3):
            for h in range(2):
                slot = h * hs + 3 * j + s
                used[slot] = True
                w = rows[:, slot, :]                                        # 4 words = 8 bf16
                bits = np.stack([w & 0xffff, w >> 16], axis=-1).reshape(n, 8)
                vals = (bits.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
                out[s, :, 16 * j + 8 * h:16 * j + 8 * h + 8] = vals
    assert np.all(rows[:, ~used, :] == 0)                                   # padding slots are cleared
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("kp", [16, 64, 152, 232, 304])
def test_split_rows_are_exact(kp):
    """x0 + x1 + x2 reproduces every fp32 value bit for bit (the split kernel's only approximation is
    then the three dropped partial products); non-finite values keep their class.  (Below 2^-110 the
    low parts fall into the bf16 subnormals and lose bits; packed rows never hold such values.)"""
    import torch

    from hyperion_amd import engine

    rng = np.random.RandomState(kp)
    n = 300
    x = (rng.randn(n, kp) * np.exp(rng.uniform(-30, 30, size=(n, kp)))).astype(np.float32)
    x[0, :8] = [0.0, -0.0, 1.0, -1.0, 3.3895314e38, -3.4028235e38, 1e-30, -2.5e-31]
    G = torch.from_numpy(x).cuda()
    H = engine.split_rows(G, "triple")
    parts = _decode_split(H.H, n, kp)
    back = parts.sum(axis=0)[:, :kp].astype(np.float32)
    bad = np.argwhere(back != x)
    assert bad.size == 0, "first mismatches: %s" % [(tuple(i), x[tuple(i)], back[tuple(i)]) for i in bad[:5]]
    assert np.all(parts[:, :, kp:] == 0)
    # magnitudes fall by 2^-8 per part (round to nearest: half an 8-bit ulp; one bit of slack for
    # the largest fp32, whose leading part is kept finite by rounding down)
    assert np.all(np.abs(parts[1]) <= np.abs(parts[0]) * 2.0 ** -7)
    assert np.all(np.abs(parts[2]) <= np.abs(parts[0]) * 2.0 ** -15)
    y = np.ones((4, kp), dtype=np.float32)
    y[1, 3] = np.inf
    y[2, 5] = np.nan
    y[3, 0] = -np.inf
    p = _decode_split(engine.split_rows(torch.from_numpy(y).cuda(), "triple").H, 4, kp)
    assert np.isposinf(p[0, 1, 3]) and np.isnan(p[0, 2, 5]) and np.isneginf(p[0, 3, 0])
    assert np.all(p[1:] == 0)


def _decode_pair(H, n, kp):
    """fp16 pair rows -> (parts float64 [2, n, 16 * ks16] in scaled units, inverse scale [n]).  Layout
    as for triples with two parts per chunk; the last slot of a row holds the inverse scale."""
    ks16 = 2 * ((kp + 31) // 32)
    hs = (2 * ks16 + 15) // 16 * 16
    row_slots = (hs + 2 * ks16) | 1
    assert H.numel() == n * row_slots * 4 + 256
    rows = H[:n * row_slots * 4].cpu().numpy().view(np.uint32).reshape(n, row_slots, 4)
    out = np.zeros((2, n, 16 * ks16), dtype=np.float64)
    used = np.zeros(row_slots, dtype=bool)
    used[-1] = True
    for j in range(ks16):
        for s in range(2):
            for h in range(2):
                slot = h * hs + 2 * j + s
                used[slot] = True
                w = rows[:, slot, :]
                bits = np.stack([w & 0xffff, w >> 16], axis=-1).reshape(n, 8).astype(np.uint16)
                out[s, :, 16 * j + 8 * h:16 * j + 8 * h + 8] = bits.view(np.float16).astype(np.float64)
    assert np.all(rows[:, ~used, :] == 0) and np.all(rows[:, -1, 1:] == 0)
    return out, rows[:, -1, 0].copy().view(np.float32).astype(np.float64)


@pytest.mark.gpu
@pytest.mark.parametrize("kp", [16, 64, 152, 232, 304])
def test_pair_rows_represent_scaled_values(kp):
    """Every row is scaled by a power of two that puts its largest magnitude in [2^14, 2^15); x0 + x1
    is within 2^-22 of the scaled value (2^-25 absolute where the low part leaves the normal fp16
    range); the inverse scale sits in the last slot; non-finite values keep their class and do not
    set the scale; a row of zeros has scale one."""
    import torch

    from hyperion_amd import engine

    rng = np.random.RandomState(kp)
    n = 300
    x = (rng.randn(n, kp) * np.exp(rng.uniform(-4, 4, size=(n, kp)))).astype(np.float32)
    x *= np.exp2(rng.randint(-100, 100, size=(n, 1))).astype(np.float32)  # rows of very different size
    x[0] = 0.0
    x[1] = 0.25
    x[1, :4] = [np.inf, -np.inf, np.nan, 1.0]
    x[2, :] = 1e-42  # subnormal fp32
    parts, inv = _decode_pair(engine.split_rows(torch.from_numpy(x).cuda(), "pair").H, n, kp)
    fin = np.where(np.isfinite(x), np.abs(x), 0).max(axis=1).astype(np.float64)
    scaled_max = fin / inv
    big = fin > 2.0 ** -111  # below that the scale is capped so that its inverse stays a normal float
    assert np.all((scaled_max[big] >= 2.0 ** 14) & (scaled_max[big] < 2.0 ** 15)) and inv[2] == 2.0 ** -126
    assert inv[0] == 1.0 and np.all(np.log2(inv) == np.round(np.log2(inv)))
    want = x.astype(np.float64) / inv[:, None]
    got = parts.sum(axis=0)[:, :kp]
    ok = np.isfinite(x)
    with np.errstate(invalid="ignore"):
        err = np.abs(got - want)
    assert np.all(err[ok] <= np.maximum(2.0 ** -22 * np.abs(want[ok]), 2.0 ** -25))
    assert np.all(parts[:, :, kp:] == 0)
    assert np.isposinf(parts[0, 1, 0]) and np.isneginf(parts[0, 1, 1]) and np.isnan(parts[0, 1, 2])
    assert np.all(parts[1, 1, :3] == 0) and got[1, 3] == 2.0 ** 14


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["pair", "triple"])
@pytest.mark.parametrize("kp", [16, 32, 64, 96, 128, 152, 192, 232, 264, 304, 320])
@pytest.mark.parametrize("n1,n2", [(1, 1), (31, 33), (128, 2048), (333, 2100), (700, 4133)])
def test_split_score_matches_float64(kp, n1, n2, fmt):
    """Both 16-bit formats (scaled fp16 pairs, bf16 triples) against a float64 product of the same
    fp32 rows: the error stays within the fp32 bound 8 * 2^-24 * sum|a b| on every entry, as does the
    fp32 kernel's."""
    import torch

    from hyperion_amd import engine

    g = torch.Generator(device="cuda").manual_seed(kp * 7 + n1)
    G1 = torch.randn((n1, kp), device="cuda", generator=g) * 3.0
    G2 = torch.randn((n2, kp), device="cuda", generator=g) * 0.7
    G1[:, kp - 2] = torch.randn(n1, device="cuda", generator=g) * 500.0
    G1[:, kp - 1] = 1.0
    G2[:, kp - 2] = 1.0
    G2[:, kp - 1] = torch.randn(n2, device="cuda", generator=g) * 500.0
    ref = (G1.double() @ G2.double().T).cpu().numpy()
    bound = (G1.double().abs() @ G2.double().abs().T).cpu().numpy() * engine.ERR_COEFF
    S_split = engine.score(G1, G2, split=fmt).cpu().numpy()
    S_fp32 = engine.score(G1, G2, split=False).cpu().numpy()
    assert S_split.shape == (n1, n2)
    assert np.all(np.abs(S_split - ref) <= bound)
    assert np.all(np.abs(S_fp32 - ref) <= bound)


@pytest.mark.parametrize("fmt", ["pair", "triple", False])
def test_score_error_bound_and_bias_on_the_calibrated_model(fmt):
    """The same bound on the packed rows of the calibrated VoxCeleb-size model, whose scores are sums that
    cancel (mean |score| 13 against sum |a b| of about 40): every entry within 8 * 2^-24 * sum|a b| of the
    float64 product, and no one-sided error (a truncating accumulation would show as a mean signed error of
    the order of the unit 2^-24 * sum|a b|; measured -0.03 units for the pairs, -0.001 for the fp32 kernel)."""
    import synth
    import torch

    from hyperion_amd import engine
    from hyperion_amd.engine import SIDE_ENROL, SIDE_TEST

    m, x, cls = synth.voxceleb_like()
    plan = make("splda", m).score_plan(None)
    xd = engine.upload(x[:2400])
    G1, G2 = plan.prologue(xd, SIDE_ENROL), plan.prologue(xd, SIDE_TEST)
    ref = G1.double() @ G2.double().T
    unit = (G1.double().abs() @ G2.double().abs().T) * 2.0 ** -24
    e = (engine.score(G1, G2, split=fmt).double() - ref) / unit
    assert e.abs().max().item() <= engine.ERR_COEFF * 2.0 ** 24
    key = torch.from_numpy(cls[:2400, None] == cls[None, :2400]).cuda()
    for part in (e, e[key], e[~key]):
        assert abs(part.mean().item()) <= 0.1, part.mean().item()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["pair", "triple"])
def test_split_score_propagates_nan_rows(fmt):
    import torch

    from hyperion_amd import engine

    G1 = torch.randn((70, 32), device="cuda")
    G2 = torch.randn((90, 32), device="cuda")
    G1[5, 3] = float("nan")
    G2[7, 0] = float("nan")
    S = engine.score(G1, G2, split=fmt).cpu().numpy()
    assert np.all(np.isnan(S[5])) and np.all(np.isnan(S[:, 7]))
    mask = np.ones_like(S, dtype=bool)
    mask[5] = False
    mask[:, 7] = False
    assert np.all(np.isfinite(S[mask]))


@pytest.mark.gpu
def test_pair_score_rows_of_very_different_size():
    """Per-row scaling: rows whose magnitudes differ by 2^60 (and a row of zeros) score as accurately
    as rows of one size."""
    import torch

    from hyperion_amd import engine

    g = torch.Generator(device="cuda").manual_seed(3)
    G1 = torch.randn((257, 152), device="cuda", generator=g)
    G2 = torch.randn((300, 152), device="cuda", generator=g)
    G1 *= torch.exp2(torch.randint(-30, 30, (257, 1), device="cuda", generator=g).float())
    G2 *= torch.exp2(torch.randint(-30, 30, (300, 1), device="cuda", generator=g).float())
    G1[11] = 0.0
    ref = (G1.double() @ G2.double().T).cpu().numpy()
    bound = (G1.double().abs() @ G2.double().abs().T).cpu().numpy() * engine.ERR_COEFF
    S = engine.score(G1, G2, split="pair").cpu().numpy()
    assert np.all(np.abs(S - ref) <= bound) and np.all(S[11] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("n,nc,k,disc,shift", [
    (7, 50, 10, 0, 0.0), (7, 50, 10, 0, -100.0), (5, 50, 10, 3, 0.0), (9, 700, 33, 0, 0.0),
    (4, 1500, 100, 5, 0.0), (6, 5000, 300, 0, 0.0), (3, 20000, 300, 0, 0.0), (3, 20000, 300, 7, -3.0),
    (2, 40960, 1000, 0, 0.0), (3, 300, 300, 0, 0.0), (2, 45000, 64, 2, 1.0), (2, 53000, 300, 0, 0.0),
    (3, 65536, 300, 3, -2.0)])
def test_topk_rows_matches_stable_argsort(n, nc, k, disc, shift):
    """Every size class of the register-resident top-k (and the workspace path above 40 960) against a
    stable descending argsort: ranks discard .. discard+k-1, ties to the lower index (the order of
    the members inside a list is a layout matter, see test_topk_lists_are_ordered_by_bank_class); one
    case has a fifth of each row tied at one value."""
    import torch

    from hyperion_amd.score_norm.adapt_s_norm import lists_to_numpy, topk_rows

    rng = np.random.RandomState(nc + k)
    A = (rng.randn(n, nc) * 3 + shift).astype(np.float32)
    if nc == 700:
        A[:, ::5] = A[:, :1]
    idx = lists_to_numpy(topk_rows(torch.from_numpy(A).cuda(), k, disc), k)
    order = np.argsort(-A, axis=1, kind="stable")
    assert np.array_equal(np.sort(idx, axis=1), np.sort(order[:, disc:disc + k], axis=1))


@pytest.mark.gpu
@pytest.mark.parametrize("n,nc,k,disc,scale,shift", [
    (20000, 900, 50, 0, 3.0, 0.0), (4000, 20000, 300, 2, 5.0, -8.0), (3000, 12000, 300, 0, 0.5, 40.0),
    (3000, 20000, 300, 0, 25.0, -60.0)])
def test_topk_rows_many_rows(n, nc, k, disc, scale, shift):
    """Thousands of rows, so that every CU runs several workgroups of the selection at once (a handful of
    rows never has two of them share a CU, and an ordering fault between the waves of a workgroup only shows
    under that load): every row against a stable descending argsort, twice with identical lists.  The score
    distributions put the selection boundary at positive values, among negative ones, and in a narrow band
    far from zero (where most of the first steps are decided from the row's extreme values alone)."""
    import torch

    from hyperion_amd.score_norm.adapt_s_norm import lists_to_numpy, topk_rows

    rng = np.random.default_rng(n + nc)
    A = rng.standard_normal((n, nc), dtype=np.float32) * np.float32(scale) + np.float32(shift)
    Ad = torch.from_numpy(A).cuda()
    lists = topk_rows(Ad, k, disc)
    idx = np.sort(lists_to_numpy(lists, k), axis=1)
    again = topk_rows(Ad, k, disc)
    assert torch.equal(lists, again)
    part = np.argpartition(-A, (disc, disc + k - 1), axis=1)[:, disc:disc + k] if disc else \
        np.argpartition(-A, k - 1, axis=1)[:, :k]
    want = np.sort(part, axis=1)
    bad = np.nonzero((idx != want).any(axis=1))[0]
    for r in bad:  # argpartition picks any of the members tied at a boundary: settle those rows by a stable sort
        order = np.argsort(-A[r], kind="stable")
        assert np.array_equal(idx[r], np.sort(order[disc:disc + k])), int(r)


@pytest.mark.gpu
@pytest.mark.parametrize("nc,k,disc", [(900, 50, 0), (900, 40, 3), (5000, 300, 0), (20000, 300, 0), (20000, 300, 2),
                                       (30000, 300, 0)])
def test_topk_rows_sign_and_range_cases(nc, k, disc):
    """Rows whose scores are all positive, all negative, all of one exponent, of both signs with the boundary
    just above or just below zero, with signed zeros, infinities, tiny and huge magnitudes, and with fewer
    positive scores than the selection holds: the selection decides the steps of the sign bit and of shared
    exponent bits from the row's range alone, so every combination of ranges has to give the members a stable
    descending argsort gives (-0.0 below +0.0 as keys: no row here has both at a boundary).  A last row holds
    the all-ones bit pattern, the NaN whose order-preserving key is the one the selection reserves for "no
    element": the row must still select k distinct members inside the row."""
    import torch

    from hyperion_amd.score_norm.adapt_s_norm import lists_to_numpy, topk_rows

    rng = np.random.default_rng(nc + k + disc)
    z = rng.standard_normal((14, nc)).astype(np.float32)
    rows = [
        np.abs(z[0]) + 0.5,                         # all positive
        -np.abs(z[1]) - 0.5,                        # all negative
        1.0 + 0.5 * rng.random(nc, dtype=np.float32),   # one exponent
        -(1.0 + 0.5 * rng.random(nc, dtype=np.float32)),
        z[4] * 3 - 6.0,                             # few positives: boundary positive for small k only
        z[5] * 3 + 6.0,                             # few negatives
        z[6] * 1e-30,                               # tiny magnitudes of both signs
        z[7] * 1e30,
        np.where(z[8] > 1.5, z[8], np.float32(-0.0)).astype(np.float32),   # most of the row -0.0
        np.where(z[9] > 2.5, z[9], z[9] * 1e-20).astype(np.float32),
        z[10].copy(), z[11].copy(), z[12].copy(),
    ]
    rows[10][::97] = np.inf
    rows[10][5::89] = -np.inf
    npos = k + disc - 3                             # fewer positive scores than discard + k
    rows[11] = -np.abs(rows[11]) - 1e-3
    rows[11][rng.choice(nc, npos, replace=False)] = np.abs(z[13][:npos]) + 1e-3
    rows[12] = np.abs(rows[12]) + 1e-3
    rows[12][rng.choice(nc, nc - (k + disc + 3), replace=False)] *= -1   # a few more positives than discard + k
    A = np.stack(rows).astype(np.float32)
    last = rng.standard_normal(nc).astype(np.float32)
    last.view(np.uint32)[[3, nc // 2, nc - 1]] = 0xFFFFFFFF
    A = np.concatenate([A, last[None]], axis=0)
    idx = lists_to_numpy(topk_rows(torch.from_numpy(A).cuda(), k, disc), k)
    order = np.argsort(-A[:-1], axis=1, kind="stable")
    want = np.sort(order[:, disc:disc + k], axis=1)
    got = np.sort(idx[:-1], axis=1)
    for r in range(got.shape[0]):
        assert np.array_equal(got[r], want[r]), r
    assert len(set(idx[-1].tolist())) == k and idx[-1].max() < nc


@pytest.mark.gpu
def test_topk_lists_are_ordered_by_bank_class():
    """Lists of up to 2048 members leave the top-k kernel ordered by (index mod 32), inside a class by
    (index mod 256, index) -- the thread that held the member, then its item --, rotated by
    (row mod 32) / 32 of the list; the same call twice gives the same lists."""
    import torch

    from hyperion_amd.score_norm.adapt_s_norm import lists_to_numpy, topk_rows

    rng = np.random.RandomState(5)
    n, nc, k = 70, 20000, 300
    A = torch.from_numpy((rng.randn(n, nc) * 3).astype(np.float32)).cuda()
    idx = lists_to_numpy(topk_rows(A, k, 0), k)
    assert np.array_equal(idx, lists_to_numpy(topk_rows(A, k, 0), k))
    for r in range(n):
        members = np.sort(idx[r])
        by_class = members[np.lexsort((members, members % 256, members % 32))]
        assert np.array_equal(idx[r], np.roll(by_class, -((r % 32) * k // 32))), r


@pytest.mark.gpu
@pytest.mark.parametrize("masked", [False, True])
def test_adapt_s_norm_cohort_larger_than_lds(masked):
    """45 000 cohort members: the register-resident top-k (40 960) does not apply, and the cohort does
    not fit LDS two rows at a time; the workspace top-k with the segmented gather (unmasked) or the
    global-memory gather (masked) give the same result."""
    from hyperion_amd.score_norm import AdaptSNorm

    rng = np.random.RandomState(17)
    ne, nt, nc = 6, 9, 45_000
    sc = rng.randn(ne, nt) * 4
    ct = rng.randn(nc, nt) * 3 - 5
    ec = rng.randn(ne, nc) * 3 - 5
    kw = {}
    if masked:
        kw = {"mask_coh_test": rng.rand(nc, nt) > 0.1, "mask_enr_coh": rng.rand(ne, nc) > 0.1}
    ref = O.adapt_s_norm(sc, ct, ec, 200, 3, **kw)
    for refine in (False, None):
        got = AdaptSNorm(nbest=200, nbest_discard=3, refine=refine).predict(sc, ct, ec, **kw)
        err = np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)
        assert err.max() <= 2e-5, float(err.max())


@pytest.mark.gpu
def test_projected_model_self_scoring(golden):
    """The diarization consumer (hyperion/diarization/diar_ahc_plda.py:131-143): model projected
    onto a PCA basis, every frame scored against every frame."""
    from hyperion_amd.pdfs.plda import SPLDA

    g = golden("splda_project")
    p = SPLDA(mu=g["m_mu"], V=g["m_V"], W=g["m_W"]).project(g["T"], g["delta_mu"])
    s = p.llr_1vs1(g["x"], g["x"])
    err = np.abs(s - g["scores"]) / np.maximum(np.abs(g["scores"]), 1.0)
    assert err.max() <= 1e-4, err.max()
    assert np.max(np.abs(s - s.T)) <= 1e-4 * np.abs(s).max()


def test_scores_agree_with_the_trainable_layer_formula(golden):
    """Independent statement of the closed form (the reference's trainable SPLDA layer, W = U U^T,
    optional length normalisation inside the model: hyperion/torch/models/plda/splda.py:182-222)
    against the device scores, cross and self."""
    from hyperion_amd.pdfs.plda import SPLDA

    g = golden("llr_1vs1")
    m = model_from(g, "splda_b_", "splda")
    x1, x2 = g["x1_b"], g["x2_b"]
    ev, E = np.linalg.eigh(m["W"])
    U = E * np.sqrt(ev)
    model = SPLDA(mu=m["mu"], V=m["V"], W=m["W"])
    for a, b in ((x1, x2), (x1, None)):
        ref = O.splda_torch_layer_llr(a, b, m["mu"], m["V"], U)
        got = model.llr_1vs1(a, a if b is None else b)
        assert np.allclose(got, ref, rtol=1e-4, atol=0)
        ref = O.splda_torch_layer_llr(a, b, m["mu"], m["V"], U, lnorm=True)
        bb = a if b is None else b
        got = model.llr_NvsM(a, bb, ids1=np.arange(len(a)), ids2=np.arange(len(bb)), method="vavg-lnorm")
        assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)) <= 1e-4


@pytest.mark.gpu
def test_adapt_s_norm_warns_when_boundary_is_not_redecided():
    """More than 64 cohort scores that fp32 cannot tell from the top-k boundary: the float64 re-decision
    is given up for that row and the caller is told (the result then follows the fp32 tie order)."""
    import warnings

    from hyperion_amd.score_norm import AdaptSNorm

    rng = np.random.RandomState(2)
    ne, nt, nc = 4, 5, 400
    sc = rng.randn(ne, nt)
    ct = rng.randn(nc, nt)
    ec = rng.randn(ne, nc)
    ec[1, :200] = 3.0 + 1e-9 * rng.randn(200)  # 200 scores equal in fp32, distinct in float64, across the boundary
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        AdaptSNorm(nbest=100).predict(sc, ct, ec)
    assert any("could not be re-decided" in str(x.message) for x in w)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        AdaptSNorm(nbest=100).predict(sc, ct, rng.randn(ne, nc))
    assert not any("could not be re-decided" in str(x.message) for x in w)


@pytest.mark.gpu
@pytest.mark.parametrize("d,lda,K,n", [(512, 200, 150, 1000), (512, 200, 150, 64), (96, 40, 12, 333), (99, 37, 10, 1),
                                       (512, 200, 150, 262144 + 77), (96, 40, 12, 262144 + 5)])
def test_prologue_pairs_equals_prologue_then_pair_rows(d, lda, K, n):
    """The prologue kernel that writes fp16 pairs itself against the same plan's packed rows converted
    by the stand-alone kernel: the pairs, the inverse scales and the optional packed rows, bit for
    bit, on both sides of a trial, rows that do not fill the last workgroup included.  From 262 144
    rows in one launch the pair-writing entry point runs its 128-row tile, while hypmi_prologue (chunks
    of at most 65 536 rows) always runs the 64-row one: the two largest cases compare the two tile
    shapes, with a ragged last workgroup."""
    import torch

    import synth
    from hyperion_amd import _lib, engine
    from hyperion_amd.engine import SIDE_ENROL, SIDE_TEST
    from hyperion_amd.pdfs.plda import SPLDA
    from hyperion_amd.transforms import LDA, CentWhiten, LNorm, TransformList

    chain = synth.chain_cfg2(d=d, lda_dim=lda, seed=4)
    m = synth.make_splda(lda, K)
    tl = TransformList([CentWhiten(mu=chain[0][1], T=chain[0][2], name="cw"), LNorm(name="ln"),
                        LDA(mu=chain[2][1], T=chain[2][2], name="lda")])
    plan = SPLDA(mu=m["mu"], V=m["V"], W=m["W"]).score_plan(tl)
    assert _lib.load().hypmi_prologue_pairs_supported(plan._h) == 1
    if n < 100_000:
        x = engine.upload(np.random.RandomState(n).randn(n, d).astype(np.float32) * 2)
    else:
        x = engine.upload(torch.randn((n, d), device="cuda", generator=torch.Generator(device="cuda").manual_seed(n)) * 2)
    for side in (SIDE_ENROL, SIDE_TEST):
        G = plan.prologue(x, side)
        want = engine.split_rows(G, "pair")
        got, G2 = plan.prologue_split(x, side, keep_rows=True)
        nb = _lib.load().hypmi_pair_row_bytes(plan.kp) * n // 4
        assert torch.equal(got.H[:nb], want.H[:nb])
        assert torch.equal(G2, G)
        got_only = plan.prologue_split(x, side)
        assert torch.equal(got_only.H[:nb], want.H[:nb])
    both = plan.prologue_split_both(x)  # one pass, both packings
    for side, got in zip((SIDE_ENROL, SIDE_TEST), both):
        want = engine.split_rows(plan.prologue(x, side), "pair")
        assert torch.equal(got.H[:nb], want.H[:nb])


@pytest.mark.gpu
@pytest.mark.parametrize("rows,nc,k,lists", [
    (5, 30_000, 300, 70),      # more than two rows' worth of LDS: the cohort is taken in two segments
    (3, 50_000, 300, 1100),    # three segments, lists in a partial second round, odd row count
    (4, 65_535, 64, 9000),     # the largest cohort (four segments), more lists than one workgroup takes
    (2, 40_900, 7, 33),        # lists shorter than a group of eight over three segments
    (1, 45_000, 300, 5),       # a single row
    (3, 20_440, 17, 1025),     # two rows per workgroup, the largest cohort that allows it; odd row count
    (2, 20_441, 8, 5),         # one member more: back to one row per workgroup
    (7, 9_000, 1, 33),         # four rows per workgroup, lists of one member
    (9, 4_000, 9, 2_050),      # a partial last group of one entry, more than two rounds of lists per lane
    (4, 600, 600, 3),          # every member selected
    (1, 64, 64, 1)])
def test_gather_stats_shapes(rows, nc, k, lists):
    """hypmi_gather_stats on every rows-per-workgroup variant of the LDS-resident kernel, ragged list
    lengths and list counts, against float64 numpy on the same lists (the lists come from the top-k
    kernel, so they are in its bank order)."""
    import torch

    from hyperion_amd.score_norm.adapt_s_norm import gather_stats, lists_to_numpy, topk_rows

    rng = np.random.RandomState(rows * 1000 + k)
    X = (rng.randn(rows, nc) * 4 - 9).astype(np.float32)
    X[0, :] += 50.0 * (rng.rand(nc) < 0.01)  # a few far outliers in one row
    if lists * nc <= 50_000_000:
        sel_src = torch.from_numpy((rng.randn(lists, nc) * 3).astype(np.float32)).cuda()
    else:  # what the lists are selected from only has to be random
        sel_src = torch.randn((lists, nc), device="cuda", generator=torch.Generator(device="cuda").manual_seed(k)) * 3
    idx = topk_rows(sel_src, k, 0)
    del sel_src
    L = lists_to_numpy(idx, k)
    assert np.all(np.sort(L, axis=1)[:, 1:] > np.sort(L, axis=1)[:, :-1]) and L.max() < nc  # k different members
    mean, istd, _ = gather_stats(torch.from_numpy(X).cuda(), None, idx, lists, 1e-5, k)
    mean, istd = mean[:, :lists].cpu().numpy().astype(np.float64), istd[:, :lists].cpu().numpy().astype(np.float64)
    sel = X.astype(np.float64)[:, L]  # [rows, lists, k]
    m_ref = sel.mean(axis=2)
    s_ref = np.maximum(sel.std(axis=2), 1e-5)
    assert np.max(np.abs(mean - m_ref) / np.maximum(np.abs(m_ref), 1.0)) <= 2e-6
    assert np.max(np.abs(istd * s_ref - 1.0)) <= (2e-5 if k > 1 else 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("nc", [20_000, 45_000])  # undivided, and in three segments
def test_gather_stats_tail_selection_and_constant_members(nc):
    """Selections far out in the tail of their row (where a one-pass variance about the row's pivot
    would cancel) and selections of equal values: the kernel redoes those pairs around the list's own
    pivot; the count of such pairs is reported."""
    import ctypes

    import torch

    from hyperion_amd import _lib
    from hyperion_amd.score_norm.adapt_s_norm import gather_stats, lists_to_numpy, topk_rows

    rng = np.random.RandomState(4)
    rows, k, lists = 6, 300, 40
    X = (rng.randn(rows, nc) * 6 - 14).astype(np.float32)
    X[2, :400] = 7.25          # a block of equal values
    X[3, :400] = 60.0 + 0.01 * rng.randn(400).astype(np.float32)  # a narrow cluster 12 sigma away
    sel_src = (rng.randn(lists, nc)).astype(np.float32)
    sel_src[:20, :400] += 100.0  # the first 20 lists select inside the first 400 members
    idx = topk_rows(torch.from_numpy(sel_src).cuda(), k, 0)
    L = lists_to_numpy(idx, k)
    lib = _lib.load()
    n0 = ctypes.c_ulonglong(0)
    _lib.check(lib.hypmi_gather_redo_count(ctypes.byref(n0), 1))
    mean, istd, _ = gather_stats(torch.from_numpy(X).cuda(), None, idx, lists, 1e-5, k)
    n1 = ctypes.c_ulonglong(0)
    _lib.check(lib.hypmi_gather_redo_count(ctypes.byref(n1), 0))
    assert n1.value >= 20  # at least the lists inside the blocks of rows 2 and 3 (pairs of rows share a redo)
    mean, istd = mean[:, :lists].cpu().numpy().astype(np.float64), istd[:, :lists].cpu().numpy().astype(np.float64)
    sel = X.astype(np.float64)[:, L]
    m_ref, s_ref = sel.mean(axis=2), np.maximum(sel.std(axis=2), 1e-5)
    assert np.all(istd[2, :20] == np.float32(1e5))  # exactly zero variance -> the floor
    assert np.max(np.abs(mean - m_ref) / np.maximum(np.abs(m_ref), 1.0)) <= 2e-6
    ok = np.ones_like(s_ref, bool)
    ok[2, :20] = False
    assert np.max(np.abs(istd * s_ref - 1.0)[ok]) <= 2e-5


@pytest.mark.gpu
def test_gather_clock_diagnostic_and_rounded_rows():
    """hypmi_gather_clock: zero after a reset, a plausible shader clock after a launch of the LDS-resident
    gather.  The same launches compare lists of 300 members (walked on rows rounded to 16 bits of mantissa in
    LDS) and of 200 (unrounded rows) with float64 statistics: both within 2e-6 of a standard deviation in the
    mean and 2e-5 in the deviation."""
    import ctypes

    import torch

    from hyperion_amd import _lib
    from hyperion_amd.score_norm.adapt_s_norm import gather_stats, lists_to_numpy, topk_rows

    rng = np.random.RandomState(11)
    nc, rows, lists = 20000, 64, 2048
    X = (rng.randn(rows, nc) * 7 - 12).astype(np.float32)
    lib = _lib.load()
    ghz = ctypes.c_double(-1.0)
    _lib.check(lib.hypmi_gather_clock(ctypes.byref(ghz), 1))
    _lib.check(lib.hypmi_gather_clock(ctypes.byref(ghz), 0))
    assert ghz.value == 0.0
    for k in (300, 200):
        idx = topk_rows(torch.from_numpy(rng.randn(lists, nc).astype(np.float32)).cuda(), k, 0)
        L = lists_to_numpy(idx, k)
        mean, istd, _ = gather_stats(torch.from_numpy(X).cuda(), None, idx, lists, 1e-5, k)
        mean, istd = mean[:, :lists].cpu().numpy().astype(np.float64), istd[:, :lists].cpu().numpy().astype(np.float64)
        sel = X.astype(np.float64)[:, L]
        m_ref, s_ref = sel.mean(axis=2), sel.std(axis=2)
        assert np.max(np.abs(mean - m_ref) / s_ref) <= 2e-6, k
        assert np.max(np.abs(istd * s_ref - 1.0)) <= 2e-5, k
    _lib.check(lib.hypmi_gather_clock(ctypes.byref(ghz), 0))
    assert 0.5 < ghz.value < 3.0, ghz.value


@pytest.mark.gpu
@pytest.mark.parametrize("ne,nt,nc,k", [(37, 1500, 600, 50), (130, 70, 20_440, 300), (3, 2100, 30_000, 40),
                                        (260, 1030, 4_000, 9), (6, 20_000, 600, 50),  # lists split over blocks
                                        (131, 70, 45_000, 300), (5, 9_000, 65_535, 24)])  # cohorts taken in segments
def test_gather_apply_equals_gather_then_apply(ne, nt, nc, k):
    """The Z-part statistics kernel that finishes the normalisation itself gives the bits of the two
    separate launches, for every rows-per-workgroup variant and ragged shapes (row groups are
    renumbered in sets of 128, most of which are empty here); cohorts that are taken in segments have
    the paired form only."""
    import torch

    from hyperion_amd import _lib, engine
    from hyperion_amd.score_norm.adapt_s_norm import (gather_apply, gather_apply_supported, gather_pairs_supported,
                                                      gather_stats, seg_plan, topk_rows)

    rng = np.random.RandomState(ne + nt)
    A_ec = torch.from_numpy((rng.randn(ne, nc) * 5 - 8).astype(np.float32)).cuda()
    A_tc = torch.from_numpy((rng.randn(nt, nc) * 5 - 8).astype(np.float32)).cuda()
    S = torch.from_numpy((rng.randn(ne, nt) * 9).astype(np.float32)).cuda()
    t_idx = topk_rows(A_tc, k, 0)
    e_idx = topk_rows(A_ec, k, 0)
    mz, iz, ldz = gather_stats(A_ec, None, t_idx, nt, 1e-5, k)
    mt, it, ldt = gather_stats(A_tc, None, e_idx, ne, 1e-5, k)
    ld = engine.round_up(nt, 32)
    ref = torch.full((ne, ld), 7.0, dtype=torch.float32, device="cuda")
    lib = _lib.load()
    _lib.check(lib.hypmi_asnorm_apply(engine._ptr(S), S.stride(0), ne, nt, engine._ptr(mz), engine._ptr(iz), ldz,
                                      engine._ptr(mt), engine._ptr(it), ldt, engine._ptr(ref), ld,
                                      engine.stream_ptr()))
    assert gather_pairs_supported(nc, ne, k)
    if seg_plan(nc, k) is None:
        assert gather_apply_supported(nc, ne)
        out = torch.full((ne, ld), 7.0, dtype=torch.float32, device="cuda")
        gather_apply(A_ec, t_idx, nt, 1e-5, k, S, mt, it, ldt, out)
        assert torch.equal(out, ref)  # including the untouched padding columns
    assert not gather_apply_supported(45_000, ne) and seg_plan(45_000, k) is not None and seg_plan(20_440, k) is None
    # the same two steps with mean and 1 / std of a (test, model) pair side by side in one array
    from hyperion_amd.score_norm.adapt_s_norm import gather_apply_pairs, gather_stats_pairs

    st, ldp = gather_stats_pairs(A_tc, e_idx, ne, 1e-5, k)
    assert st.shape == (nt, ldp, 2)
    assert torch.equal(st[:, :ne, 0], mt[:, :ne]) and torch.equal(st[:, :ne, 1], it[:, :ne])
    out2 = torch.full((ne, ld), 7.0, dtype=torch.float32, device="cuda")
    gather_apply_pairs(A_ec, t_idx, nt, 1e-5, k, S, st, ldp, out2)
    assert torch.equal(out2, ref)


@pytest.mark.gpu
def test_backend_fused_and_separate_normalisation_agree():
    import synth
    from hyperion_amd.backend import PLDABackend
    from hyperion_amd.pdfs.plda import SPLDA

    m = synth.make_splda(40, 12)
    xe, _ = synth.raw_xvectors(300, d=40, seed=21, num_centers=30)
    xt, _ = synth.raw_xvectors(700, d=40, seed=22, num_centers=30)
    xc, _ = synth.raw_xvectors(400, d=40, seed=23, num_centers=200)
    be = PLDABackend(SPLDA(mu=m["mu"], V=m["V"], W=m["W"]), None, nbest=50, test_chunk=512)
    coh = be.prepare_cohort(xe, xc)
    fused = be.score_snorm(xt, coh).cpu().numpy()
    be.fuse_apply = False
    assert np.array_equal(be.score_snorm(xt, coh).cpu().numpy(), fused)


@pytest.mark.gpu
def test_topk_lazy_exact_rows_equal_full_exact_rows():
    """Top-k with float64 rows made only for the rows whose boundary is in doubt (two calls, a row map
    in the second) selects what top-k with float64 rows for every row selects; near-ties are planted so
    that a good part of the rows needs the second call."""
    import torch

    import synth
    from hyperion_amd import engine
    from hyperion_amd.engine import SIDE_ENROL, SIDE_TEST
    from hyperion_amd.pdfs.plda import SPLDA
    from hyperion_amd.score_norm.adapt_s_norm import lists_to_numpy, topk_rows
    from hyperion_amd.transforms import LDA, CentWhiten, LNorm, TransformList

    chain = synth.chain_cfg2(d=96, lda_dim=40, seed=3)
    m = synth.make_splda(40, 12)
    tl = TransformList([CentWhiten(mu=chain[0][1], T=chain[0][2], name="cw"), LNorm(name="ln"),
                        LDA(mu=chain[2][1], T=chain[2][2], name="lda")])
    plan = SPLDA(mu=m["mu"], V=m["V"], W=m["W"]).score_plan(tl)
    assert plan.pairs_in_prologue()
    xr, _ = synth.raw_xvectors(700, d=96, seed=31, num_centers=40)
    xc, _ = synth.raw_xvectors(3000, d=96, seed=32, num_centers=300)
    xc[2700:] = xc[:300] * (1.0 + 3e-8 * np.random.RandomState(5).randn(300, 1))  # near-duplicates: near-ties
    k = 60
    xr_d, xc_d = engine.upload(xr), engine.upload(xc)
    Gr, bnd32 = plan.prologue_split(xr_d, SIDE_ENROL, bounds=True)
    A = engine.score(Gr, plan.prologue_split(xc_d, SIDE_TEST))
    Gc64, bc = plan.prologue_f64(engine.upload64(xc), SIDE_TEST)
    Gr64, br = plan.prologue_f64(engine.upload64(xr), SIDE_ENROL)
    assert torch.allclose(bnd32, br, rtol=1e-5, atol=1e-6)
    common = {"G_coh": Gc64, "coh_gmax": float(bc[:, 0].max()), "coh_qmax": float(bc[:, 1].max())}
    full = dict(common, G_rows=Gr64, bnd_rows=br, count=True)
    ref = np.sort(lists_to_numpy(topk_rows(A, k, 3, full), k), axis=1)
    asked = []

    def rows64(rows):
        asked.append(rows.cpu().numpy())
        return plan.prologue_f64(engine.upload64(xr[rows.cpu().numpy()]), SIDE_ENROL)

    lazy = dict(common, bnd_rows=bnd32, rows64=rows64, count=True)
    got = np.sort(lists_to_numpy(topk_rows(A, k, 3, lazy), k), axis=1)
    assert np.array_equal(got, ref)
    assert lazy["counters"] == full["counters"] and full["counters"][0] > 20
    assert len(asked) == 1 and lazy["redone"] == len(asked[0]) and 20 < lazy["redone"] < 350
    assert np.all(np.diff(asked[0]) > 0)
    # and the float64 oracle's own selection
    S64 = (Gr64[:, :-2] @ Gc64[:, :-2].T + Gr64[:, -2:-1] + Gc64[:, -2][None, :]).cpu().numpy()
    order = np.argsort(-S64, axis=1, kind="stable")[:, 3:3 + k]
    assert np.array_equal(np.sort(order, axis=1), got)


@pytest.mark.gpu
def test_gather_stats_two_close_members_far_from_the_row_mean():
    """Lists of two members whose values differ by 1e-4 of their distance from the row's mean: the
    pair is recomputed from the rows themselves (the LDS copy, shifted by the row pivot, is already
    rounded to about 1e-5 of that difference's scale), so the standard deviation keeps its digits."""
    import torch

    from hyperion_amd.score_norm.adapt_s_norm import gather_stats

    rng = np.random.RandomState(8)
    rows, nc, lists = 6, 5000, 300
    X = (rng.randn(rows, nc) * 20 + 40).astype(np.float32)
    pairs = rng.choice(nc // 2, size=lists, replace=False)
    a, b = 2 * pairs, 2 * pairs + 1
    X[:, b] = X[:, a] + (0.004 * rng.rand(rows, lists) + 0.001).astype(np.float32)  # close pairs, up to 60 away
    idx = np.zeros((1, lists, 8), dtype=np.uint16)
    idx[0, :, 0], idx[0, :, 1] = a, b
    mean, istd, _ = gather_stats(torch.from_numpy(X).cuda(), None, torch.from_numpy(idx.view(np.int16)).cuda(),
                                 lists, 1e-5, 2)
    x64 = X.astype(np.float64)
    sel = np.stack([x64[:, a], x64[:, b]], axis=2)
    s_ref = np.maximum(sel.std(axis=2), 1e-5)
    assert np.max(np.abs(istd[:, :lists].cpu().numpy() * s_ref - 1.0)) <= 2e-6
    assert np.max(np.abs(mean[:, :lists].cpu().numpy() - sel.mean(axis=2))) <= 4e-6 * 60


def test_two_host_threads_with_their_own_models_and_streams():
    """Two Python threads, each with its own model object (its own plan and workspaces) and its own
    stream, score and normalise at the same time through the host-array API: the same results as one thread
    alone (staging buffers and the error string are per thread, plans are not shared)."""
    import threading

    import synth
    import torch

    from hyperion_amd.score_norm import AdaptSNorm

    m = synth.make_splda(64, 20)
    x = [synth.length_norm(synth.sample_splda(m, n, 1, seed)) for n, seed in ((700, 1), (1600, 2), (900, 3))]
    alone = make("splda", m)
    ref_s = alone.llr_1vs1(x[0], x[1])
    ref_n = AdaptSNorm(nbest=50).predict(ref_s, alone.llr_1vs1(x[2], x[1]), alone.llr_1vs1(x[0], x[2]))
    errors = []

    def work():
        try:
            plda = make("splda", m)
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(6):
                    s = plda.llr_1vs1(x[0], x[1])
                    n = AdaptSNorm(nbest=50).predict(s, plda.llr_1vs1(x[2], x[1]), plda.llr_1vs1(x[0], x[2]))
                    assert np.array_equal(s, ref_s) and np.array_equal(n, ref_n)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work) for _ in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
