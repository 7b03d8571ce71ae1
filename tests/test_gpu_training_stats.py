"""Training-side statistics on the device (float64 matrix cores) against the reference's results
(tests/golden/training_stats.npz) and against numpy at sizes that exercise every block shape."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d", [(1, 1), (3, 5), (70, 64), (257, 65), (1000, 130), (5000, 200), (20000, 512)])
@pytest.mark.parametrize("weighted", [False, True])
def test_gram_matches_float64(n, d, weighted):
    from hyperion_amd import engine

    rng = np.random.RandomState(n + d)
    x = rng.randn(n, d) * np.exp(rng.uniform(-2, 2, d)) + rng.randn(d)
    w = rng.uniform(0.1, 3.0, n) if weighted else None
    C, F = engine.gram64(x, w)
    wx = x if w is None else x * w[:, None]
    ref_C, ref_F = wx.T @ x, wx.sum(axis=0)
    scale = np.sqrt(np.outer(np.diag(ref_C), np.diag(ref_C)))
    assert np.max(np.abs(C.cpu().numpy() - ref_C) / scale) <= 1e-13
    assert np.allclose(F.cpu().numpy(), ref_F, rtol=1e-12, atol=1e-12 * np.abs(wx).sum(axis=0).max())
    assert np.array_equal(C.cpu().numpy(), C.cpu().numpy().T)  # both halves come from the same sums


def test_statistics_and_training_are_the_same_bits_from_run_to_run():
    """Partial sums are added in a fixed order (like the reference's serial sums): repeated runs of the
    Gram statistics and of an EM fit give identical bits; the atomic variant of the Gram kernel agrees
    with the ordered one to rounding."""
    import ctypes

    import torch

    import synth
    from hyperion_amd import _lib, engine
    from hyperion_amd.pdfs.plda import FRPLDA, PLDA, SPLDA

    rng = np.random.RandomState(3)
    x = rng.randn(30000, 200) * 3 + 1
    w = rng.uniform(0.5, 2.0, 30000)
    runs = [engine.gram64(x, w) for _ in range(3)]
    for C, F in runs[1:]:
        assert torch.equal(C, runs[0][0]) and torch.equal(F, runs[0][1])
    X, W = engine.upload64(x), engine.upload64(w[:, None])[:, 0].contiguous()
    Ca = torch.empty((200, 200), dtype=torch.float64, device="cuda")
    Fa = torch.empty((200,), dtype=torch.float64, device="cuda")
    _lib.check(_lib.load().hypmi_gram_f64(engine._ptr(X), 30000, 200, 200, engine._ptr(W), engine._ptr(Ca), 200,
                                          engine._ptr(Fa), engine.stream_ptr()))
    assert torch.allclose(0.5 * (Ca + Ca.T), runs[0][0], rtol=1e-12, atol=1e-9)
    assert torch.allclose(Fa, runs[0][1], rtol=1e-12, atol=1e-9)

    m = synth.make_splda(40, 12)
    xs = rng.randn(4000, 40)
    ids = np.repeat(np.arange(500), 8)
    for make in (lambda: SPLDA(y_dim=12), lambda: FRPLDA(), lambda: PLDA(y_dim=12, z_dim=6)):
        fits = []
        for _ in range(2):
            model = make()
            elbo = model.fit(xs, class_ids=ids, epochs=4)
            params = [np.array(v) for k, v in sorted(vars(model).items())
                      if isinstance(v, np.ndarray) and not k.startswith("_")]
            fits.append([np.array(elbo[0] if isinstance(elbo, tuple) else elbo)] + params)
        assert len(fits[0]) >= 4
        for a, b in zip(fits[0], fits[1]):
            assert np.array_equal(a, b), type(model).__name__


def test_gram_of_nothing_and_of_pitched_rows():
    import torch

    from hyperion_amd import engine

    C, F = engine.gram64(np.zeros((0, 7)))
    assert not C.any() and not F.any()
    wide = torch.randn((300, 40), dtype=torch.float64, device="cuda")
    C, _ = engine.gram64(wide[:, 3:20].contiguous())
    ref = wide[:, 3:20].T @ wide[:, 3:20]
    assert torch.allclose(C, ref, rtol=1e-12, atol=1e-12)


def test_fits_match_reference(golden):
    from hyperion_amd.pdfs.plda import SPLDA
    from hyperion_amd.transforms import LDA, CentWhiten, SbSw

    g = golden("training_stats")
    x, y, w = g["x"], g["y"], g["w"]
    cw = CentWhiten()
    cw.fit(x)
    assert np.allclose(cw.mu, g["cw_mu"], rtol=1e-12) and np.allclose(cw.T, g["cw_T"], rtol=1e-7, atol=1e-9)
    cww = CentWhiten()
    cww.fit(x, sample_weight=w)
    assert np.allclose(cww.mu, g["cww_mu"], rtol=1e-12) and np.allclose(cww.T, g["cww_T"], rtol=1e-7, atol=1e-9)
    few = CentWhiten()
    few.fit(x[:6])
    assert np.allclose(few.mu, g["few_mu"], rtol=1e-12) and np.allclose(few.T, g["few_T"])
    given = CentWhiten()
    given.fit(mu=g["cw_mu"], S=np.cov(x.T, bias=True))
    assert np.allclose(given.T, g["cw_T"], rtol=1e-7, atol=1e-9)
    sb = SbSw()
    sb.fit(x, y)
    for got, want in ((sb.mu, g["sb_mu"]), (sb.Sb, g["sb_Sb"]), (sb.Sw, g["sb_Sw"])):
        assert np.allclose(got, want, rtol=1e-10, atol=1e-11)
    # two halves of the classes accumulated, then normalised once
    half = np.isin(y, np.unique(y)[:11])
    acc = SbSw()
    acc.fit(x[half], y[half], normalize=False)
    acc.fit(x[~half], y[~half], normalize=False)
    acc.normalize()
    assert np.allclose(acc.Sw, g["sb_Sw"], rtol=1e-10) and np.allclose(acc.Sb, g["sb_Sb"], rtol=1e-10, atol=1e-11)
    lda = LDA(lda_dim=6)
    lda.fit(x, y)
    assert np.allclose(lda.mu, g["lda_mu"], rtol=1e-10) and np.allclose(lda.T, g["lda_T"], rtol=1e-6, atol=1e-8)
    from hyperion_amd.transforms import PCA

    for tag, kw in (("pca3_", dict(pca_dim=3)), ("pcar_", dict(pca_var_r=0.8, whiten=True))):
        pca = PCA(**kw)
        pca.fit(x)
        assert pca.T.shape == g[tag + "T"].shape
        assert np.allclose(pca.mu, g[tag + "mu"], rtol=1e-12) and np.allclose(pca.T, g[tag + "T"], rtol=1e-7, atol=1e-9)
    # fitted transforms run through the device prologue like loaded ones
    z = lda.predict(cw.predict(x))
    assert z.shape == (len(x), 6) and np.all(np.isfinite(z))
    for tag, kw in (("", {}), ("w_", dict(sample_weight=w, scale_factor=0.5))):
        N, F, S = SPLDA.compute_stats_hard(x, y, **kw)
        assert np.allclose(N, g["st_" + tag + "N"], rtol=1e-6)
        assert np.allclose(F, g["st_" + tag + "F"], rtol=1e-5, atol=1e-4)  # per-class sums are fp32
        assert np.allclose(S, g["st_" + tag + "S"], rtol=1e-12)
