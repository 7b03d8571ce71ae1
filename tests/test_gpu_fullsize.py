"""The configurations of the headline benchmark at their FULL sizes, checked through properties that
do not need the reference at that size:

* bilinearity: a score is the inner product of two packed rows, so the sum of a whole score matrix
  (and of any row or column) equals the inner product of the summed packed rows, evaluated here in
  float64 from the same device rows;
* slabs of the device matrix (first / last / interior rows and columns) against the float64 oracle
  recomputed on the host from the same raw vectors;
* for adaptive S-norm, individual entries recomputed by the oracle from the two cohort score vectors
  of that (enrol, test) pair.

Inputs are drawn on the device (seeded torch generator) so that nothing large crosses PCIe.
"""
import numpy as np
import pytest

import synth
from oracle import plda_ref as O

pytestmark = pytest.mark.gpu


def _models():
    from hyperion_amd.pdfs.plda import FRPLDA, PLDA, SPLDA

    def mk(m):
        if m["type"] == "splda":
            return SPLDA(mu=m["mu"], V=m["V"], W=m["W"])
        if m["type"] == "frplda":
            return FRPLDA(mu=m["mu"], B=m["B"], W=m["W"])
        return PLDA(mu=m["mu"], V=m["V"], U=m["U"], D=m["D"])

    return mk


def _chain_objects(chain):
    from hyperion_amd.transforms import LDA, CentWhiten, LNorm, TransformList

    return TransformList([CentWhiten(mu=chain[0][1], T=chain[0][2], name="cw"), LNorm(name="ln"),
                          LDA(mu=chain[2][1], T=chain[2][2], name="lda")])


def _randn(n, d, seed, scale=1.0, shift=0.0):
    import torch

    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn((n, d), device="cuda", generator=g) * scale + shift


def _host64(t):
    return t.cpu().numpy().astype(np.float64)


def _check_bilinear_sums(S, G1, G2, rows, cols):
    """sum(S), sum(S[i, :]) and sum(S[:, j]) against float64 inner products of summed packed rows."""
    import torch

    s1, s2 = G1.double().sum(dim=0), G2.double().sum(dim=0)
    abs_total = float(S.abs().sum(dtype=torch.float64))
    total = float(S.sum(dtype=torch.float64))
    assert abs(total - float(s1 @ s2)) <= 1e-6 * abs_total
    for i in rows:
        ref = float(G1[i].double() @ s2)
        assert abs(float(S[i].sum(dtype=torch.float64)) - ref) <= 1e-6 * float(S[i].abs().sum(dtype=torch.float64))
    for j in cols:
        ref = float(s1 @ G2[j].double())
        col = S[:, j]
        assert abs(float(col.sum(dtype=torch.float64)) - ref) <= 1e-6 * float(col.abs().sum(dtype=torch.float64))


def _assert_close(got, ref, tol=1e-4):
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)
    assert err.max() <= tol, float(err.max())


def test_chain_splda_100k_by_100k():
    """CentWhiten + LNorm + LDA(200) + SPLDA(150), 100 000 x 100 000 trials on one GPU."""
    from hyperion_amd import engine
    from hyperion_amd.engine import SIDE_ENROL, SIDE_TEST

    n1 = n2 = 100_000
    chain = synth.chain_cfg2(d=512, lda_dim=200)
    model = synth.make_splda(200, 150)
    plda = _models()(model)
    plan = plda.score_plan(_chain_objects(chain))
    x1, x2 = _randn(n1, 512, 11), _randn(n2, 512, 12)
    G1, G2 = plan.prologue(x1, SIDE_ENROL), plan.prologue(x2, SIDE_TEST)
    S = engine.score(G1, G2)
    assert tuple(S.shape) == (n1, n2)
    _check_bilinear_sums(S, G1, G2, rows=[0, 31, 54_321, n1 - 1], cols=[0, 77_777, n2 - 1])
    for r0, c0 in [(0, 0), (n1 - 48, n2 - 80), (49_990, 99_900), (99_936, 17)]:
        rs, cs = slice(r0, r0 + 48), slice(c0, c0 + 80)
        y1 = O.chain_predict(_host64(x1[rs]), chain)
        y2 = O.chain_predict(_host64(x2[cs]), chain)
        _assert_close(_host64(S[rs, cs]), O.llr_1vs1(model, y1, y2))


@pytest.mark.parametrize("kind,method", [("frplda", "vavg-lnorm"), ("plda", "savg"), ("frplda", "vavg"),
                                         ("frplda", "book"), ("plda", "book")])
def test_nvsm_50k_enrol_of_3_by_200k(kind, method):
    """FRPLDA / PLDA multi-session enrolment: 50 000 models of 3 segments against 200 000
    single-segment tests (llr_Nvs1 = llr_NvsM with one segment per test)."""
    from hyperion_amd import engine
    from hyperion_amd.engine import SIDE_ENROL, SIDE_TEST

    n_models, segs, n2, d, K = 50_000, 3, 200_000, 200, 150
    model = (synth.make_frplda if kind == "frplda" else synth.make_plda)(d, K)
    plda = _models()(model)
    x1 = _randn(n_models * segs, d, 21, scale=1.0)
    x1 = x1 * (d ** 0.5) / x1.norm(dim=1, keepdim=True)
    x2 = _randn(n2, d, 22)
    x2 = x2 * (d ** 0.5) / x2.norm(dim=1, keepdim=True)
    ids1 = np.repeat(np.arange(n_models), segs)
    S = plda.llr_Nvs1(x1, x2, ids1=ids1, method=method)
    assert tuple(S.shape) == (n_models, n2) and S.is_cuda
    # slabs against the oracle's llr_Nvs1 on the segments of those models
    for m0, c0 in [(0, 0), (n_models - 40, n2 - 64), (25_000, 123_456)]:
        seg = slice(m0 * segs, (m0 + 40) * segs)
        ref = O.llr_Nvs1(model, _host64(x1[seg]), _host64(x2[c0:c0 + 64]), ids1=ids1[seg] - m0, method=method)
        _assert_close(_host64(S[m0:m0 + 40, c0:c0 + 64]), ref)
    if method == "book":
        # By-the-book scoring packs its rows per count group in the model's eigenbasis; the whole-matrix
        # property used here is that the matrix is the sum of an enrol term, a test term and one
        # bilinear form: the second difference over any 2 x 2 minor is the bilinear part alone, which
        # the oracle gives from the same four vectors (segment sets).
        import torch

        rows = torch.tensor([3, 17_777, n_models - 2], device="cuda")
        cols = torch.tensor([9, 150_001, n2 - 3], device="cuda")
        sub = _host64(S[rows][:, cols])
        seg_rows = np.concatenate([np.arange(int(r) * segs, (int(r) + 1) * segs) for r in rows.cpu()])
        ref = O.llr_Nvs1(model, _host64(x1[torch.from_numpy(seg_rows).cuda()]), _host64(x2[cols]),
                         ids1=np.repeat(np.arange(3), segs), method="book")
        _assert_close(sub, ref)
        d2 = sub[0, 0] - sub[0, 1] - sub[1, 0] + sub[1, 1]
        r2 = ref[0, 0] - ref[0, 1] - ref[1, 0] + ref[1, 1]
        assert abs(d2 - r2) <= 1e-3 * max(1.0, abs(r2))
        return
    # whole-matrix property: the score is bilinear in the packed rows that were multiplied
    if method == "savg":
        from hyperion_amd.pdfs.plda.plda_base import pool_rows

        plan = plda.score_plan()
        G1 = pool_rows(plan.prologue(x1, SIDE_ENROL), ids1, mean=True)[0].contiguous()
    else:
        plan = plda.score_plan(lnorm_first=(method == "vavg-lnorm"))
        G1 = plan.prologue(pool_rows_mean(x1, ids1), SIDE_ENROL)
    G2 = plan.prologue(engine.upload(x2), SIDE_TEST)
    _check_bilinear_sums(S, G1, G2, rows=[0, n_models - 1], cols=[5, n2 - 1])


def pool_rows_mean(x, ids):
    from hyperion_amd.pdfs.plda.plda_base import pool_rows

    return pool_rows(x, ids, mean=True)[0]


@pytest.mark.parametrize("with_chain", [False, True])
def test_adapt_s_norm_10k_enrol_by_1m_test(with_chain):
    """Adaptive S-norm with a 20 000 cohort and the top 300, 10 000 enrolment models x 1 000 000 tests
    (chunks of 65 536 tests): 96 entries spread over the matrix, first and last chunk included,
    recomputed by the oracle from that pair's enrol-vs-cohort and cohort-vs-test score vectors.
    with_chain: raw 512-d x-vectors through CentWhiten + LNorm + LDA(200) in the prologue (the
    configuration as the benchmark runs it); otherwise 200-d length-normalised inputs."""
    from hyperion_amd.backend import PLDABackend

    ne, nt, nc, d, K, nbest = 10_000, 1_000_000, 20_000, 200, 150, 300
    model = synth.make_splda(d, K)
    plda = _models()(model)
    chain = synth.chain_cfg2(d=512, lda_dim=d) if with_chain else None

    def unit(n, seed):
        if with_chain:
            return _randn(n, 512, seed)
        x = _randn(n, d, seed)
        return x * (d ** 0.5) / x.norm(dim=1, keepdim=True)

    def _host64(t):  # oracle inputs: after the oracle's own chain when there is one
        a = t.cpu().numpy().astype(np.float64)
        return O.chain_predict(a, chain) if with_chain else a

    xe, xc, xt = unit(ne, 31), unit(nc, 32), unit(nt, 33)
    be = PLDABackend(plda, _chain_objects(chain) if with_chain else None, nbest=nbest, test_chunk=65536)
    coh = be.prepare_cohort(xe, xc)
    out = be.score_snorm(xt, coh)
    assert tuple(out.shape) == (ne, nt)
    import torch

    assert bool(torch.isfinite(out[:, ::997]).all())
    rng = np.random.RandomState(5)
    es = np.concatenate([[0, ne - 1], rng.randint(0, ne, 6)])
    ts = np.concatenate([[0, nt - 1, 65_535, 65_536, 983_040], rng.randint(0, nt, 7)])
    ye, yt, yc = _host64(xe[es]), _host64(xt[ts]), _host64(xc)
    ref = O.adapt_s_norm(O.llr_1vs1(model, ye, yt), O.llr_1vs1(model, yc, yt), O.llr_1vs1(model, ye, yc), nbest, 0)
    got = out[torch.from_numpy(es).cuda()][:, torch.from_numpy(ts).cuda()].cpu().numpy().astype(np.float64)
    _assert_close(got, ref, tol=2e-5)


def test_adapt_s_norm_1m_enrol_shard_of_tests():
    """One GPU's view of the 1M x 1M configuration: all 1 000 000 enrolment models (an 80 GB
    enrol-vs-cohort panel, one million top-300 lists) against a shard of 8192 tests, streamed through
    the reduce callback in chunks; sampled entries against the oracle, and the streamed sums against
    the sums of the same blocks."""
    import torch

    from hyperion_amd.backend import PLDABackend

    ne, nt, nc, d, K, nbest = 1_000_000, 8192, 20_000, 200, 150, 300
    model = synth.make_splda(d, K)
    plda = _models()(model)

    def unit(n, seed):
        x = _randn(n, d, seed)
        return x * (d ** 0.5) / x.norm(dim=1, keepdim=True)

    xe, xc, xt = unit(ne, 41), unit(nc, 42), unit(nt, 43)
    be = PLDABackend(plda, None, nbest=nbest, test_chunk=4096)
    coh = be.prepare_cohort(xe, xc)
    assert tuple(coh["A_ec"].shape) == (ne, nc)
    rng = np.random.RandomState(6)
    es = np.concatenate([[0, ne - 1, 65_535, 65_536], rng.randint(0, ne, 4)])
    ts = np.concatenate([[0, nt - 1, 4095, 4096], rng.randint(0, nt, 4)])
    es_dev = torch.from_numpy(es).cuda()
    picked, sums = {}, []

    def sink(c0, block):  # block: [n_enrol, chunk] normalised scores of tests c0 .. c0 + chunk
        sums.append(float(block.sum(dtype=torch.float64)))
        for j, t in enumerate(ts):
            if c0 <= t < c0 + block.shape[1]:
                picked[j] = block[es_dev, t - c0].double().cpu().numpy()

    assert be.score_snorm(xt, coh, reduce=sink) is None
    assert len(sums) == 2 and all(np.isfinite(sums)) and len(picked) == len(ts)
    got = np.stack([picked[j] for j in range(len(ts))], axis=1)
    ye, yt, yc = _host64(xe[es_dev]), _host64(xt[torch.from_numpy(ts).cuda()]), _host64(xc)
    ref = O.adapt_s_norm(O.llr_1vs1(model, ye, yt), O.llr_1vs1(model, yc, yt), O.llr_1vs1(model, ye, yc), nbest, 0)
    _assert_close(got, ref, tol=2e-5)
