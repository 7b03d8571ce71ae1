"""Generate tests/golden/*.npz from the imported reference (run in the build container only).

    python3 -B oracle/gen_golden.py

TEST INFRASTRUCTURE.  Each fixture holds seeded inputs and the float64 outputs of the reference
classes on them; the script also asserts that oracle/plda_ref.py reproduces every output to
1e-10 (relative to the output scale), which is what pins the restatement.  The fixtures are
data only: no reference source text is stored.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402

import load_ref  # noqa: E402
import plda_ref as O  # noqa: E402
import synth  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
R = load_ref.load()


def close(a, b, tol=1e-10):
    a, b = np.asarray(a, float), np.asarray(b, float)
    scale = max(1.0, float(np.nanmax(np.abs(b)))) if b.size else 1.0
    err = float(np.nanmax(np.abs(a - b))) if b.size else 0.0
    assert np.array_equal(np.isnan(a), np.isnan(b))
    assert err <= tol * scale, (err, scale)
    return err


def ref_model(m):
    if m["type"] == "splda":
        return R.SPLDA(mu=m["mu"].copy(), V=m["V"].copy(), W=m["W"].copy())
    if m["type"] == "frplda":
        return R.FRPLDA(mu=m["mu"].copy(), B=m["B"].copy(), W=m["W"].copy())
    r = R.PLDA(mu=m["mu"].copy(), V=m["V"].copy(), U=m["U"].copy(), D=m["D"].copy())
    assert r.is_init
    r._WV = r._VW  # instance attribute so that the broken llr_NvsM_book (plda.py:481) can run
    return r


def small_models(d, K, seed):
    rng = np.random.RandomState(seed)
    V = rng.randn(K, d)
    A = rng.randn(d, d)
    W = A @ A.T / d + np.eye(d)
    A2 = rng.randn(d, d)
    B = A2 @ A2.T / d + 0.5 * np.eye(d)
    return {
        "splda": {"type": "splda", "mu": rng.randn(d), "V": V, "W": W},
        "frplda": {"type": "frplda", "mu": rng.randn(d), "B": B, "W": 2 * W},
        "plda": {"type": "plda", "mu": rng.randn(d), "V": V, "U": rng.randn(3, d),
                 "D": np.abs(rng.randn(d)) + 1},
    }


def flat(prefix, m):
    return {prefix + k: v for k, v in m.items() if k != "type"}


def gen_selfcheck():
    """Known answers recorded from an earlier run of the reference on the same seeds (loader self-check)."""
    rng = np.random.RandomState(0)
    d, yd = 16, 4
    V = rng.randn(yd, d)
    A = rng.randn(d, d)
    W = A @ A.T / d + np.eye(d)
    m = R.SPLDA(mu=rng.randn(d), V=V, W=W)
    x = m.sample(5, 2, seed=1)
    s = m.llr_1vs1(x[::2], x[1::2])
    assert np.allclose(s[0, :3], [2.97965075, -7.73765166, -77.20524113], atol=1e-7)
    fr = R.FRPLDA(mu=rng.randn(d), B=W, W=2 * W)
    s2 = fr.llr_1vs1(x[::2], x[1::2])
    assert np.allclose(s2[0, :2], [28.60123297, 1.69248982], atol=1e-7)
    pl = R.PLDA(mu=rng.randn(d), V=V, U=rng.randn(3, d), D=np.abs(rng.randn(d)) + 1)
    s3 = pl.llr_1vs1(x[::2], x[1::2])
    assert np.allclose(s3[0, :2], [2.50270301, -8.85197729], atol=1e-7)
    sc, ct, ec = rng.randn(6, 7), rng.randn(30, 7), rng.randn(6, 30)
    s4 = R.AdaptSNorm(nbest=10).predict(sc, ct, ec)
    assert np.allclose(s4[0, :3], [1.04342288, 0.54618728, -1.02196569], atol=1e-7)
    print("selfcheck ok")


def gen_1vs1():
    out = {}
    for d, K, n1, n2, tag in [(24, 6, 9, 11, "a"), (40, 12, 33, 70, "b")]:
        models = small_models(d, K, seed=100 + d)
        rng = np.random.RandomState(5 + d)
        x1 = rng.randn(n1, d)
        x2 = rng.randn(n2, d) * 1.5 + 0.1
        out["x1_" + tag], out["x2_" + tag] = x1, x2
        for name, m in models.items():
            s = ref_model(m).llr_1vs1(x1, x2)
            close(O.llr_1vs1(m, x1, x2), s)
            out.update(flat("%s_%s_" % (name, tag), m))
            out["%s_%s_scores" % (name, tag)] = s
    np.savez_compressed(os.path.join(OUT, "llr_1vs1.npz"), **out)
    print("llr_1vs1 ok")


def gen_transforms():
    rng = np.random.RandomState(11)
    d, d2, d3, n = 20, 20, 7, 15
    x = rng.randn(n, d) * 2 + 0.3
    mu1, T1 = rng.randn(d), rng.randn(d, d2)
    mu2, T2 = rng.randn(d2), rng.randn(d2, d3)
    mu3, T3 = rng.randn(d), rng.randn(d, d3)
    out = dict(x=x, mu1=mu1, T1=T1, mu2=mu2, T2=T2, mu3=mu3, T3=T3)
    out["cw"] = R.CentWhiten(mu=mu1, T=T1).predict(x)
    close(O.cent_whiten_predict(x, mu1, T1), out["cw"])
    out["cw_noT"] = R.CentWhiten(mu=mu1, T=None).predict(x)
    out["cw_nomu"] = R.CentWhiten(mu=None, T=T1).predict(x)
    out["lnorm"] = R.LNorm(mu=mu1, T=T1).predict(x)
    close(O.lnorm_predict(x, mu1, T1), out["lnorm"])
    out["lnorm_plain"] = R.LNorm().predict(x)
    close(O.lnorm_predict(x), out["lnorm_plain"])
    out["lda"] = R.LDA(mu=mu3, T=T3).predict(x)
    close(O.lda_predict(x, mu3, T3), out["lda"])
    out["pca"] = R.PCA(mu=mu3, T=T3).predict(x)
    out["pca_nomu"] = R.PCA(mu=None, T=T3).predict(x)
    # recipe order LDA -> LNorm (egs/voxceleb/v1/steps_be/train-be-v1.py:92)
    tl = R.TransformList([R.LDA(mu=mu3, T=T3, name="lda"), R.LNorm(name="ln")])
    out["chain_lda_lnorm"] = tl.predict(x)
    close(O.chain_predict(x, [("lda", mu3, T3), ("lnorm", None, None)]), out["chain_lda_lnorm"])
    # benchmark order CentWhiten -> LNorm -> LDA
    tl = R.TransformList([R.CentWhiten(mu=mu1, T=T1, name="cw"), R.LNorm(name="ln"),
                          R.LDA(mu=mu2, T=T2, name="lda")])
    out["chain_cw_lnorm_lda"] = tl.predict(x)
    close(O.chain_predict(x, [("cw", mu1, T1), ("lnorm", None, None), ("lda", mu2, T2)]),
          out["chain_cw_lnorm_lda"])
    # two length norms: PCA -> LNorm(mu,T) -> LDA -> LNorm
    tl = R.TransformList([R.PCA(mu=mu1, T=T1, name="pca"), R.LNorm(mu=mu2, T=None, name="ln1"),
                          R.LDA(mu=mu2, T=T2, name="lda"), R.LNorm(name="ln2")])
    out["chain_two_norms"] = tl.predict(x)
    close(O.chain_predict(x, [("pca", mu1, T1), ("lnorm", mu2, None), ("lda", mu2, T2),
                              ("lnorm", None, None)]), out["chain_two_norms"])
    # zero vector through LNorm (the +1e-10 guard)
    xz = x.copy()
    xz[3] = 0
    out["x_zero_row"] = xz
    out["lnorm_zero_row"] = R.LNorm().predict(xz)
    close(O.lnorm_predict(xz), out["lnorm_zero_row"])
    np.savez_compressed(os.path.join(OUT, "transforms.npz"), **out)
    print("transforms ok")


def gen_nvsm():
    d, K = 40, 12
    models = small_models(d, K, seed=77)
    rng = np.random.RandomState(8)
    # enrol: mixed counts {1,2,3}, ids not sorted; test: counts {1,2}
    ids1 = np.array([0, 0, 0, 1, 2, 2, 3, 3, 3, 4, 5, 5, 6, 6, 6, 7])
    perm = rng.permutation(len(ids1))
    ids1 = ids1[perm]
    ids2 = np.array([0, 1, 1, 2, 3, 3, 4, 5, 6, 6, 7, 8, 9, 9])
    x1 = rng.randn(len(ids1), d)
    x2 = rng.randn(len(ids2), d) + 0.2
    out = dict(x1=x1, x2=x2, ids1=ids1, ids2=ids2)
    for name, m in models.items():
        out.update(flat(name + "_", m))
        for method in ["book", "vavg", "vavg-lnorm", "savg"]:
            key = method.replace("-", "_")
            s = ref_model(m).llr_NvsM(x1.copy(), x2.copy(), ids1=ids1, ids2=ids2, method=method)
            close(O.llr_NvsM(m, x1, x2, ids1, ids2, method), s)
            out["%s_NvsM_%s" % (name, key)] = s
            s = ref_model(m).llr_Nvs1(x1.copy(), x2.copy(), ids1=ids1, method=method)
            close(O.llr_Nvs1(m, x1, x2, ids1, method), s)
            out["%s_Nvs1_%s" % (name, key)] = s
        # stats passed directly (ids=None): D = (N, F, S)
        N1, F1, _ = O.compute_stats_hard(x1, ids1)
        N2, F2, _ = O.compute_stats_hard(x2, ids2)
        s = ref_model(m).llr_NvsM((N1, F1.copy(), None), (N2, F2.copy(), None), method="book")
        close(s, out["%s_NvsM_book" % name])
    # compute_stats_hard itself, incl. an id that never occurs (class 2 missing)
    idsg = np.array([0, 1, 1, 3, 3, 3])
    xg = rng.randn(6, 5)
    N, F, S = R.SPLDA.compute_stats_hard(xg, idsg)
    No, Fo, _ = O.compute_stats_hard(xg, idsg)
    close(No, N)
    close(Fo, F)
    out.update(stats_x=xg, stats_ids=idsg, stats_N=N, stats_F=F)
    np.savez_compressed(os.path.join(OUT, "llr_NvsM.npz"), **out)
    print("NvsM ok")


def gen_score_norm():
    rng = np.random.RandomState(21)
    ne, nt, nc = 13, 17, 64
    sc = rng.randn(ne, nt) * 5 - 3
    ct = rng.randn(nc, nt) * 4 - 6
    ec = rng.randn(ne, nc) * 4 - 6
    out = dict(scores=sc, coh_test=ct, enr_coh=ec)
    for nbest, disc, tag in [(10, 0, "n10"), (20, 3, "n20d3"), (100, 0, "clamp"), (100, 60, "clampd")]:
        s = R.AdaptSNorm(nbest=nbest, nbest_discard=disc).predict(sc.copy(), ct.copy(), ec.copy())
        close(O.adapt_s_norm(sc, ct, ec, nbest, disc), s)
        out["asnorm_" + tag] = s
    mct = rng.rand(nc, nt) > 0.2
    mec = rng.rand(ne, nc) > 0.2
    out.update(mask_coh_test=mct, mask_enr_coh=mec)
    s = R.AdaptSNorm(nbest=10).predict(sc.copy(), ct.copy(), ec.copy(), mct, mec)
    close(O.adapt_s_norm(sc, ct, ec, 10, 0, mct, mec), s)
    out["asnorm_masked"] = s
    s = R.SNorm().predict(sc.copy(), ct.copy(), ec.copy())
    close(O.s_norm(sc, ct, ec), s)
    out["snorm"] = s
    s = R.SNorm().predict(sc.copy(), ct.copy(), ec.copy(), mct, mec)
    close(O.s_norm(sc, ct, ec, mct, mec), s)
    out["snorm_masked"] = s
    out["tnorm"] = R.TNorm().predict(sc.copy(), ct.copy())
    close(O.t_norm(sc, ct), out["tnorm"])
    out["znorm"] = R.ZNorm().predict(sc.copy(), ec.copy())
    close(O.z_norm(sc, ec), out["znorm"])
    # constant cohort -> std floor
    ec0 = np.full_like(ec, 1.5)
    out["enr_coh_const"] = ec0
    out["znorm_floor"] = R.ZNorm().predict(sc.copy(), ec0.copy())
    close(O.z_norm(sc, ec0), out["znorm_floor"])
    out["asnorm_floor"] = R.AdaptSNorm(nbest=10).predict(sc.copy(), ct.copy(), ec0.copy())
    close(O.adapt_s_norm(sc, ct, ec0, 10, 0), out["asnorm_floor"])
    # ZT- and TZ-norm: the reference's classes cannot be constructed (zt_norm.py:17, tz_norm.py:17
    # name SNorm without importing it), so their predict bodies are run here on the reference's own
    # ZNorm / TNorm objects
    cc = rng.randn(nc, nc) * 4 - 6
    mcc = rng.rand(nc, nc) > 0.2
    out.update(coh_coh=cc, mask_coh_coh=mcc)
    try:
        from hyperion.score_norm import ZTNorm
        ZTNorm()
        raise SystemExit("ZTNorm can be constructed: use it directly")
    except NameError:
        pass
    zn, tn = R.ZNorm(), R.TNorm()
    for tag, (a, b, c) in (("", (None, None, None)), ("_masked", (mct, mec, mcc))):
        # the Z-norm of the cohort-vs-test scores takes mask_enr_coh as written in zt_norm.py:38-40;
        # with masks that needs ne == nc, so the masked case uses mask_coh_coh there (shapes force it)
        zt = tn.predict(zn.predict(sc.copy(), ec.copy(), b), zn.predict(ct.copy(), cc.copy(), c), a)
        tz = zn.predict(tn.predict(sc.copy(), ct.copy(), a), tn.predict(ec.copy(), cc.copy(), c), b)
        out["ztnorm" + tag], out["tznorm" + tag] = zt, tz
        if tag == "":
            close(O.zt_norm(sc, ct, ec, cc), zt), close(O.tz_norm(sc, ct, ec, cc), tz)
        else:
            close(O.zt_norm(sc, ct, ec, cc, a, b, c), zt), close(O.tz_norm(sc, ct, ec, cc, a, b, c), tz)
    np.savez_compressed(os.path.join(OUT, "score_norm.npz"), **out)
    print("score_norm ok")


def gen_metrics():
    out = {}
    # the reference's own test inputs (tests/hyperion/metrics/test_eer.py:14-29,
    # test_dcf.py:14-44): tar = linspace(-2,10,1000)+3, non = linspace(-10,2,1000)+3
    tar = np.linspace(-2, 10, 1000) + 3
    non = np.linspace(-10, 2, 1000) + 3
    eer = R.compute_eer(tar, non)
    assert 0.166 < eer < 0.167
    close(O.compute_eer(tar, non), eer)
    dcf = R.compute_min_dcf(tar, non, 0.5)[0]
    assert 0.332 < dcf < 0.334
    close(O.compute_min_dcf(tar, non, 0.5)[0], dcf)
    out.update(lin_eer=eer, lin_min_dcf=dcf)
    rng = np.random.RandomState(4)
    tar = rng.randn(400) * 4 + 8
    non = rng.randn(3000) * 8 - 14
    out.update(tar=tar, non=non, eer=R.compute_eer(tar, non),
               min_dcf_005=R.compute_min_dcf(tar, non, 0.05)[0],
               min_dcf_001=R.compute_min_dcf(tar, non, 0.01)[0])
    close(O.compute_eer(tar, non), out["eer"])
    close(O.compute_min_dcf(tar, non, 0.05)[0], out["min_dcf_005"])
    close(O.compute_min_dcf(tar, non, 0.01)[0], out["min_dcf_001"])
    # the rest of the detection metrics (hyperion/metrics/dcf.py, eer.py), with ties and priors that
    # put a threshold exactly on a score
    from hyperion.metrics.dcf import compute_act_dcf, compute_dcf, fast_eval_dcf_eer
    from hyperion.metrics.eer import compute_prbep
    from hyperion.metrics.roc import compute_rocch
    tq, nq = np.round(tar, 1), np.round(non, 1)  # many ties, within and across the classes
    priors = np.array([0.001, 0.01, 0.05, 0.5, 1 / (1 + np.exp(tq[7]))])
    priors.sort()
    for tag, (a, b) in (("", (tar, non)), ("q_", (tq, nq))):
        pm, pf = compute_rocch(a, b)
        close(O.compute_rocch(a, b)[0], pm), close(O.compute_rocch(a, b)[1], pf)
        mind = R.compute_min_dcf(a, b, priors)
        actd = compute_act_dcf(a, b, priors)
        act1 = compute_act_dcf(a, b, 0.05, normalize=False)
        fast = fast_eval_dcf_eer(a, b, priors)
        for x, y in zip(O.compute_min_dcf_multi(a, b, priors), mind):
            close(x, y)
        for x, y in zip(O.compute_act_dcf(a, b, priors), actd):
            close(x, y)
        for x, y in zip(O.compute_act_dcf(a, b, 0.05, normalize=False), act1):
            close(x, y)
        for x, y in zip(O.fast_eval_dcf_eer(a, b, priors), fast):
            close(x, y)
        close(O.compute_prbep(a, b), compute_prbep(a, b))
        close(O.compute_dcf(pm, pf, priors, False), compute_dcf(pm, pf, priors, False))
        out.update({tag + "rocch_pmiss": pm, tag + "rocch_pfa": pf, tag + "prbep": compute_prbep(a, b),
                    tag + "min_dcf": np.stack(mind), tag + "act_dcf": np.stack(actd), tag + "act_dcf_005_raw": np.array([float(np.ravel(v)[0]) for v in act1]),
                    tag + "fast_min": fast[0], tag + "fast_act": fast[1], tag + "fast_eer": fast[2], tag + "fast_prbep": fast[3]})
    out.update(priors=priors, tar_q=tq, non_q=nq)
    # math helpers (tests/hyperion/utils/test_math.py pins them against scipy inv/det)
    A = rng.randn(9, 9)
    A = A @ A.T + np.eye(9)
    X = rng.randn(4, 9)
    fh, Rch, logdet, invA = R.math.invert_pdmat(A, right_inv=True, return_logdet=True, return_inv=True)
    out.update(pd_A=A, pd_X=X, pd_logdet=logdet, pd_inv=invA, pd_XiA=fh(X))
    close(O.logdet_pdmat(A), logdet)
    close(O.inv_pdmat(A), invA)
    fh, logdetR = R.math.invert_trimat(Rch, right_inv=True, return_logdet=True)[:2]
    out.update(tri_R=Rch, tri_XiR=fh(X), tri_logdet=logdetR)
    close(O.right_solve_upper(Rch, X), out["tri_XiR"])
    np.savez_compressed(os.path.join(OUT, "metrics_math.npz"), **out)
    print("metrics ok")


def gen_medium():
    """A mid-size SPLDA case on the calibrated synthetic model (scores cross zero) plus the
    benchmark chain, for the GPU parity tests: reference scores for 96 x 160 trials at d=64."""
    m = synth.make_splda(64, 20)
    x = synth.length_norm(synth.sample_splda(m, 32, 8, seed=7))
    x1, x2 = x[::2][:96], x[1::2][:160]
    s = ref_model(m).llr_1vs1(x1, x2)
    close(O.llr_1vs1(m, x1, x2), s)
    out = dict(x1=x1.astype(np.float32), x2=x2.astype(np.float32), scores=s)
    out.update(flat("m_", m))
    # x stored as float32 to keep the file small; scores are for the float32-rounded inputs
    s = ref_model(m).llr_1vs1(out["x1"].astype(float), out["x2"].astype(float))
    out["scores"] = s
    np.savez_compressed(os.path.join(OUT, "splda_medium.npz"), **out)
    print("medium ok")


def gen_project():
    """SPLDA.project + self-scoring, the diarization flow (hyperion/diarization/diar_ahc_plda.py:
    131-143): PCA basis T and mean shift, projected model, llr_1vs1(x, x)."""
    rng = np.random.RandomState(12)
    m = small_models(12, 4, 5)["splda"]
    T = np.linalg.qr(rng.randn(12, 7))[0] * np.exp(rng.uniform(-0.5, 0.5, 7))
    delta = 0.3 * rng.randn(12)
    r = ref_model(m).project(T, delta.copy())
    mu, V, W = O.splda_project(m["mu"], m["V"], m["W"], T, delta)
    close(mu, r.mu), close(V, r.V), close(W, r.W)
    x = rng.randn(40, 7)
    s = r.llr_1vs1(x, x)
    close(O.llr_1vs1({"type": "splda", "mu": mu, "V": V, "W": W}, x, x), s)
    out = dict(T=T, delta_mu=delta, p_mu=r.mu, p_V=r.V, p_W=r.W, x=x, scores=s)
    out.update(flat("m_", m))
    np.savez_compressed(os.path.join(OUT, "splda_project.npz"), **out)
    print("project ok")


def gen_model_eval():
    """Conditional log-densities, statistics centring, the named pooling variants and the PCA
    dimension chooser of the reference on seeded inputs."""
    rng = np.random.RandomState(31)
    d, K = 12, 4
    ms = small_models(d, K, 9)
    n = 37
    x, y, z = rng.randn(n, d), rng.randn(n, K), rng.randn(n, 3)
    out = dict(x=x, y=y, z=z)
    r = ref_model(ms["splda"])
    out["splda_logp_y"] = r.log_probx_g_y(x, y)
    close(O.splda_log_probx_g_y(ms["splda"], x, y), out["splda_logp_y"])
    r = ref_model(ms["plda"])
    # PLDA.log_probx_g_y itself stops with a ValueError unless n == x_dim (plda.py:414 applies the
    # left inverse to row vectors); its evident intent is SPLDA's density with the within-class
    # precision of compute_aux, which is what the reference is asked for here
    try:
        r.log_probx_g_y(x, y)
        raise AssertionError("the reference's PLDA.log_probx_g_y ran: pin it directly")
    except ValueError:
        pass
    out["plda_logp_y"] = R.SPLDA(mu=ms["plda"]["mu"].copy(), V=ms["plda"]["V"].copy(),
                                 W=np.asarray(r._W).copy()).log_probx_g_y(x, y)
    out["plda_logp_yz"] = r.log_probx_g_yz(x, y, z)
    close(O.plda_log_probx_g_y(ms["plda"], x, y), out["plda_logp_y"])
    close(O.plda_log_probx_g_yz(ms["plda"], x, y, z), out["plda_logp_yz"])
    for k in ("_DU", "_J", "_Lz", "_log_Lz", "_W", "_VW", "_VWV"):
        out["plda_aux" + k] = np.asarray(getattr(r, k))
    # centring of statistics
    ids = rng.randint(0, 9, n)
    ids[:9] = np.arange(9)
    D = R.SPLDA.compute_stats_hard(x, ids)
    mu = rng.randn(d)
    Nc, Fc, Sc = R.SPLDA.center_stats(D, mu)
    for a, b in zip(O.center_stats(O.compute_stats_hard(x, ids), mu), (Nc, Fc, Sc)):
        close(a, b)
    D0 = R.SPLDA.compute_stats_hard_v0(x, ids)
    out.update(ids=ids, cs_mu=mu, cs_N=Nc, cs_F=Fc, cs_S=Sc, v0_N=D0[0], v0_F=D0[1], v0_S=D0[2])
    # the pooling variants under their own names
    x2 = rng.randn(21, d)
    ids2 = rng.randint(0, 5, 21)
    ids2[:5] = np.arange(5)
    for name, m in ms.items():
        r = ref_model(m)
        D1 = r.compute_stats_hard(x, ids)
        D2 = r.compute_stats_hard(x2, ids2)
        for ln in (True, False):
            tag = "%s_%s" % (name, "ln" if ln else "raw")
            out["nvsm_vavg_" + tag] = r.llr_NvsM_vavg(D1, D2, do_lnorm=ln)
            out["nvs1_vavg_" + tag] = r.llr_Nvs1_vavg(D1, x2.copy(), do_lnorm=ln)
            meth = "vavg-lnorm" if ln else "vavg"
            close(O.llr_NvsM(m, x, x2, ids, ids2, method=meth), out["nvsm_vavg_" + tag])
            close(O.llr_Nvs1(m, x, x2, ids, method=meth), out["nvs1_vavg_" + tag])
        out["nvsm_savg_" + name] = r.llr_NvsM_savg(x, ids, x2, ids2)
        out["nvs1_savg_" + name] = r.llr_Nvs1_savg(x, ids, x2)
        close(O.llr_NvsM(m, x, x2, ids, ids2, method="savg"), out["nvsm_savg_" + name])
        close(O.llr_Nvs1(m, x, x2, ids, method="savg"), out["nvs1_savg_" + name])
        out.update(flat(name + "_", m))
    out.update(x2=x2, ids2=ids2)
    # PCA dimension chooser: full rank, rank deficient, energy shares
    xr = rng.randn(50, 5) @ rng.randn(5, d)
    out["pca_x_full"], out["pca_x_def"] = x, xr
    cases = []
    for tag, xx in (("full", x), ("def", xr)):
        for var_r in (1, 0.5, 0.9, 0.99):
            for min_dim in (2, 7):
                got = R.PCA.get_pca_dim_for_var_ratio(xx, var_r, min_dim)
                assert got == O.pca_dim_for_var_ratio(xx, var_r, min_dim)
                cases.append((0 if tag == "full" else 1, var_r, min_dim, got))
    out["pca_dim_cases"] = np.array(cases, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "model_eval.npz"), **out)
    print("model eval ok")


def gen_ahc_levels():
    """Every level of the reference's AHC tree (hyperion/clustering/ahc.py:115-158) on a seeded score
    matrix: the [n, n] table of flat clusters and the homogeneity / completeness curve."""
    from hyperion.clustering.ahc import AHC

    rng = np.random.RandomState(77)
    n = 23
    true = rng.randint(0, 4, n)
    emb = rng.randn(4, 6)[true] * 2 + rng.randn(n, 6)
    scores = emb @ emb.T
    a = AHC(method="average", metric="llr")
    a.fit(scores.copy())
    a.compute_flat_clusters()
    h, c = a.evaluate_homogeneity_completeness_tradeoff(true)
    for i in range(n):  # the table agrees with the single-level method of the same class
        assert np.array_equal(a.flat_clusters[i], a.get_flat_clusters_from_num_clusters(n - i))
    np.savez_compressed(os.path.join(OUT, "ahc_levels.npz"), scores=scores, true=true, Z=a.Z,
                        flat_clusters=a.flat_clusters, homogeneity=h, completeness=c)
    print("ahc levels ok")


def gen_roc_cal_cos():
    """The observed ROC, pool-adjacent-violators on general sequences, the non-parametric
    calibration and cosine scoring of the reference on seeded inputs (ties included)."""
    from hyperion.metrics.roc import compute_roc
    from hyperion.metrics.utils import opt_loglr, pavx

    rng = np.random.RandomState(53)
    tar = np.round(rng.randn(300) + 1.5, 1)  # one decimal: many ties within and across the classes
    non = np.round(rng.randn(2000) - 0.5, 1)
    out = dict(tar=tar, non=non)
    out["roc_pmiss"], out["roc_pfa"] = compute_roc(tar, non)
    for a, b in zip(O.compute_roc(tar, non), (out["roc_pmiss"], out["roc_pfa"])):
        assert np.array_equal(a, b)
    y = np.round(rng.randn(500).cumsum() / 5 + rng.randn(500), 2)
    out["pav_y"] = y
    out["pav_ghat"], out["pav_width"], out["pav_height"] = pavx(y)
    for a, b in zip(O.pavx(y), (out["pav_ghat"], out["pav_width"], out["pav_height"])):
        assert np.array_equal(a, b)
    for method in ("laplace", "raw"):
        with np.errstate(divide="ignore"):
            t, n = opt_loglr(tar, non, method)
        out["cal_tar_" + method], out["cal_non_" + method] = t, n
        a, b = O.opt_loglr(tar, non, method)
        assert np.array_equal(a, t) and np.array_equal(b, n)
    for d in (512, 37):
        x1, x2 = rng.randn(29, d), rng.randn(41, d) * np.exp(rng.randn(41, 1))
        out["cos%d_x1" % d], out["cos%d_x2" % d] = x1, x2
        out["cos%d" % d] = R.math.cosine_scoring(x1, x2)
        close(O.cosine_scoring(x1, x2), out["cos%d" % d], 1e-14)
    np.savez_compressed(os.path.join(OUT, "roc_cal_cos.npz"), **out)
    print("roc / calibration / cosine ok")


def gen_math_helpers():
    """Outputs of the small helpers of hyperion/utils/math.py:102-347 on seeded inputs."""
    M = R.math
    rng = np.random.RandomState(19)
    d = 10
    out = {}
    r = rng.randn(7, d) * 5
    out.update(r=r, softmax=M.softmax(r.copy()), softmax0=M.softmax(r.copy(), axis=0),
               logsumexp=M.logsumexp(r.copy()), logsumexp0=M.logsumexp(r.copy(), axis=0))
    xs = np.concatenate((rng.randn(20) * 30, [-150.0, -100.5, 0.0, 120.0]))
    with np.errstate(over="ignore"):
        out.update(xs=xs, logsigmoid=M.logsigmoid(xs.copy()), neglogsigmoid=M.neglogsigmoid(xs.copy()),
                   sigmoid=M.sigmoid(xs.copy()))
    spd = lambda: (lambda a: a @ a.T / d + np.eye(d))(rng.randn(d, d))  # noqa: E731
    mu1, mu2, S1, S2 = rng.randn(d), rng.randn(d), spd(), spd()
    out.update(mu1=mu1, mu2=mu2, S1=S1, S2=S2, fisher=M.fisher_ratio(mu1, S1, mu2, S2),
               fisher_precs=M.fisher_ratio_with_precs(mu1, S1, mu2, S2))
    A = spd()
    for lower in (False, True):
        t = "_l" if lower else "_u"
        out["sym2vec" + t] = M.symmat2vec(A, lower=lower)
        out["sym2vec_df" + t] = M.symmat2vec(A, lower=lower, diag_factor=0.5)
        out["vec2sym" + t] = M.vec2symmat(out["sym2vec" + t], lower=lower)
        out["vec2sym_df" + t] = M.vec2symmat(out["sym2vec_df" + t], lower=lower, diag_factor=2)
        out["tri2vec" + t] = M.trimat2vec(np.tril(A) if lower else np.triu(A), lower=lower)
        out["vec2tri" + t] = M.vec2trimat(out["tri2vec" + t], lower=lower)
    import scipy.linalg as sla

    S = 0.3 * spd()
    F = spd()
    out.update(A=A, vf_S=S, vf_F=F, vf=M.fullcov_varfloor(S.copy(), F.copy()),
               vf_scalar=M.fullcov_varfloor(S.copy(), 0.7))
    for lower in (False, True):
        t = "_l" if lower else "_u"
        cF, cS = sla.cholesky(F, lower=lower), sla.cholesky(S, lower=lower)
        out["vf_chol" + t] = M.fullcov_varfloor(S.copy(), cF.copy(), F_is_chol=True, lower=lower)
        out["vf_from_chol" + t] = M.fullcov_varfloor_from_cholS(cS.copy(), cF.copy(), lower=lower)
        out["vf_from_chol_scalar" + t] = M.fullcov_varfloor_from_cholS(cS.copy(), np.sqrt(0.7), lower=lower)
    ids = rng.randint(0, 6, 15)
    out.update(ids=ids, onehot=M.int2onehot(ids), onehot9=M.int2onehot(ids, 9))
    np.savez_compressed(os.path.join(OUT, "math_helpers.npz"), **out)
    print("math helpers ok")


def gen_training_stats():
    """Whitening / LDA fits and PLDA sufficient statistics of the reference on labelled vectors
    (hyperion/transforms/cent_whiten.py:35-62, sb_sw.py:25-50, lda.py:36-68,
    hyperion/pdfs/plda/plda_base.py:249-274)."""
    from hyperion.transforms.sb_sw import SbSw

    rng = np.random.RandomState(31)
    n, d, ncls = 400, 10, 23
    y = rng.randint(0, ncls, n)
    y[:ncls] = np.arange(ncls)
    centres = 2.0 * rng.randn(ncls, d)
    x = centres[y] + rng.randn(n, d) * np.exp(rng.uniform(-1, 1, d)) + 3.0
    w = rng.uniform(0.5, 2.0, n)
    out = dict(x=x, y=y, w=w)
    cw = R.CentWhiten()
    cw.fit(x)
    out["cw_mu"], out["cw_T"] = cw.mu, cw.T
    mu, T = O.cent_whiten_fit(x)
    close(mu, cw.mu), close(T, cw.T, 1e-8)
    cww = R.CentWhiten()
    cww.fit(x, sample_weight=w)
    out["cww_mu"], out["cww_T"] = cww.mu, cww.T
    mu, T = O.cent_whiten_fit(x, w)
    close(mu, cww.mu), close(T, cww.T, 1e-8)
    few = R.CentWhiten()
    few.fit(x[:6])
    out["few_mu"], out["few_T"] = few.mu, few.T
    sb = SbSw()
    sb.fit(x, y)
    out["sb_mu"], out["sb_Sb"], out["sb_Sw"] = sb.mu, sb.Sb, sb.Sw
    mu, Sb, Sw = O.sb_sw(x, y)
    close(mu, sb.mu), close(Sb, sb.Sb), close(Sw, sb.Sw)
    lda = R.LDA(lda_dim=6)
    lda.fit(x, y)
    out["lda_mu"], out["lda_T"] = lda.mu, lda.T
    mu, T = O.lda_fit(x, y, 6)
    close(mu, lda.mu), close(T, lda.T, 1e-8)
    for tag, kw in (("pca3_", dict(pca_dim=3)), ("pcar_", dict(pca_var_r=0.8, whiten=True))):
        pca = R.PCA(**kw)
        pca.fit(x)
        out[tag + "mu"], out[tag + "T"] = pca.mu, pca.T
        mu, T = O.pca_fit(x, **kw)
        close(mu, pca.mu), close(T, pca.T, 1e-8)
    for tag, kw in (("", {}), ("w_", dict(sample_weight=w, scale_factor=0.5))):
        N, F, S = R.SPLDA.compute_stats_hard(x, y, **kw)
        out["st_" + tag + "N"], out["st_" + tag + "F"], out["st_" + tag + "S"] = N, F, S
    np.savez_compressed(os.path.join(OUT, "training_stats.npz"), **out)
    print("training stats ok")


def gen_splda_training():
    """SPLDA EM training of the reference (hyperion/pdfs/plda/plda_base.py:34-88,
    hyperion/pdfs/plda/splda.py:50-242) on labelled vectors: posterior of the first model, and the
    ELBO sequence and final parameters of three training set-ups."""
    rng = np.random.RandomState(77)
    d, K, ncls = 8, 3, 40
    counts = rng.randint(1, 7, ncls)
    y = np.repeat(np.arange(ncls), counts)
    n = y.shape[0]
    Vt = rng.randn(K, d)
    x = rng.randn(ncls, K)[y] @ Vt + 0.7 * rng.randn(n, d) * np.exp(rng.uniform(-0.5, 0.5, d)) + 1.5
    perm = rng.permutation(n)
    x, y = x[perm], y[perm]
    w = rng.uniform(0.5, 2.0, n)
    out = dict(x=x, y=y, w=w)

    D = R.SPLDA.compute_stats_hard(x, y)
    Dw = R.SPLDA.compute_stats_hard(x, y, sample_weight=w)
    for tag, DD in (("", D), ("w_", Dw)):
        r = R.SPLDA(y_dim=K)
        r.initialize(DD)
        out["init_" + tag + "mu"], out["init_" + tag + "V"], out["init_" + tag + "W"] = r.mu, r.V, r.W
        mu, V, W = O.splda_initialize(DD, K)
        close(mu, r.mu), close(V, r.V), close(W, r.W)
        m0 = {"type": "splda", "mu": r.mu.copy(), "V": r.V.copy(), "W": r.W.copy()}
        yy, cov, logpy, Ry, Py = r.compute_py_g_x(DD, return_cov=True, return_logpy_0=True, return_acc=True)
        o = O.splda_compute_py_g_x(m0, DD, return_cov=True, return_logpy_0=True, return_acc=True)
        for a, b in zip(o, (yy, cov, logpy, Ry, Py)):
            close(a, b)
        out["post_" + tag + "y"], out["post_" + tag + "cov"], out["post_" + tag + "logpy"] = yy, cov, logpy
        out["post_" + tag + "Ry"], out["post_" + tag + "Py"] = Ry, Py
        st = r.Estep(DD)
        so = O.splda_estep(m0, DD)
        for a, b in zip(so, st):
            close(a, b)
        out["elbo0_" + tag[:1]] = r.elbo(st)
        close(O.splda_elbo(m0, so), r.elbo(st))

    setups = {
        "a": (D, dict(epochs=6, ml_md="ml+md"), {}),
        "b": (Dw, dict(epochs=4, ml_md="ml"), {}),
        "c": (D, dict(epochs=5, ml_md="ml+md", md_epochs=[1, 3]), dict(update_mu=False)),
        "d": (D, dict(epochs=3, ml_md="md"), dict(update_W=False)),
    }
    for tag, (DD, kw, upd) in setups.items():
        r = R.SPLDA(y_dim=K, **upd)
        weights = w if DD is Dw else None
        # fit() recomputes the statistics from x; sample weights are not passed on there
        # (plda_base.py:56), so the weighted set-up drives the same loop on the weighted statistics
        if weights is None:
            elbo, elbo_norm = r.fit(x, class_ids=y, **kw)
        else:
            r.initialize(DD)
            elbo = np.zeros(kw["epochs"])
            for e in range(kw["epochs"]):
                st = r.Estep(DD)
                elbo[e] = r.elbo(st)
                r.MstepML(st)
            elbo_norm = elbo / np.sum(DD[0])
        m, eo = O.splda_fit(DD, K, **kw, **upd)
        close(eo, elbo, 1e-9), close(m["mu"], r.mu, 1e-8), close(m["V"], r.V, 1e-8), close(m["W"], r.W, 1e-8)
        out["fit_%s_elbo" % tag], out["fit_%s_elbo_norm" % tag] = elbo, elbo_norm
        out["fit_%s_mu" % tag], out["fit_%s_V" % tag], out["fit_%s_W" % tag] = r.mu, r.V, r.W
    np.savez_compressed(os.path.join(OUT, "splda_training.npz"), **out)
    print("splda training ok")


def gen_frplda_training():
    """FRPLDA EM training of the reference (hyperion/pdfs/plda/plda_base.py:34-88,
    hyperion/pdfs/plda/frplda.py:49-236), its sampler, and the model interpolation used for domain
    adaptation (splda.py:401-419, frplda.py:396-410, plda.py:564-589): the set-up of the reference's
    own FRPLDA test (tests/hyperion/pdfs/plda/test_frplda.py:22-29) plus an 8-dimensional one with
    ragged and weighted counts."""
    from scipy import linalg as la

    out = {}
    mu = np.array([0.5, 0])
    Sb = np.array([[1, 0.05], [0.05, 0.1]])
    Sw = np.array([[0.1, 0.05], [0.05, 1]])
    truth = {"type": "frplda", "mu": mu, "B": la.inv(Sb), "W": la.inv(Sw)}
    r0 = ref_model(truth)
    x2, y2 = r0.sample(50, 10, seed=1024, return_y=True)
    xo, yo = O.plda_sample(truth, 50, 10, seed=1024, return_y=True)
    close(xo, x2), close(yo, y2)
    out.update(flat("t2_", truth))
    out["t2_x"], out["t2_y"] = x2, y2
    ids2 = np.repeat(np.arange(50), 10)
    r = R.FRPLDA()
    elbo, elbo_norm = r.fit(x2, ids2)
    m, eo = O.frplda_fit(O.compute_stats_hard(x2, ids2))
    close(eo, elbo, 1e-9), close(m["mu"], r.mu, 1e-8), close(m["B"], r.B, 1e-8), close(m["W"], r.W, 1e-8)
    out["t2_elbo"], out["t2_elbo_norm"] = elbo, elbo_norm
    out["t2_fit_mu"], out["t2_fit_B"], out["t2_fit_W"] = r.mu, r.B, r.W

    rng = np.random.RandomState(78)
    d, ncls = 8, 40
    counts = rng.randint(1, 7, ncls)
    y = np.repeat(np.arange(ncls), counts)
    n = y.shape[0]
    x = 1.3 * rng.randn(ncls, d)[y] * np.exp(rng.uniform(-0.5, 0.5, d)) + 0.7 * rng.randn(n, d) @ rng.randn(d, d) / 2 + 1.5
    perm = rng.permutation(n)
    x, y = x[perm], y[perm]
    w = rng.uniform(0.5, 2.0, n)
    out.update(x=x, y=y, w=w)
    D = R.FRPLDA.compute_stats_hard(x, y)
    Dw = R.FRPLDA.compute_stats_hard(x, y, sample_weight=w)
    for tag, DD in (("", D), ("w_", Dw)):
        r = R.FRPLDA()
        r.initialize(DD)
        out["init_" + tag + "mu"], out["init_" + tag + "B"], out["init_" + tag + "W"] = r.mu, r.B, r.W
        for a, b in zip(O.frplda_initialize(DD), (r.mu, r.B, r.W)):
            close(a, b)
        m0 = {"type": "frplda", "mu": r.mu.copy(), "B": r.B.copy(), "W": r.W.copy()}
        got = r.compute_py_g_x(DD, return_cov=True, return_logpy_0=True, return_acc=True)
        for a, b in zip(O.frplda_compute_py_g_x(m0, DD, return_cov=True, return_logpy_0=True, return_acc=True), got):
            close(a, b)
        for name, v in zip(("y", "cov", "logpy", "Ry", "Py"), got):
            out["post_" + tag + name] = v
        st = r.Estep(DD)
        so = O.frplda_estep(m0, DD)
        for a, b in zip(so, st):
            close(a, b)
        out["elbo0_" + tag[:1]] = r.elbo(st)
        close(O.frplda_elbo(m0, so), r.elbo(st))

    setups = {
        "a": (D, dict(epochs=6, ml_md="ml+md"), {}),
        "b": (Dw, dict(epochs=4, ml_md="ml"), {}),
        "c": (D, dict(epochs=5, ml_md="ml"), dict(update_mu=False)),
        "d": (D, dict(epochs=3, ml_md="ml"), dict(update_B=False, fullcov_W=False)),
    }
    for tag, (DD, kw, upd) in setups.items():
        r = R.FRPLDA(**upd)
        if DD is D:
            if not upd.get("update_mu", True):  # a model to keep the mean of
                r.initialize(DD)
                r.mu = r.mu + 0.02
            elbo, elbo_norm = r.fit(x, class_ids=y, **kw)
            m_start = None
            if not upd.get("update_mu", True):
                mu0, B0, W0 = O.frplda_initialize(DD)
                m_start = {"type": "frplda", "mu": mu0 + 0.02, "B": B0, "W": W0}
        else:  # fit() does not pass sample weights on (plda_base.py:56): the same loop on weighted statistics
            r.initialize(DD)
            elbo = np.zeros(kw["epochs"])
            for e in range(kw["epochs"]):
                st = r.Estep(DD)
                elbo[e] = r.elbo(st)
                r.MstepML(st)
            elbo_norm = elbo / np.sum(DD[0])
            m_start = None
        m, eo = O.frplda_fit(DD, m=m_start, **kw, **upd)
        close(eo, elbo, 1e-9), close(m["mu"], r.mu, 1e-8), close(m["B"], r.B, 1e-8), close(m["W"], r.W, 1e-8)
        out["fit_%s_elbo" % tag], out["fit_%s_elbo_norm" % tag] = elbo, elbo_norm
        out["fit_%s_mu" % tag], out["fit_%s_B" % tag], out["fit_%s_W" % tag] = r.mu, r.B, r.W

    # samplers and model interpolation of the three model types
    models = small_models(6, 3, 5)
    others = small_models(6, 3, 6)
    for kind in ("splda", "frplda", "plda"):
        a, b = models[kind], others[kind]
        out.update(flat("adapt_%s_a_" % kind, a))
        out.update(flat("adapt_%s_b_" % kind, b))
        ra, rb = ref_model(a), ref_model(b)
        xs = ra.sample(4, 3, seed=11)
        close(O.plda_sample(a, 4, 3, seed=11), xs)
        out["sample_%s" % kind] = xs
        ra.weighted_avg_model(rb, 0.75, 0.5, 0.25)
        mo = O.plda_weighted_avg_model(a, b, 0.75, 0.5, 0.25)
        for k in mo:
            if k != "type":
                # rows of an interpolated V / U are eigenvectors: equal up to sign
                if k in ("V", "U"):
                    close(np.abs(mo[k] @ mo[k].T), np.abs(getattr(ra, k) @ getattr(ra, k).T), 1e-9)
                    close(mo[k].T @ mo[k], getattr(ra, k).T @ getattr(ra, k), 1e-9)
                else:
                    close(mo[k], getattr(ra, k), 1e-9)
                out["adapt_%s_out_%s" % (kind, k)] = getattr(ra, k)
    np.savez_compressed(os.path.join(OUT, "frplda_training.npz"), **out)
    print("frplda training ok")


def gen_plda_training():
    """EM training of the full PLDA model in the reference (hyperion/pdfs/plda/plda.py:91-394):
    first model, posterior, E-step accumulators, ELBO sequences and trained parameters."""
    rng = np.random.RandomState(79)
    d, K, Z, ncls = 8, 3, 2, 40
    counts = rng.randint(1, 7, ncls)
    y = np.repeat(np.arange(ncls), counts)
    n = y.shape[0]
    x = (rng.randn(ncls, K)[y] @ rng.randn(K, d) + 0.8 * rng.randn(n, Z) @ rng.randn(Z, d)
         + 0.5 * rng.randn(n, d) * np.exp(rng.uniform(-0.5, 0.5, d)) + 1.5)
    perm = rng.permutation(n)
    x, y = x[perm], y[perm]
    w = rng.uniform(0.5, 2.0, n)
    out = dict(x=x, y=y, w=w, y_dim=K, z_dim=Z)
    D = R.PLDA.compute_stats_hard(x, y)
    Dw = R.PLDA.compute_stats_hard(x, y, sample_weight=w)
    names = ("N", "M", "F", "S", "logpy", "y_acc", "Ry1", "Ry", "Cy", "Py", "Rz1", "Rz", "Ryz", "Cz")
    for tag, DD in (("", D), ("w_", Dw)):
        r = R.PLDA(y_dim=K, z_dim=Z)
        r.initialize(DD)
        init = O.plda_initialize(DD, K, Z)
        for a, b, k in zip(init, (r.mu, r.V, r.U, r.D), ("mu", "V", "U", "D")):
            close(a, b, 1e-9)
            out["init_" + tag + k] = b
        m0 = {"type": "plda", "mu": r.mu.copy(), "V": r.V.copy(), "U": r.U.copy(), "D": r.D.copy()}
        got = r.compute_py_g_x(DD, return_cov=True, return_logpy_0=True, return_acc=True)
        for a, b in zip(O.plda_compute_py_g_x(m0, DD, return_cov=True, return_logpy_0=True, return_acc=True), got):
            close(a, b)
        for name, v in zip(("y", "cov", "logpy", "Ry", "Py"), got):
            out["post_" + tag + name] = v
        st = r.Estep(DD)
        so = O.plda_estep(m0, DD)
        for a, b, k in zip(so, st, names):
            close(a, b, 1e-9)
            out["estep_" + tag + k] = np.asarray(b)
        out["elbo0_" + tag[:1]] = r.elbo(st)
        close(O.plda_elbo(m0, so), r.elbo(st))

    setups = {
        "a": (D, dict(epochs=6, ml_md="ml+md"), {}),
        "b": (Dw, dict(epochs=2, ml_md="ml"), {}),
        "c": (D, dict(epochs=5, ml_md="ml+md", md_epochs=[1, 3]), dict(update_D=False)),
        "d": (D, dict(epochs=3, ml_md="md"), {}),
        "e": (D, dict(epochs=3, ml_md="ml"), dict(update_mu=False)),
    }
    for tag, (DD, kw, upd) in setups.items():
        r = R.PLDA(y_dim=K, z_dim=Z, **upd)
        if DD is D:
            elbo, elbo_norm = r.fit(x, class_ids=y, **kw)
        else:  # fit() does not pass sample weights on (plda_base.py:56): the same loop on weighted statistics
            r.initialize(DD)
            elbo = np.zeros(kw["epochs"])
            for e in range(kw["epochs"]):
                st = r.Estep(DD)
                elbo[e] = r.elbo(st)
                r.MstepML(st)
            elbo_norm = elbo / np.sum(DD[0])
        m, eo = O.plda_fit(DD, K, Z, **kw, **upd)
        close(eo, elbo, 1e-9)
        for k in ("mu", "V", "U", "D"):
            close(m[k], getattr(r, k), 1e-7)
            out["fit_%s_%s" % (tag, k)] = getattr(r, k)
        out["fit_%s_elbo" % tag], out["fit_%s_elbo_norm" % tag] = elbo, elbo_norm
    np.savez_compressed(os.path.join(OUT, "plda_training.npz"), **out)
    print("plda training ok")


def gen_diarization():
    """The diarization consumer of the reference (hyperion/diarization/diar_ahc_plda.py:122-168,
    hyperion/clustering/ahc.py:36-110): length normalisation, PCA fitted on the recording, projected
    SPLDA model, self-scoring, average-linkage AHC cut at a threshold."""
    from hyperion.diarization.diar_ahc_plda import DiarAHCPLDA

    rng = np.random.RandomState(21)
    m = small_models(12, 4, 9)["splda"]
    model = ref_model(m)
    spk = rng.randint(0, 3, 90)
    y = 2.0 * rng.randn(3, 4)[spk]
    x = m["mu"] + y @ m["V"] + rng.multivariate_normal(np.zeros(12), np.linalg.inv(m["W"]), 90)
    out = dict(x=x, spk=spk)
    out.update(flat("m_", m))
    for tag, var_r, thr in (("full", 1, 4.0), ("pca", 0.8, 2.0)):
        diar = DiarAHCPLDA(ref_model(m), preproc=R.LNorm(), threshold=thr, pca_var_r=var_r)
        ids = diar.cluster(x.copy())
        out["ids_" + tag], out["Z_" + tag] = ids, diar._ahc.Z
        got = O.diar_ahc_plda_cluster(m, x, thr, var_r)
        assert np.array_equal(got[0], ids)
        close(got[1], diar._ahc.Z, 1e-9)
        print("diarization", tag, "clusters:", len(np.unique(ids)))
    # unsupervised calibration of the self-scores (diar_ahc_plda.py:100-120): a recording of three
    # speakers and one of a single speaker, for which the BIC says "one Gaussian"
    one = m["mu"] + (2.0 * rng.randn(4)) @ m["V"] + rng.multivariate_normal(np.zeros(12), np.linalg.inv(m["W"]), 6)
    out["x_one"] = one
    for tag, xx in (("three", x), ("one", one)):
        xl = O.lnorm_predict(xx)
        scores = model.llr_1vs1(xl, xl)
        scores_cal, bic, gmm = DiarAHCPLDA._unsup_gmm_calibration(scores.copy())
        scale, bias, bic_o, pi, mu, lam = O.unsup_gmm_calibration(scores)
        close(bic_o, bic, 1e-9), close(pi, gmm.pi, 1e-9), close(mu, gmm.mu.ravel(), 1e-9), close(np.atleast_1d(lam), gmm.Lambda.ravel(), 1e-9)
        close(scale * scores + bias, scores_cal, 1e-9)
        out["cal_%s_bic" % tag], out["cal_%s_pi" % tag] = bic, gmm.pi
        out["cal_%s_mu" % tag], out["cal_%s_Lambda" % tag] = gmm.mu.ravel(), gmm.Lambda.ravel()
        diar = DiarAHCPLDA(ref_model(m), preproc=R.LNorm(), threshold=4.0, do_unsup_cal=True, use_bic=True)
        ids = diar.cluster(xx.copy())
        out["ids_bic_" + tag] = ids
        assert np.array_equal(O.diar_ahc_plda_cluster(m, xx, 4.0, 1, use_bic=True)[0], ids)
        print("unsup cal", tag, "bic", bic, "clusters", len(np.unique(ids)))
    np.savez_compressed(os.path.join(OUT, "diarization.npz"), **out)
    print("diarization ok")


if __name__ == "__main__":
    gen_model_eval()
    gen_math_helpers()
    gen_roc_cal_cos()
    gen_ahc_levels()
    os.makedirs(OUT, exist_ok=True)
    gen_frplda_training()
    gen_plda_training()
    gen_diarization()
    gen_splda_training()
    gen_training_stats()
    gen_project()
    gen_selfcheck()
    gen_1vs1()
    gen_transforms()
    gen_nvsm()
    gen_score_norm()
    gen_metrics()
    gen_medium()
