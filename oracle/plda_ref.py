"""CPU oracle: float64 numpy restatement of the reference's PLDA back-end scoring path.

TEST INFRASTRUCTURE ONLY.  This module is the checker for the HIP path: it may be imported by
tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg, and by nothing else.  The
product package (hyperion_amd/) never imports it and has no CPU fallback.

Every function follows the reference's operation order (same numpy/scipy calls in the same
sequence) so that it can double as the "port" CPU baseline; the reference location it restates is
cited in each docstring (paths relative to the reference tree).  Parity of this restatement is
pinned by tests/golden/*.npz, produced by oracle/gen_golden.py from the imported reference.

All arithmetic is float64, embeddings are rows.
"""
import numpy as np
import scipy.linalg as sla

F64 = np.float64


# ----------------------------------------------------------------------------- utils/math.py
def right_solve_upper(R, X):
    """X @ inv(R) for upper-triangular R (hyperion/utils/math.py:81-82, right_inv branch)."""
    return sla.solve_triangular(R.T, X.T, lower=True).T


def chol_upper(A):
    """Upper Cholesky factor R with R.T @ R = A (scipy default used at utils/math.py:17,38)."""
    return sla.cholesky(np.array(A, dtype=F64), lower=False)


def logdet_pdmat(A):
    """hyperion/utils/math.py:14-18."""
    return 2.0 * np.sum(np.log(np.diag(chol_upper(A))))


def inv_pdmat(A):
    """A^{-1} through cho_solve (hyperion/utils/math.py:38-57 with return_inv=True)."""
    R = chol_upper(A)
    return sla.cho_solve((R, False), np.eye(A.shape[0]))


# ----------------------------------------------------------------------------- transforms
def cent_whiten_predict(x, mu=None, T=None):
    """hyperion/transforms/cent_whiten.py:25-33.  The 1-D T branch of the reference raises
    NameError (bare `T`); the evident intent, x * T, is what is restated here."""
    if mu is not None:
        x = x - mu
    if T is not None:
        x = x * T if T.ndim == 1 else np.dot(x, T)
    return x


def lnorm_predict(x, mu=None, T=None):
    """hyperion/transforms/lnorm.py:14-17."""
    x = cent_whiten_predict(x, mu, T)
    mx = np.sqrt(np.sum(x ** 2, axis=1, keepdims=True)) + 1e-10
    return np.sqrt(x.shape[1]) * x / mx


def lda_predict(x, mu, T):
    """hyperion/transforms/lda.py:31-34 and pca.py:39-42 (identical arithmetic)."""
    if mu is not None:
        x = x - mu
    return np.dot(x, T)


def chain_predict(x, chain):
    """hyperion/transforms/transform_list.py:41-44.  `chain` is a list of
    (kind, mu, T) with kind in {"cw", "lnorm", "lda", "pca"}."""
    for kind, mu, T in chain:
        if kind == "cw":
            x = cent_whiten_predict(x, mu, T)
        elif kind == "lnorm":
            x = lnorm_predict(x, mu, T)
        elif kind in ("lda", "pca"):
            x = lda_predict(x, mu, T)
        else:
            raise ValueError(kind)
    return x


# ----------------------------------------------------------------------------- PLDA 1-vs-1
def _combine(gt1, gt2, gn1, gn2, const):
    """Common tail of every llr_1vs1 (e.g. hyperion/pdfs/plda/splda.py:303-316)."""
    Qnon_1 = np.sum(gn1 * gn1, axis=1)[:, None]
    Qnon_2 = np.sum(gn2 * gn2, axis=1)
    Qtar_1 = np.sum(gt1 * gt1, axis=1)[:, None]
    Qtar_2 = np.sum(gt2 * gt2, axis=1)
    scores = 2 * np.dot(gt1, gt2.T)
    scores += Qtar_1 - Qnon_1 + Qtar_2 - Qnon_2
    scores += const
    scores *= 0.5
    return scores


def _llr_from_WV(x1, x2, mu, WV, VV):
    """Shared body of SPLDA.llr_1vs1 (splda.py:279-316) and PLDA.llr_1vs1 (plda.py:435-472)."""
    I = np.eye(VV.shape[0], dtype=F64)
    Rnon = chol_upper(I + VV)
    logLnon = 2 * np.sum(np.log(np.diag(Rnon)))
    Rtar = chol_upper(I + 2 * VV)
    logLtar = 2 * np.sum(np.log(np.diag(Rtar)))
    VWF1 = np.dot(x1 - mu, WV)
    VWF2 = np.dot(x2 - mu, WV)
    gn1 = right_solve_upper(Rnon, VWF1)
    gn2 = right_solve_upper(Rnon, VWF2)
    gt1 = right_solve_upper(Rtar, VWF1)
    gt2 = right_solve_upper(Rtar, VWF2)
    return _combine(gt1, gt2, gn1, gn2, 2 * logLnon - logLtar)


def splda_llr_1vs1(x1, x2, mu, V, W):
    """hyperion/pdfs/plda/splda.py:275-316.  V is [y_dim, d]; W is the d x d within-class
    precision."""
    WV = np.dot(W, V.T)
    VV = np.dot(V, WV)
    return _llr_from_WV(x1, x2, mu, WV, VV)


def splda_project(mu, V, W, T, delta_mu=None):
    """SPLDA.project (hyperion/pdfs/plda/splda.py:421-431): the model of y = (x - delta_mu) T.
    Returns (mu, V, W) of the projected model."""
    mu = np.asarray(mu, F64)
    if delta_mu is not None:
        mu = mu - delta_mu
    Sw = T.T @ (inv_pdmat(W) @ T)
    return mu @ T, V @ T, inv_pdmat(Sw)


def splda_torch_layer_llr(x1, x2, mu, V, U, lnorm=False):
    """The scoring formula of the reference's trainable SPLDA layer
    (hyperion/torch/models/plda/splda.py:112-222, in numpy): the within-class precision is
    parametrised as W = U U^T, the optional `lnorm` is sqrt(d) x / |x| (torch/models/plda/
    plda_base.py:85-86), the triangular factors are applied from the right, and the scores are
    0.5 (2 Qtar_12 + (Qtar_1 - Qnon_1) + (Qtar_2 - Qnon_2) + 2 logLnon - logLtar).  x2 = None gives
    `llr_self`.  (The layer itself cannot be built under this image's torch: its constructor calls
    torch.symeig, splda.py:72, which torch 2 removed.)"""
    if x2 is None:
        x2 = x1
    if lnorm:
        x1 = np.sqrt(x1.shape[-1]) * x1 / np.linalg.norm(x1, axis=-1, keepdims=True)
        x2 = np.sqrt(x2.shape[-1]) * x2 / np.linalg.norm(x2, axis=-1, keepdims=True)
    W = np.dot(U, U.T)
    WV = np.dot(W, V.T)
    VV = np.dot(V, WV)
    I = np.eye(V.shape[0])
    Rnon, Rtar = chol_upper(I + VV), chol_upper(I + 2 * VV)
    logLnon = 2 * np.sum(np.log(np.diag(Rnon)))
    logLtar = 2 * np.sum(np.log(np.diag(Rtar)))
    VWF1, VWF2 = np.dot(x1 - mu, WV), np.dot(x2 - mu, WV)
    gn1, gn2 = right_solve_upper(Rnon, VWF1), right_solve_upper(Rnon, VWF2)
    gt1, gt2 = right_solve_upper(Rtar, VWF1), right_solve_upper(Rtar, VWF2)
    Q1 = (np.sum(gt1 * gt1, axis=1) - np.sum(gn1 * gn1, axis=1))[:, None]
    Q2 = np.sum(gt2 * gt2, axis=1) - np.sum(gn2 * gn2, axis=1)
    scores = 2 * np.dot(gt1, gt2.T)
    scores += Q1 + Q2
    scores += logLnon + logLnon - logLtar
    scores *= 0.5
    return scores


def plda_aux(V, U, D):
    """hyperion/pdfs/plda/plda.py:77-89: returns (_VW [d, y_dim], _VWV [y_dim, y_dim])."""
    DV = V * D
    DU = U * D
    J = np.dot(V, DU.T)
    Lz = np.eye(U.shape[0], dtype=F64) + np.dot(DU, U.T)
    Rz = chol_upper(Lz)
    DUiLz = sla.cho_solve((Rz, False), DU).T  # == mult_iLz(DU.T) with right_inv=True
    VW = DV.T - np.dot(DUiLz, J.T)
    VWV = np.dot(V, VW)
    return VW, VWV


def plda_llr_1vs1(x1, x2, mu, V, U, D):
    """hyperion/pdfs/plda/plda.py:430-472."""
    VW, VWV = plda_aux(V, U, D)
    return _llr_from_WV(x1, x2, mu, VW, VWV)


def frplda_llr_1vs1(x1, x2, mu, B, W):
    """hyperion/pdfs/plda/frplda.py:258-303."""
    Rnon = chol_upper(B + W)
    logLnon = 2 * np.sum(np.log(np.diag(Rnon)))
    Rtar = chol_upper(B + 2 * W)
    logLtar = 2 * np.sum(np.log(np.diag(Rtar)))
    WF1 = np.dot(x1, W)
    WF2 = np.dot(x2, W)
    Bmu = np.dot(mu, B)
    gn1 = right_solve_upper(Rnon, WF1 + Bmu)
    gn2 = right_solve_upper(Rnon, WF2 + Bmu)
    gt1 = right_solve_upper(Rtar, WF1 + 0.5 * Bmu)
    gt2 = right_solve_upper(Rtar, WF2 + 0.5 * Bmu)
    const = 2 * logLnon - logLtar - logdet_pdmat(B) + np.inner(np.dot(mu, B), mu)
    return _combine(gt1, gt2, gn1, gn2, const)


def llr_1vs1(model, x1, x2):
    """Dispatch on a model dict {"type": "splda"|"plda"|"frplda", params...}."""
    t = model["type"]
    if t == "splda":
        return splda_llr_1vs1(x1, x2, model["mu"], model["V"], model["W"])
    if t == "plda":
        return plda_llr_1vs1(x1, x2, model["mu"], model["V"], model["U"], model["D"])
    if t == "frplda":
        return frplda_llr_1vs1(x1, x2, model["mu"], model["B"], model["W"])
    raise ValueError(t)


# ----------------------------------------------------------------------------- N-vs-M
def compute_stats_hard(x, class_ids):
    """hyperion/pdfs/plda/plda_base.py:249-274 without sample weights.  Classes are
    0..max(id); an id that never occurs gives N = 0 and F = 0 (later 0/0 = NaN, as in the
    reference).  S = x^T x is the global second-order statistic that training reads."""
    num_classes = int(np.max(class_ids)) + 1
    N = np.zeros((num_classes,), dtype=F64)
    F = np.zeros((num_classes, x.shape[1]), dtype=F64)
    for i in range(num_classes):
        idx = class_ids == i
        N[i] = np.sum(idx)
        F[i] = np.sum(x[idx], axis=0)
    return N, F, np.dot(x.T, x)


def center_stats(D, mu):
    """hyperion/pdfs/plda/plda_base.py:284-290."""
    N, F, S = D
    Fc = F - np.outer(N, mu)
    Fmu = np.outer(np.sum(F, axis=0), mu)
    Sc = S - Fmu - Fmu.T + np.sum(N) * np.outer(mu, mu)
    return N, Fc, Sc


def splda_log_probx_g_y(m, x, y):
    """hyperion/pdfs/plda/splda.py:264-273."""
    logW = logdet_pdmat(m["W"])
    delta = x - m["mu"] - np.dot(y, m["V"])
    logp = -x.shape[-1] * np.log(2 * np.pi) + logW - np.sum(np.dot(delta, m["W"]) * delta, axis=-1)
    return logp / 2


def plda_log_probx_g_y(m, x, y):
    """hyperion/pdfs/plda/plda.py:407-417: precision given through its inverse diag(1/D) + U^T U."""
    iW = np.diag(1 / m["D"]) + np.dot(m["U"].T, m["U"])
    R = chol_upper(iW)
    logiW = 2.0 * np.sum(np.log(np.diag(R)))
    delta = x - m["mu"] - np.dot(y, m["V"])
    Wdelta = sla.cho_solve((R, False), delta.T).T
    logp = -x.shape[-1] * np.log(2 * np.pi) - logiW - np.sum(Wdelta * delta, axis=-1)
    return logp / 2


def plda_log_probx_g_yz(m, x, y, z):
    """hyperion/pdfs/plda/plda.py:419-428."""
    logD = np.sum(np.log(m["D"]))
    delta = x - m["mu"] - np.dot(y, m["V"]) - np.dot(z, m["U"])
    logp = -x.shape[-1] * np.log(2 * np.pi) + logD - np.sum(m["D"] * delta ** 2, axis=-1)
    return logp / 2


def pca_dim_for_var_ratio(x, var_r=1, min_dim=2):
    """hyperion/transforms/pca.py:44-59."""
    if var_r == 1:
        rank = np.linalg.matrix_rank(x)
        if rank <= min_dim:
            rank = np.linalg.matrix_rank(np.dot(x.T, x))
    else:
        sv = sla.svd(x, compute_uv=False)
        Ecc = np.cumsum(sv ** 2)
        Ecc = Ecc / Ecc[-1]
        rank = np.where(Ecc > var_r)[0][0]
    return max(min_dim, rank)


def _book_pair(F1, F2, n1, n2, proj_non, proj_tar, logL):
    """One (N1_i, N2_j) block of llr_NvsM_book (splda.py:332-376).  proj_non(n) / proj_tar(n)
    map first-order stats to gamma_non / gamma_tar for count n."""
    gn1 = proj_non(n1)(F1)
    gn2 = proj_non(n2)(F2)
    gt1 = proj_tar(n1 + n2)(F1)
    gt2 = proj_tar(n1 + n2)(F2)
    Qnon_1 = np.sum(gn1 * gn1, axis=1)[:, None]
    Qnon_2 = np.sum(gn2 * gn2, axis=1)
    Qtar_1 = np.sum(gt1 * gt1, axis=1)[:, None]
    Qtar_2 = np.sum(gt2 * gt2, axis=1)
    s = 2 * np.dot(gt1, gt2.T)
    s += Qtar_1 - Qnon_1 + Qtar_2 - Qnon_2
    s += logL(n1) + logL(n2) - logL(n1 + n2)
    return s


def llr_NvsM_book(model, D1, D2):
    """SPLDA splda.py:318-379, FRPLDA frplda.py:305-366, PLDA plda.py:474-539.  The PLDA
    version of the reference raises AttributeError (`self._WV` is never set); the evident
    intent, `_VW` from compute_aux, is restated.  Unlike the reference this does not subtract
    N*mu from the caller's F in place."""
    N1, F1 = np.asarray(D1[0], F64), np.array(D1[1], dtype=F64)
    N2, F2 = np.asarray(D2[0], F64), np.array(D2[1], dtype=F64)
    t = model["type"]
    mu = model["mu"]
    if t in ("splda", "plda"):
        if t == "splda":
            WV = np.dot(model["W"], model["V"].T)
            VV = np.dot(model["V"], WV)
        else:
            WV, VV = plda_aux(model["V"], model["U"], model["D"])
        I = np.eye(VV.shape[0], dtype=F64)
        F1 = F1 - N1[:, None] * mu
        F2 = F2 - N2[:, None] * mu

        def proj(n):
            R = chol_upper(I + n * VV)
            return lambda F: right_solve_upper(R, np.dot(F, WV))

        def logL(n):
            return 2 * np.sum(np.log(np.diag(chol_upper(I + n * VV))))

        proj_non = proj_tar = proj
        tail = 0.0
    elif t == "frplda":
        B, W = model["B"], model["W"]
        Bmu = np.dot(mu, B)

        def proj_non(n):
            R = chol_upper(B + n * W)
            return lambda F: right_solve_upper(R, np.dot(F, W) + Bmu)

        def proj_tar(n):
            R = chol_upper(B + n * W)
            return lambda F: right_solve_upper(R, np.dot(F, W) + 0.5 * Bmu)

        def logL(n):
            return 2 * np.sum(np.log(np.diag(chol_upper(B + n * W))))

        tail = -logdet_pdmat(B) + np.inner(np.dot(mu, B), mu)
    else:
        raise ValueError(t)

    scores = np.zeros((len(N1), len(N2)), dtype=F64)
    for n1 in np.unique(N1):
        for n2 in np.unique(N2):
            i = np.where(N1 == n1)[0]
            j = np.where(N2 == n2)[0]
            scores[np.ix_(i, j)] = _book_pair(F1[i], F2[j], n1, n2, proj_non, proj_tar, logL)
    scores += tail
    scores *= 0.5
    return scores


def llr_NvsM(model, x1, x2, ids1=None, ids2=None, method="vavg-lnorm"):
    """hyperion/pdfs/plda/plda_base.py:292-322.  ids=None means x already is (N, F, S)."""
    if method == "savg":
        s11 = llr_1vs1(model, x1, x2)
        N, F, _ = compute_stats_hard(s11, ids1)
        sN1 = F / N[:, None]
        N, F, _ = compute_stats_hard(sN1.T, ids2)
        return F.T / N
    D1 = x1 if ids1 is None else compute_stats_hard(x1, ids1)
    D2 = x2 if ids2 is None else compute_stats_hard(x2, ids2)
    if method == "book":
        return llr_NvsM_book(model, D1, D2)
    a1 = D1[1] / np.expand_dims(D1[0], axis=-1)
    a2 = D2[1] / np.expand_dims(D2[0], axis=-1)
    if method == "vavg-lnorm":
        a1 = lnorm_predict(a1)
        a2 = lnorm_predict(a2)
    elif method != "vavg":
        raise ValueError(method)
    return llr_1vs1(model, a1, a2)


def llr_Nvs1(model, x1, x2, ids1=None, method="vavg-lnorm"):
    """hyperion/pdfs/plda/plda_base.py:324-351."""
    if method == "savg":
        s11 = llr_1vs1(model, x1, x2)
        N, F, _ = compute_stats_hard(s11, ids1)
        return F / N[:, None]
    D1 = x1 if ids1 is None else compute_stats_hard(x1, ids1)
    if method == "book":
        D2 = compute_stats_hard(x2, np.arange(x2.shape[0]))
        return llr_NvsM_book(model, D1, D2)
    a1 = D1[1] / np.expand_dims(D1[0], axis=-1)
    if method == "vavg-lnorm":
        a1 = lnorm_predict(a1)
        x2 = lnorm_predict(x2)
    elif method != "vavg":
        raise ValueError(method)
    return llr_1vs1(model, a1, x2)


# ----------------------------------------------------------------------------- score norm
def adapt_s_norm(scores, scores_coh_test, scores_enr_coh, nbest=100, nbest_discard=0,
                 mask_coh_test=None, mask_enr_coh=None):
    """hyperion/score_norm/adapt_s_norm.py:21-88.  The reference zeroes masked cohort scores in
    the caller's arrays; this restatement works on copies.  The std floor is the literal 1e-5 of
    adapt_s_norm.py:62,85."""
    assert scores_enr_coh.shape[1] == scores_coh_test.shape[0]
    assert nbest_discard < scores_enr_coh.shape[1]
    nbest = min(nbest, scores_enr_coh.shape[1] - nbest_discard)
    if mask_coh_test is not None:
        scores_coh_test = np.where(mask_coh_test, scores_coh_test, 0.0)
    if mask_enr_coh is not None:
        scores_enr_coh = np.where(mask_enr_coh, scores_enr_coh, 0.0)

    best_idx = np.flipud(np.argsort(scores_coh_test, axis=0))[nbest_discard:nbest_discard + nbest]
    z = np.zeros_like(scores)
    for i in range(scores.shape[1]):
        sel = best_idx[:, i]
        mu_z = np.mean(scores_enr_coh[:, sel], axis=1, keepdims=True)
        if mask_enr_coh is None:
            s_z = np.std(scores_enr_coh[:, sel], axis=1, keepdims=True)
        else:
            norm = np.mean(mask_enr_coh[:, sel], axis=1, keepdims=True)
            mu_z = mu_z / norm
            s_z = np.sqrt(np.mean(scores_enr_coh[:, sel] ** 2, axis=1, keepdims=True) / norm
                          - mu_z ** 2)
        s_z = np.clip(s_z, a_min=1e-5, a_max=None)
        z[:, i] = ((scores[:, i:i + 1] - mu_z) / s_z)[:, 0]

    best_idx = np.fliplr(np.argsort(scores_enr_coh, axis=1))[:, nbest_discard:nbest_discard + nbest]
    t = np.zeros_like(scores)
    for i in range(scores.shape[0]):
        sel = best_idx[i]
        mu_t = np.mean(scores_coh_test[sel, :], axis=0, keepdims=True)
        if mask_coh_test is None:
            s_t = np.std(scores_coh_test[sel, :], axis=0, keepdims=True)
        else:
            norm = np.mean(mask_coh_test[sel, :], axis=0, keepdims=True)
            mu_t = mu_t / norm
            s_t = np.sqrt(np.mean(scores_coh_test[sel, :] ** 2, axis=0, keepdims=True) / norm
                          - mu_t ** 2)
        s_t = np.clip(s_t, a_min=1e-5, a_max=None)
        t[i, :] = (scores[i:i + 1, :] - mu_t) / s_t
    return (z + t) / np.sqrt(2)


def z_norm(scores, scores_enr_coh, mask=None, std_floor=1e-5):
    """hyperion/score_norm/z_norm.py:16-32 (on a copy of the cohort scores)."""
    if mask is None:
        mu = np.mean(scores_enr_coh, axis=1, keepdims=True)
        s = np.std(scores_enr_coh, axis=1, keepdims=True)
    else:
        c = np.where(mask, scores_enr_coh, 0.0)
        n = np.mean(mask, axis=1, keepdims=True)
        mu = np.mean(c, axis=1, keepdims=True) / n
        s = np.sqrt(np.mean(c ** 2, axis=1, keepdims=True) / n - mu ** 2)
    s = np.where(s < std_floor, std_floor, s)
    return (scores - mu) / s


def t_norm(scores, scores_coh_test, mask=None, std_floor=1e-5):
    """hyperion/score_norm/t_norm.py:15-31 (on a copy of the cohort scores)."""
    if mask is None:
        mu = np.mean(scores_coh_test, axis=0, keepdims=True)
        s = np.std(scores_coh_test, axis=0, keepdims=True)
    else:
        c = np.where(mask, scores_coh_test, 0.0)
        n = np.mean(mask, axis=0, keepdims=True)
        mu = np.mean(c, axis=0, keepdims=True) / n
        s = np.sqrt(np.mean(c ** 2, axis=0, keepdims=True) / n - mu ** 2)
    s = np.where(s < std_floor, std_floor, s)
    return (scores - mu) / s


def s_norm(scores, scores_coh_test, scores_enr_coh, mask_coh_test=None, mask_enr_coh=None,
           std_floor=1e-5):
    """hyperion/score_norm/s_norm.py:22-34."""
    z = z_norm(scores, scores_enr_coh, mask_enr_coh, std_floor)
    t = t_norm(scores, scores_coh_test, mask_coh_test, std_floor)
    return (z + t) / np.sqrt(2)


# ----------------------------------------------------------------------------- training statistics
def zt_norm(scores, scores_coh_test, scores_enr_coh, scores_coh_coh, mask_coh_test=None, mask_enr_coh=None,
            mask_coh_coh=None):
    """The predict body of hyperion/score_norm/zt_norm.py:21-45 (the class itself cannot be
    constructed there: its __init__ names SNorm, which the module does not import): Z-norm of the
    scores and of the cohort-vs-test scores, then T-norm of the first by the second.  As written
    there, the Z-norm of the cohort-vs-test scores is given mask_enr_coh, whose shape fits only with
    as many enrolment models as cohort members; the mask of that stage's cohort scores,
    mask_coh_coh, is what is used here."""
    s_z = z_norm(scores, scores_enr_coh, mask_enr_coh)
    ct_z = z_norm(scores_coh_test, scores_coh_coh, mask_coh_coh)
    return t_norm(s_z, ct_z, mask_coh_test)


def tz_norm(scores, scores_coh_test, scores_enr_coh, scores_coh_coh, mask_coh_test=None, mask_enr_coh=None,
            mask_coh_coh=None):
    """The predict body of hyperion/score_norm/tz_norm.py:21-45 (same unconstructible __init__):
    T-norm of the scores and of the enrol-vs-cohort scores, then Z-norm of the first by the second."""
    s_t = t_norm(scores, scores_coh_test, mask_coh_test)
    ec_t = t_norm(scores_enr_coh, scores_coh_coh, mask_coh_coh)
    return z_norm(s_t, ec_t, mask_enr_coh)


def sb_sw(x, class_ids):
    """SbSw.fit + normalize (hyperion/transforms/sb_sw.py:25-50): (mu, Sb, Sw)."""
    dim = x.shape[1]
    Sb, Sw, mu = np.zeros((dim, dim)), np.zeros((dim, dim)), np.zeros(dim)
    u_ids = np.unique(class_ids)
    for i in u_ids:
        xi = x[class_ids == i]
        mi = np.mean(xi, axis=0)
        mu += mi
        Sb += np.outer(mi, mi)
        Sw += (xi - mi).T @ (xi - mi) / len(xi)
    mu /= len(u_ids)
    return mu, Sb / len(u_ids) - np.outer(mu, mu), Sw / len(u_ids)


# ----------------------------------------------------------------------------- diarization consumer
def ahc_fit(scores, method="average"):
    """hyperion/clustering/ahc.py:36-59, metric "llr": linkage on max - score over the upper
    triangle, heights mapped back to scores."""
    from scipy.cluster.hierarchy import linkage

    idx = np.triu(np.ones_like(scores, dtype=bool), k=1)
    s = scores[idx]
    max_score = np.max(s)
    Z = linkage(-s + max_score, method=method)
    Z[:, 2] = -Z[:, 2] + max_score
    return Z


def ahc_flat_clusters(Z, thr):
    """hyperion/clustering/ahc.py:81-110: merges whose score is >= thr are applied in order; labels
    are the ranks of the surviving node numbers."""
    return ahc_flat_clusters_n(Z, Z.shape[0] + 1 - int(np.sum(Z[:, 2] >= thr)))


def ahc_flat_clusters_n(Z, num_clusters):
    """hyperion/clustering/ahc.py:81-99: the first N - num_clusters merges applied in order."""
    N = Z.shape[0] + 1
    num_clusters = min(N, num_clusters)
    flat = np.arange(N, dtype=int)
    for i in range(N - num_clusters):
        flat[np.logical_or(flat == Z[i, 0], flat == Z[i, 1])] = N + i
    return np.unique(flat, return_inverse=True)[1]


def diar_ahc_plda_cluster(m, x, threshold=0.0, pca_var_r=1, use_bic=False):
    """hyperion/diarization/diar_ahc_plda.py:122-168 with a parameter-free LNorm as the
    pre-processor: (cluster ids, linkage).  use_bic: the unsupervised calibration is run and a
    negative BIC (one Gaussian explains the scores) puts every segment in one cluster; the
    calibrated scores themselves reach the clustering only together with a histogram plot there
    (diar_ahc_plda.py:151-154), which is not restated."""
    x = lnorm_predict(x)
    mu, V, W = m["mu"], m["V"], m["W"]
    if pca_var_r < 1:
        pmu, T = pca_fit(x, pca_var_r=pca_var_r, whiten=True)
        x = lnorm_predict(np.dot(x - pmu, T))
        mu, V, W = splda_project(mu, V, W, T, pmu)
    scores = splda_llr_1vs1(x, x, mu, V, W)
    if use_bic and unsup_gmm_calibration(scores)[2] < 0:
        return np.zeros(len(x), dtype=int), None
    Z = ahc_fit(scores)
    return ahc_flat_clusters(Z, threshold), Z


def unsup_gmm_calibration(scores, var_floor=1e-3):
    """hyperion/diarization/diar_ahc_plda.py:100-120 on the scores above the diagonal, with the
    tied-variance Gaussian mixture of hyperion/pdfs/mixtures/gmm_tied_diag_cov.py restated for one
    dimension: one Gaussian from the moments of the scores (:53-58; one EM epoch changes nothing);
    its split into two at mean +- one deviation with the same precision (split_comp, :105-121); 20
    EM epochs in which the component variances are floored at var_floor times their mean and then
    pooled into the one shared variance (Mstep, :77-103).  The two Gaussians sharing a variance is
    what makes the log-likelihood ratio of "same speaker" against "different speakers" a linear map
    scale * s + bias of the score.  Returns (scale, bias, bic, pi [2], mu [2], Lambda)."""
    n_all = scores.shape[0]
    x = scores[np.triu(np.ones((n_all, n_all), dtype=bool), 1)]
    n = len(x)
    m1, lam1 = np.mean(x), 1.0 / np.std(x) ** 2

    def comp_llk(pi, mu, lam):
        return np.log(pi) + 0.5 * np.log(lam) - 0.5 * np.log(2 * np.pi) - 0.5 * lam * (x[:, None] - mu) ** 2

    def log_prob(pi, mu, lam):
        r = comp_llk(pi, mu, lam)
        mx = np.max(r, axis=-1, keepdims=True)
        return np.log(np.sum(np.exp(r - mx), axis=-1) + 1e-20) + mx[:, 0]

    sd = 1.0 / np.sqrt(lam1)
    pi, mu, lam = np.array([0.5, 0.5]), np.array([m1 + sd, m1 - sd]), lam1
    for _ in range(20):
        r = comp_llk(pi, mu, lam)
        z = np.exp(r - np.max(r, axis=-1, keepdims=True))
        z /= np.sum(z, axis=-1, keepdims=True)
        N = np.sum(z, axis=0)
        mu = np.dot(z.T, x) / N
        S = np.dot(z.T, x * x) / N - mu * mu
        S = np.maximum(S, var_floor * np.mean(S))
        lam = 1.0 / (np.sum(N * S) / np.sum(N))
        pi = N / np.sum(N)
    scale = (mu[0] - mu[1]) * lam
    bias = (mu[1] ** 2 - mu[0] ** 2) * lam / 2 + np.log(pi[0]) - np.log(pi[1])
    bic = np.mean(log_prob(pi, mu, lam) - log_prob(np.array([1.0]), np.array([m1]), lam1)) - 4 * np.log(n) / 2 / n
    return scale, bias, bic, pi, mu, lam


# ----------------------------------------------------------------------------- SPLDA training
def _cho_solve_right(R, X):
    """X @ inv(R.T R) as invert_pdmat(right_inv=True) does it (hyperion/utils/math.py:41)."""
    return sla.cho_solve((R, False), X.T).T


def splda_initialize(D, y_dim, fullcov_W=True):
    """hyperion/pdfs/plda/splda.py:50-71: class means, their SVD for V, the residual scatter for W."""
    N, F, S = D
    Vytilde = F / N[:, None]
    mu = np.mean(Vytilde, axis=0)
    Vy = Vytilde - mu
    _, s, Vt = sla.svd(Vy, full_matrices=False)
    V = s[:y_dim, None] * Vt[:y_dim, :]
    C = (S - np.dot((N[:, None] * Vytilde).T, Vytilde)) / np.sum(N)
    W = inv_pdmat(C) if fullcov_W else 1 / np.diag(C)
    return mu, V, W


def splda_compute_py_g_x(m, D, return_cov=False, return_logpy_0=False, return_acc=False):
    """hyperion/pdfs/plda/splda.py:73-158: posterior of the speaker factor of every class; the
    classes are visited by distinct count when all counts are whole numbers, one by one otherwise."""
    N, F, _ = D
    mu, V, W = m["mu"], m["V"], m["W"]
    y_dim = V.shape[0]
    M = F.shape[0]
    WV = np.dot(W, V.T)
    VV = np.dot(V, WV)
    gamma = np.dot(F - mu, WV)
    whole = bool(np.all(np.ceil(N) == N))
    compute_inv = return_cov or return_acc
    y = np.zeros((M, y_dim))
    Sigma_y = np.zeros((M, y_dim, y_dim)) if return_cov else None
    logpy = -0.5 * y_dim * np.log(2 * np.pi) * np.ones((M,))
    Py, Ry = np.zeros((y_dim, y_dim)), np.zeros((y_dim, y_dim))
    for k in (np.unique(N) if whole else range(M)):
        if whole:
            i = (N == k).nonzero()[0]
            N_i, M_i = k, len(i)
        else:
            i, N_i, M_i = k, N[k], 1
        R = chol_upper(np.eye(y_dim) + N_i * VV)
        y[i, :] = _cho_solve_right(R, gamma[i, :])
        if compute_inv:
            iL = _cho_solve_right(R, np.eye(y_dim))
        if return_cov:
            Sigma_y[i, :, :] = iL
        if return_logpy_0:
            logL = 2 * np.sum(np.log(np.diag(R)))
            logpy[i] += 0.5 * (logL - np.sum(y[i, :] * gamma[i, :], axis=-1))
        if return_acc:
            Py += M_i * iL
            Ry += N_i * M_i * iL
    out = [y]
    if return_cov:
        out.append(Sigma_y)
    if return_logpy_0:
        out.append(logpy)
    if return_acc:
        out += [Ry, Py]
    return out[0] if len(out) == 1 else tuple(out)


def splda_estep(m, D):
    """hyperion/pdfs/plda/splda.py:160-179."""
    N, F, S = D
    y, logpy, Ry, Py = splda_compute_py_g_x(m, D, return_logpy_0=True, return_acc=True)
    Niy = y * N[:, None]
    Ry = Ry + np.dot(Niy.T, y)
    Py = Py + np.dot(y.T, y)
    return (np.sum(N), F.shape[0], np.sum(F, axis=0), S, np.sum(logpy), np.sum(y, axis=0),
            np.sum(Niy, axis=0), Ry, np.dot(F.T, y), Py)


def splda_elbo(m, stats):
    """hyperion/pdfs/plda/splda.py:181-196."""
    N, M, F, S, logpy_x = stats[:5]
    mu, V, W = m["mu"], m["V"], m["W"]
    x_dim, y_dim = mu.shape[0], V.shape[0]
    Fmu = np.outer(F, mu)
    Shat = S - Fmu - Fmu.T + N * np.outer(mu, mu)
    logpx_y = 0.5 * (-N * x_dim * np.log(2 * np.pi) + N * logdet_pdmat(W)
                     - np.inner(W.ravel(), Shat.ravel()))
    return logpx_y - 0.5 * M * y_dim * np.log(2 * np.pi) - logpy_x


def splda_mstep_ml(m, stats, update_mu=True, update_V=True, update_W=True, fullcov_W=True):
    """hyperion/pdfs/plda/splda.py:198-230; returns the updated model."""
    N, M, F, S, _, y_acc, Ry1, Ry, Cy, Py = stats
    mu, V, W = m["mu"], m["V"], m["W"]
    Rytilde = np.vstack((np.hstack((Ry, Ry1[:, None])), np.hstack((Ry1, N))))
    Cytilde = np.hstack((Cy, F[:, None]))
    if update_mu and not update_V:
        mu = (F - np.dot(Ry1, V)) / N
    if not update_mu and update_V:
        V = sla.cho_solve((chol_upper(Ry), False), Cy.T - np.outer(Ry1, mu))
    if update_mu and update_V:
        Vtilde = sla.cho_solve((chol_upper(Rytilde), False), Cytilde.T)
        V, mu = Vtilde[:-1, :], Vtilde[-1, :]
    if update_W:
        if update_mu and update_V:
            iW = (S - np.dot(Cy, V) - np.outer(F, mu)) / N
        else:
            Vtilde = np.vstack((V, mu))
            CVt = np.dot(Cytilde, Vtilde)
            iW = (S - CVt - CVt.T + np.dot(np.dot(Vtilde.T, Rytilde), Vtilde)) / N
        W = inv_pdmat(iW) if fullcov_W else np.diag(1 / np.diag(iW))
    return {"type": "splda", "mu": mu, "V": V, "W": W}


def splda_mstep_md(m, stats, update_mu=True, update_V=True):
    """hyperion/pdfs/plda/splda.py:232-242; returns the updated model."""
    N, M, F, S, _, y_acc, Ry1, Ry, Cy, Py = stats
    mu, V = m["mu"], m["V"]
    mu_y = y_acc / M
    if update_mu:
        mu = mu + np.dot(mu_y, V)
    if update_V:
        V = np.dot(chol_upper(Py / M - np.outer(mu_y, mu_y)), V)
    return {"type": "splda", "mu": mu, "V": V, "W": m["W"]}


def splda_fit(D, y_dim, m=None, epochs=20, ml_md="ml+md", md_epochs=None, **updates):
    """The training loop of hyperion/pdfs/plda/plda_base.py:34-88 on statistics D = (N, F, S):
    returns (model, elbo per epoch)."""
    if m is None:
        mu, V, W = splda_initialize(D, y_dim, updates.get("fullcov_W", True))
        m = {"type": "splda", "mu": mu, "V": V, "W": W}
    elbo = np.zeros((epochs,))
    md_kw = {k: v for k, v in updates.items() if k in ("update_mu", "update_V")}
    for epoch in range(epochs):
        stats = splda_estep(m, D)
        elbo[epoch] = splda_elbo(m, stats)
        if ml_md != "md":
            m = splda_mstep_ml(m, stats, **updates)
        if ml_md != "ml" and (md_epochs is None or epoch in md_epochs):
            m = splda_mstep_md(m, stats, **md_kw)
    return m, elbo


# ----------------------------------------------------------------------------- FRPLDA training
def frplda_initialize(D):
    """hyperion/pdfs/plda/frplda.py:49-75: class means for mu and the between-class covariance, the
    scatter around them for the within-class covariance; both inverted."""
    N, F, S = D
    M = F.shape[0]
    y = F / N[:, None]
    Fy = np.dot(F.T, y)
    C = S - Fy - Fy.T
    for i in range(M):
        C += N[i] * np.outer(y[i, :], y[i, :])
    C = (C + C.T) / 2
    mu = np.mean(y, axis=0)
    iB = np.dot(y.T, y) / M - np.outer(mu, mu)
    iW = C / np.sum(N)
    return mu, inv_pdmat(iB), inv_pdmat(iW)


def frplda_compute_py_g_x(m, D, return_cov=False, return_logpy_0=False, return_acc=False):
    """hyperion/pdfs/plda/frplda.py:77-158: posterior N(y_i, L_i^-1), L_i = B + N_i W, of every class
    mean; classes are visited by distinct count when all counts are whole numbers.  As there, the
    accumulator Ry comes back as zeros (only Py gathers the posterior covariances, frplda.py:146)."""
    N, F, _ = D
    mu, B, W = m["mu"], m["B"], m["W"]
    M, y_dim = F.shape
    gamma = np.dot(F, W) + np.dot(mu, B)
    whole = bool(np.all(np.ceil(N) == N))
    compute_inv = return_cov or return_acc
    y = np.zeros_like(F)
    Sigma_y = np.zeros((M, y_dim, y_dim)) if return_cov else None
    logpy = -0.5 * y_dim * np.log(2 * np.pi) * np.ones((M,))
    Py, Ry = np.zeros((y_dim, y_dim)), np.zeros((y_dim, y_dim))
    for k in (np.unique(N) if whole else range(M)):
        if whole:
            i = (N == k).nonzero()[0]
            N_i, M_i = k, len(i)
        else:
            i, N_i, M_i = k, N[k], 1
        R = chol_upper(B + N_i * W)
        y[i, :] = _cho_solve_right(R, gamma[i, :])
        if compute_inv:
            iL = _cho_solve_right(R, np.eye(y_dim))
        if return_cov:
            Sigma_y[i, :, :] = iL
        if return_logpy_0:
            logL = 2 * np.sum(np.log(np.diag(R)))
            logpy[i] += 0.5 * (logL - np.sum(y[i, :] * gamma[i, :], axis=-1))
        if return_acc:
            Py += M_i * iL
    out = [y]
    if return_cov:
        out.append(Sigma_y)
    if return_logpy_0:
        out.append(logpy)
    if return_acc:
        out += [Ry, Py]
    return out[0] if len(out) == 1 else tuple(out)


def frplda_estep(m, D):
    """hyperion/pdfs/plda/frplda.py:160-177: (N_tot, M, S, sum logpy, sum y, Ry, Cy = F^T y, Py)."""
    N, F, S = D
    y, logpy, Ry, Py = frplda_compute_py_g_x(m, D, return_logpy_0=True, return_acc=True)
    Niy = y * N[:, None]
    Ry = Ry + np.dot(Niy.T, y)
    Py = Py + np.dot(y.T, y)
    return (np.sum(N), F.shape[0], S, np.sum(logpy), np.sum(y, axis=0), Ry, np.dot(F.T, y), Py)


def frplda_elbo(m, stats):
    """hyperion/pdfs/plda/frplda.py:179-202."""
    N, M, S, logpy_x = stats[:4]
    mu, B, W = m["mu"], m["B"], m["W"]
    d = mu.shape[0]
    logpx_y = 0.5 * (-N * d * np.log(2 * np.pi) + N * logdet_pdmat(W) - np.inner(W.ravel(), S.ravel()))
    logpy = 0.5 * M * (-d * np.log(2 * np.pi) + logdet_pdmat(B) - np.inner(np.dot(mu, B), mu))
    return logpx_y + logpy - logpy_x


def frplda_mstep_ml(m, stats, update_mu=True, update_B=True, update_W=True, fullcov_W=True):
    """hyperion/pdfs/plda/frplda.py:216-233; returns the updated model.  With update_mu off the
    cross term is subtracted twice un-transposed, as there (frplda.py:225-226)."""
    N, M, S, _, y_acc, Ry, Cy, Py = stats
    mu, B, W = m["mu"], m["B"], m["W"]
    ybar = y_acc / M
    if update_mu:
        mu = ybar
    if update_B:
        if update_mu:
            iB = Py / M - np.outer(mu, mu)
        else:
            muybar = np.outer(mu, ybar)
            iB = Py / M - muybar - muybar + np.outer(mu, mu)
        B = inv_pdmat(iB)
    if update_W:
        iW = (S - Cy - Cy.T + Ry) / N
        W = inv_pdmat(iW) if fullcov_W else np.diag(1 / np.diag(iW))
    return {"type": "frplda", "mu": mu, "B": B, "W": W}


def frplda_fit(D, m=None, epochs=20, ml_md="ml+md", md_epochs=None, **updates):
    """The training loop of hyperion/pdfs/plda/plda_base.py:34-88 for the two-covariance model (its
    minimum-divergence step does nothing, frplda.py:235-236): returns (model, elbo per epoch)."""
    if m is None:
        mu, B, W = frplda_initialize(D)
        m = {"type": "frplda", "mu": mu, "B": B, "W": W}
    elbo = np.zeros((epochs,))
    for epoch in range(epochs):
        stats = frplda_estep(m, D)
        elbo[epoch] = frplda_elbo(m, stats)
        if ml_md != "md":
            m = frplda_mstep_ml(m, stats, **updates)
    return m, elbo


# ----------------------------------------------------------------------------- full PLDA training
def plda_aux_all(V, U, D):
    """Every auxiliary value of hyperion/pdfs/plda/plda.py:77-89: DU, J, the upper Cholesky factor of
    Lz = I + (U D) U^T, log|Lz|, the within-class precision W, and the products VW, VWV."""
    DV = V * D
    DU = U * D
    J = np.dot(V, DU.T)
    Rz = chol_upper(np.eye(U.shape[0]) + np.dot(DU, U.T))
    DUiLz = _cho_solve_right(Rz, DU.T)
    W = np.diag(D) - np.dot(DUiLz, DU)
    VW = DV.T - np.dot(DUiLz, J.T)
    return {"DU": DU, "J": J, "Rz": Rz, "log_Lz": 2 * np.sum(np.log(np.diag(Rz))), "W": W, "VW": VW,
            "VWV": np.dot(V, VW)}


def plda_initialize(D, y_dim, z_dim, floor_iD=1e-5):
    """hyperion/pdfs/plda/plda.py:91-117: V from the SVD of the centred class means, U from the
    leading eigenvectors of the residual scatter, D from what those leave on the diagonal."""
    N, F, S = D
    Vytilde = F / N[:, None]
    mu = np.mean(Vytilde, axis=0)
    _, s, Vt = sla.svd(Vytilde - mu, full_matrices=False)
    V = s[:y_dim, None] * Vt[:y_dim, :]
    C = (S - np.dot((N[:, None] * Vytilde).T, Vytilde)) / np.sum(N)
    w, E = sla.eigh(C)
    U = np.fliplr(E * np.sqrt(w))[:, :z_dim].T
    iD = np.diag(C - np.dot(U.T, U)).copy()
    iD[iD < floor_iD] = floor_iD
    return mu, V, U, 1 / iD


def plda_compute_py_g_x(m, D, return_cov=False, return_logpy_0=False, return_acc=False):
    """hyperion/pdfs/plda/plda.py:119-198: the SPLDA posterior with W V^T := _VW, V W V^T := _VWV."""
    N, F, _ = D
    aux = plda_aux_all(m["V"], m["U"], m["D"])
    y_dim = m["V"].shape[0]
    M = F.shape[0]
    gamma = np.dot(F - m["mu"], aux["VW"])
    whole = bool(np.all(np.ceil(N) == N))
    compute_inv = return_cov or return_acc
    y = np.zeros((M, y_dim))
    Sigma_y = np.zeros((M, y_dim, y_dim)) if return_cov else None
    logpy = -0.5 * y_dim * np.log(2 * np.pi) * np.ones((M,))
    Py, Ry = np.zeros((y_dim, y_dim)), np.zeros((y_dim, y_dim))
    for k in (np.unique(N) if whole else range(M)):
        if whole:
            i = (N == k).nonzero()[0]
            N_i, M_i = k, len(i)
        else:
            i, N_i, M_i = k, N[k], 1
        R = chol_upper(np.eye(y_dim) + N_i * aux["VWV"])
        y[i, :] = _cho_solve_right(R, gamma[i, :])
        if compute_inv:
            iL = _cho_solve_right(R, np.eye(y_dim))
        if return_cov:
            Sigma_y[i, :, :] = iL
        if return_logpy_0:
            logpy[i] += 0.5 * (2 * np.sum(np.log(np.diag(R))) - np.sum(y[i, :] * gamma[i, :], axis=-1))
        if return_acc:
            Py += M_i * iL
            Ry += N_i * M_i * iL
    out = [y]
    if return_cov:
        out.append(Sigma_y)
    if return_logpy_0:
        out.append(logpy)
    if return_acc:
        out += [Ry, Py]
    return out[0] if len(out) == 1 else tuple(out)


def plda_estep(m, D):
    """hyperion/pdfs/plda/plda.py:200-264.  The channel-factor accumulators (Rz1, Rz, Ryz, Cz) are
    algebra on the speaker-factor ones.  In Cz the reference subtracts the INNER product of the
    total first-order statistic and mu from every entry of S (np.dot of two vectors, plda.py:214);
    the restatement does the same."""
    N, F, S = D
    mu = m["mu"]
    aux = plda_aux_all(m["V"], m["U"], m["D"])
    DU, J, Rz_chol = aux["DU"], aux["J"], aux["Rz"]
    mult_iLz = lambda X: _cho_solve_right(Rz_chol, X)  # noqa: E731
    y, logpy, Ry, Py = plda_compute_py_g_x(m, D, return_logpy_0=True, return_acc=True)
    N_tot = np.sum(N)
    F_tot = np.sum(F, axis=0)
    y_acc = np.sum(y, axis=0)
    Cy = np.dot(F.T, y)
    A = np.dot(S - np.dot(F_tot.T, mu), DU.T) - np.dot(Cy, J)
    Cz = mult_iLz(A)
    Niy = y * N[:, None]
    Ry1 = np.sum(Niy, axis=0)
    Ry = Ry + np.dot(Niy.T, y)
    Py = Py + np.dot(y.T, y)
    Fc_acc = np.sum(F - np.outer(N, mu), axis=0)
    Fmu = np.outer(F_tot, mu)
    Sc = S - Fmu - Fmu.T + N_tot * np.outer(mu, mu)
    Rz1 = mult_iLz(np.dot(Fc_acc, DU.T) - np.dot(Ry1, J))
    Cbary = Cy - np.outer(mu, Ry1)
    Ryz = mult_iLz(np.dot(Cbary.T, DU.T) - np.dot(Ry, J))
    A = np.dot(np.dot(DU, Cbary), J)
    B = np.dot(np.dot(DU, Sc), DU.T) - A - A.T + np.dot(np.dot(J.T, Ry), J)
    B = mult_iLz(B)
    Rz = mult_iLz(B.T).T + N_tot * mult_iLz(np.eye(DU.shape[0]))
    return (N_tot, F.shape[0], F_tot, S, np.sum(logpy), y_acc, Ry1, Ry, Cy, Py, Rz1, Rz, Ryz, Cz)


def plda_elbo(m, stats):
    """hyperion/pdfs/plda/plda.py:266-282."""
    N, M, F, S, logpy_x = stats[:5]
    mu = m["mu"]
    aux = plda_aux_all(m["V"], m["U"], m["D"])
    Fmu = np.outer(F, mu)
    Shat = S - Fmu - Fmu.T + N * np.outer(mu, mu)
    logpx_y = 0.5 * (-N * mu.shape[0] * np.log(2 * np.pi) + N * (np.sum(np.log(m["D"])) - aux["log_Lz"])
                     - np.inner(aux["W"].ravel(), Shat.ravel()))
    return logpx_y - 0.5 * M * m["V"].shape[0] * np.log(2 * np.pi) - logpy_x


def plda_mstep_ml(m, stats, update_mu=True, update_V=True, update_U=True, update_D=True, floor_iD=1e-5):
    """hyperion/pdfs/plda/plda.py:284-362, branch by branch as written there (the partial-update
    branches included, whatever they do); returns the updated model."""
    N, M, F, S, _, y_acc, Ry1, Ry, Cy, Py, Rz1, Rz, Ryz, Cz = stats
    mu, V, U, Dg = m["mu"], m["V"], m["U"], m["D"]
    y_dim = V.shape[0]
    solve = lambda A, C: sla.cho_solve((chol_upper(A), False), C)  # noqa: E731
    if update_mu and not update_V and not update_U:
        mu = (F - np.dot(Ry1, V) - np.dot(Rz1, U)) / N
    if not update_mu and update_V and not update_U:
        V = solve(Ry, Cy.T - np.outer(Ry1, mu) - np.dot(Ryz, U))
    if not update_mu and not update_V and update_U:
        U = solve(Rz, Cz.T - np.dot(Ryz.T, V) - np.outer(Rz1, mu))
    if not update_mu and update_V and update_U:
        Rytilde = np.vstack((np.hstack((Ry, Ryz)), np.hstack((Ryz.T, Rz))))
        C = np.vstack((Cy.T - np.outer(Ry1, mu), Cz.T - np.outer(Rz1, mu)))
        Vtilde = solve(Rytilde, C)
        V, U = Vtilde[:y_dim], Vtilde[y_dim:]
    if update_mu and not update_V and update_U:
        Rytilde = np.vstack((np.hstack((Rz, Rz1[:, None])), np.hstack((Rz1, N))))
        C = np.vstack((Cz.T - np.outer(Ryz, V), F[:, None] - np.outer(Ry1, V)))
        Vtilde = solve(Rytilde, C)
        U, mu = Vtilde[:-1], Vtilde[-1]
    if not update_mu and update_V and not update_U:
        Rytilde = np.vstack((np.hstack((Ry, Ry1[:, None])), np.hstack((Ry1, N))))
        C = np.vstack((Cy.T - np.dot(Ryz, U), F[:, None] - np.dot(Rz1, U)))
        Vtilde = solve(Rytilde, C)
        V, U = Vtilde[:-1], Vtilde[-1]
    Rytilde = np.vstack((np.hstack((Ry, Ryz, Ry1[:, None])), np.hstack((Ryz.T, Rz, Rz1[:, None])),
                         np.hstack((Ry1, Rz1, N))))
    Cytilde = np.hstack((Cy, Cz, F[:, None]))
    if update_mu and update_V and update_U:
        Vtilde = solve(Rytilde, Cytilde.T)
        V, U, mu = Vtilde[:y_dim, :], Vtilde[y_dim:-1], Vtilde[-1]
    if update_D:
        Vtilde = np.vstack((V, U, mu))
        CVt = np.dot(Cytilde, Vtilde)
        iD = np.diag((S - CVt - CVt.T + np.dot(np.dot(Vtilde.T, Rytilde), Vtilde)) / N).copy()
        iD[iD < floor_iD] = floor_iD
        Dg = 1 / iD
    return {"type": "plda", "mu": mu, "V": V, "U": U, "D": Dg}


def plda_mstep_md(m, stats, update_mu=True, update_V=True, update_U=True):
    """hyperion/pdfs/plda/plda.py:364-394; returns the updated model."""
    N, M, F, S, _, y_acc, Ry1, Ry, Cy, Py, Rz1, Rz, Ryz, Cz = stats
    mu, V, U = m["mu"], m["V"], m["U"]
    mu_y = y_acc / M
    chol_Cov_y = chol_upper(Py / M - np.outer(mu_y, mu_y))
    R = Ry - np.outer(Ry1, Ry1) / N
    H = _cho_solve_right(chol_upper(R), Ryz.T - np.outer(Rz1, Ry1) / N)
    mu_z = (Rz1 - np.dot(Ry1, H.T)) / N
    RzyH = np.dot(Ryz.T, H.T)
    Cov_z = (Rz - RzyH - RzyH.T + np.dot(np.dot(H, Ry), H.T)) / N - np.outer(mu_z, mu_z)
    chol_Cov_z = chol_upper(Cov_z)
    if update_mu:
        mu = mu + np.dot(mu_y, V + np.dot(H.T, U)) + np.dot(mu_z, U)
    if update_V:
        V = np.dot(chol_Cov_y, V + np.dot(H.T, U))
    if update_U:
        U = np.dot(chol_Cov_z, U)
    return {"type": "plda", "mu": mu, "V": V, "U": U, "D": m["D"]}


def plda_fit(D, y_dim, z_dim, m=None, epochs=20, ml_md="ml+md", md_epochs=None, floor_iD=1e-5, **updates):
    """The training loop of hyperion/pdfs/plda/plda_base.py:34-88 for the full model: returns
    (model, elbo per epoch)."""
    if m is None:
        mu, V, U, Dg = plda_initialize(D, y_dim, z_dim, floor_iD)
        m = {"type": "plda", "mu": mu, "V": V, "U": U, "D": Dg}
    elbo = np.zeros((epochs,))
    md_kw = {k: v for k, v in updates.items() if k in ("update_mu", "update_V", "update_U")}
    for epoch in range(epochs):
        stats = plda_estep(m, D)
        elbo[epoch] = plda_elbo(m, stats)
        if ml_md != "md":
            m = plda_mstep_ml(m, stats, floor_iD=floor_iD, **updates)
        if ml_md != "ml" and (md_epochs is None or epoch in md_epochs):
            m = plda_mstep_md(m, stats, **md_kw)
    return m, elbo


# ----------------------------------------------------------------------------- sampling, adaptation
def plda_sample(m, num_classes, num_samples_per_class, seed=1024, return_y=False):
    """Vectors drawn from a model with numpy's RandomState, in the reference's order of draws
    (hyperion/pdfs/plda/splda.py:381-399, frplda.py:368-394, plda.py:541-562)."""
    rng = np.random.RandomState(seed=seed)
    mu = m["mu"]
    d, n = mu.shape[0], num_classes * num_samples_per_class
    if m["type"] == "frplda":
        chol_Sb = chol_upper(inv_pdmat(m["B"]))
        chol_Sw = chol_upper(inv_pdmat(m["W"]))
        z = np.dot(rng.normal(size=(n, d)), chol_Sw)
        y = np.dot(rng.normal(size=(num_classes, d)), chol_Sb) + mu
        y = np.repeat(y, num_samples_per_class, axis=0)
        return (y + z, y) if return_y else y + z
    if m["type"] == "splda":
        z = np.dot(rng.normal(size=(n, d)), chol_upper(inv_pdmat(m["W"])))
        y = np.dot(rng.normal(size=(num_classes, m["V"].shape[0])), m["V"]) + mu
        return np.repeat(y, num_samples_per_class, axis=0) + z
    z1 = rng.normal(size=(n, d)) / m["D"]
    z2 = np.dot(rng.normal(size=(n, m["U"].shape[0])), m["U"])
    y = np.dot(rng.normal(size=(num_classes, m["V"].shape[0])), m["V"]) + mu
    return np.repeat(y, num_samples_per_class, axis=0) + z1 + z2


def plda_weighted_avg_model(m, other, w_mu, w_B, w_W, floor_iD=1e-5):
    """Interpolation of a model towards another one, on covariances (hyperion/pdfs/plda/
    plda_base.py:362-363, splda.py:401-419, frplda.py:396-410, plda.py:564-589): returns the new model."""
    out = dict(m)
    out["mu"] = w_mu * other["mu"] + (1 - w_mu) * m["mu"]
    kind = m["type"]
    if kind == "frplda":
        if w_B > 0:
            out["B"] = inv_pdmat(w_B * inv_pdmat(other["B"]) + (1 - w_B) * inv_pdmat(m["B"]))
        if w_W > 0:
            out["W"] = inv_pdmat(w_W * inv_pdmat(other["W"]) + (1 - w_W) * inv_pdmat(m["W"]))
        return out
    y_dim = m["V"].shape[0]
    if w_B > 0:
        Sb = w_B * np.dot(other["V"].T, other["V"]) + (1 - w_B) * np.dot(m["V"].T, m["V"])
        w, V = sla.eigh(Sb)
        out["V"] = (np.sqrt(w[-y_dim:]) * V[:, -y_dim:]).T
    if w_W > 0 and kind == "splda":
        out["W"] = inv_pdmat(w_W * inv_pdmat(other["W"]) + (1 - w_W) * inv_pdmat(m["W"]))
    if w_W > 0 and kind == "plda":
        z_dim = m["U"].shape[0]
        Sw0 = np.dot(m["U"].T, m["U"]) + np.diag(1 / m["D"])
        Sw = w_W * (np.dot(other["U"].T, other["U"]) + np.diag(1 / other["D"])) + (1 - w_W) * Sw0
        w, U = sla.eigh(Sw)
        U = (np.sqrt(w) * U)[:, -z_dim:].T
        iD = np.diag(Sw - np.dot(U.T, U)).copy()
        iD[iD < floor_iD] = floor_iD
        out["U"], out["D"] = U, 1 / iD
    return out


def _largest_first_positive_top(V):
    V = np.fliplr(V)
    V[:, V[0, :] < 0] *= -1
    return V


def cent_whiten_fit(x, sample_weight=None):
    """CentWhiten.fit from data with more vectors than dimensions (cent_whiten.py:35-62, moments
    as hyperion/pdfs/core/normal.py:102-143): (mu, T)."""
    if sample_weight is None:
        N, F, S = x.shape[0], x.sum(axis=0), x.T @ x
    else:
        wx = sample_weight[:, None] * x
        N, F, S = sample_weight.sum(), wx.sum(axis=0), wx.T @ x
    mu = F / N
    d, V = sla.eigh(S / N - np.outer(mu, mu))
    return mu, _largest_first_positive_top(V * np.sqrt(1 / d))


def pca_fit(x, pca_dim=None, pca_var_r=None, pca_min_dim=2, whiten=False):
    """PCA.fit (hyperion/transforms/pca.py:61-106): (mu, T)."""
    mu = np.mean(x, axis=0)
    d, V = sla.eigh((x - mu).T @ (x - mu) / x.shape[0])
    d, V = np.flip(d), _largest_first_positive_top(V)
    if pca_var_r is not None:
        share = np.cumsum(d) / np.sum(d)
        pca_dim = max(np.where(share > pca_var_r)[0][0], pca_min_dim)
    if whiten:
        V = V / np.sqrt(d)
    return mu, V if pca_dim is None else V[:, :pca_dim]


def lda_fit(x, y, lda_dim=None):
    """LDA.fit (hyperion/transforms/lda.py:36-68): (mu, T)."""
    mu, Sb, Sw = sb_sw(x, y)
    _, V = sla.eigh(Sb, Sw)
    V = _largest_first_positive_top(V)
    return mu, V if lda_dim is None else V[:, :lda_dim]


# ----------------------------------------------------------------------------- metrics (gate)
def pavx(y):
    """Pool-adjacent-violators (hyperion/metrics/utils.py:30-94): returns (ghat, width, height)."""
    n = len(y)
    assert n > 0
    start = np.zeros(n, dtype=int)
    width = np.zeros(n, dtype=int)
    level = np.zeros(n, dtype=F64)
    ci = 0
    start[0], width[0], level[0] = 0, 1, y[0]
    for j in range(1, n):
        ci += 1
        start[ci], width[ci], level[ci] = j, 1, y[j]
        while ci >= 1 and level[ci - 1] >= level[ci]:
            nw = width[ci - 1] + width[ci]
            level[ci - 1] = level[ci - 1] + (width[ci] / nw) * (level[ci] - level[ci - 1])
            width[ci - 1] = nw
            ci -= 1
    ghat = np.repeat(level[:ci + 1], width[:ci + 1])
    return ghat, width[:ci + 1].copy(), level[:ci + 1].copy()


def compute_rocch(tar, non):
    """ROC convex hull vertices (hyperion/metrics/roc.py:54-102)."""
    Nt, Nn = len(tar), len(non)
    N = Nt + Nn
    scores = np.hstack((tar.ravel(), non.ravel()))
    pideal = np.hstack((np.ones(Nt), np.zeros(Nn)))
    order = np.argsort(scores, kind="mergesort")
    pideal = pideal[order]
    _, width, _ = pavx(pideal)
    nbins = len(width)
    p_miss = np.zeros(nbins + 1)
    p_fa = np.zeros(nbins + 1)
    csum = np.concatenate(([0.0], np.cumsum(pideal)))
    left, fa, miss = 0, Nn, 0
    for i in range(nbins):
        p_miss[i] = miss / Nt
        p_fa[i] = fa / Nn
        left += width[i]
        miss = csum[left]
        fa = N - left - (csum[N] - csum[left])
    p_miss[nbins] = miss / Nt
    p_fa[nbins] = fa / Nn
    return p_miss, p_fa


def rocch2eer(p_miss, p_fa):
    """hyperion/metrics/roc.py:105-140."""
    eer = 0.0
    for i in range(len(p_fa) - 1):
        xx = p_fa[i:i + 2]
        yy = p_miss[i:i + 2]
        XY = np.vstack((xx, yy)).T
        dd = np.dot(np.array([1, -1]), XY)
        if np.min(np.abs(dd)) == 0:
            eerseg = 0.0
        else:
            seg = sla.solve(XY, np.array([[1.0], [1.0]]))
            eerseg = 1.0 / np.sum(seg)
        eer = max(eer, eerseg)
    return eer


def compute_eer(tar, non):
    """hyperion/metrics/eer.py:11-22."""
    return rocch2eer(*compute_rocch(tar, non))


def compute_min_dcf(tar, non, prior, normalize=True):
    """hyperion/metrics/dcf.py:36-65 for a scalar prior: returns (min_dcf, p_miss, p_fa)."""
    p_miss, p_fa = compute_rocch(tar, non)
    dcf = prior * p_miss + (1 - prior) * p_fa
    if normalize:
        dcf = dcf / min(prior, 1 - prior)
    i = int(np.argmin(dcf))
    return dcf[i], p_miss[i], p_fa[i]


def compute_dcf(p_miss, p_fa, prior, normalize=True):
    """hyperion/metrics/dcf.py:11-33: [len(prior), len(p_miss)] (or [len(p_miss)] for one prior)."""
    prior = np.asarray(prior)
    if prior.ndim == 1:
        prior = prior[:, None]
    dcf = prior * p_miss + (1 - prior) * p_fa
    if normalize:
        dcf = dcf / np.minimum(prior, 1 - prior)
    return dcf


def compute_min_dcf_multi(tar, non, prior, normalize=True):
    """hyperion/metrics/dcf.py:36-65 for a prior or an ascending vector of priors:
    (min DCF, P_miss there, P_fa there)."""
    p_miss, p_fa = compute_rocch(tar, non)
    dcf = compute_dcf(p_miss, p_fa, prior, normalize)
    idx = np.argmin(dcf, axis=-1)
    if dcf.ndim == 1:
        return dcf[idx], p_miss[idx], p_fa[idx]
    return dcf[np.arange(dcf.shape[0]), idx], p_miss[idx], p_fa[idx]


def compute_act_dcf(tar, non, prior, normalize=True):
    """hyperion/metrics/dcf.py:68-136: detection cost at the Bayes thresholds log((1 - p) / p) of
    scores taken as log-likelihood ratios; ranks from stable sorts of [thresholds, scores]."""
    prior = np.asarray(prior)
    if prior.ndim == 1:
        assert np.all(prior == np.sort(prior, kind="mergesort")), "priors must be in ascending order"
    else:
        prior = prior[None]
    num_priors = len(prior)
    ntar, nnon = len(tar), len(non)
    t = -np.log(prior) + np.log(1 - prior)
    ii = np.argsort(np.concatenate((t, tar)), kind="mergesort")
    r = np.zeros((num_priors + ntar), dtype="int32")
    r[ii] = np.arange(1, num_priors + ntar + 1)
    n_miss = r[:num_priors] - np.arange(num_priors, 0, -1)
    ii = np.argsort(np.concatenate((t, non)), kind="mergesort")
    r = np.zeros((num_priors + nnon), dtype="int32")
    r[ii] = np.arange(1, num_priors + nnon + 1)
    n_fa = nnon - r[:num_priors] + np.arange(num_priors, 0, -1)
    p_miss = n_miss / ntar
    p_fa = n_fa / nnon
    act_dcf = prior * p_miss + (1 - prior) * p_fa
    if normalize:
        act_dcf = act_dcf / np.minimum(prior, 1 - prior)
    if len(act_dcf) == 1:
        act_dcf = act_dcf[0]
    return act_dcf, p_miss, p_fa


def compute_prbep(tar, non):
    """hyperion/metrics/eer.py:25-39: the point of the hull where #misses = #false alarms."""
    p_miss, p_fa = compute_rocch(tar, non)
    return rocch2eer(p_miss * len(tar), p_fa * len(non))


def fast_eval_dcf_eer(tar, non, prior, normalize_dcf=True):
    """hyperion/metrics/dcf.py:139-175: (min DCF, actual DCF, EER, PRBEP) from one hull."""
    p_miss, p_fa = compute_rocch(tar, non)
    dcf = compute_dcf(p_miss, p_fa, prior, normalize_dcf)
    return (np.min(dcf, axis=-1), compute_act_dcf(tar, non, prior, normalize_dcf)[0], rocch2eer(p_miss, p_fa),
            rocch2eer(p_miss * len(tar), p_fa * len(non)))


def compute_roc(true_scores, false_scores):
    """hyperion/metrics/roc.py:13-51."""
    num_true, num_false = len(true_scores), len(false_scores)
    total = num_true + num_false
    scores = np.hstack((true_scores, false_scores))
    labels = np.zeros_like(scores)
    labels[:num_true] = 1
    labels = labels[np.argsort(scores, kind="mergesort")]
    sumtrue = np.cumsum(labels)
    sumfalse = num_false - (np.arange(total) + 1 - sumtrue)
    p_miss, p_fa = np.zeros(total + 1), np.zeros(total + 1)
    p_fa[0] = 1.0
    p_miss[1:] = sumtrue / num_true
    p_fa[1:] = sumfalse / num_false
    return p_miss, p_fa


def opt_loglr(tar, non, method="laplace"):
    """hyperion/metrics/utils.py:97-150."""
    ntar, nnon = len(tar), len(non)
    n = ntar + nnon
    scores = np.concatenate((tar, non))
    p_ideal = np.zeros(n, dtype=F64)
    p_ideal[:ntar] = 1
    sort_idx = np.argsort(scores, kind="mergesort")
    p_ideal = p_ideal[sort_idx]
    if method == "laplace":
        p_ideal = np.concatenate(([1, 0], p_ideal, [1, 0]))
    p_opt, _, _ = pavx(p_ideal)
    if method == "laplace":
        p_opt = p_opt[2:-2]
    with np.errstate(divide="ignore"):
        llr = np.log(p_opt) - np.log(1 - p_opt) - np.log(ntar / nnon)
    llr += 1e-6 * np.arange(n) / n
    llr[sort_idx] = llr
    return llr[:ntar], llr[ntar:]


def cosine_scoring(x1, x2):
    """hyperion/utils/math.py:349-356."""
    x1 = x1 / np.sqrt(np.sum(x1 ** 2, axis=-1, keepdims=True))
    x2 = x2 / np.sqrt(np.sum(x2 ** 2, axis=-1, keepdims=True))
    return np.dot(x1, x2.T)
