"""Small dense helpers with the call signatures of hyperion/utils/math.py:14-99.  They act on
model matrices (K x K or d x d) on the host in float64; embeddings and scores never pass through
them - except cosine_scoring, which is a scoring back-end of its own and runs on the device."""
import numpy as np
import scipy.linalg as la


def logdet_pdmat(A):
    """log det of a positive definite matrix through its Cholesky factor."""
    assert A.shape[0] == A.shape[1]
    return 2.0 * np.sum(np.log(np.diag(la.cholesky(A))))


def invert_pdmat(A, right_inv=False, return_logdet=False, return_inv=False):
    """Returns [f, R, (logdet), (inv)] with f(v) = A^-1 v (or v A^-1 when right_inv) and R the
    upper Cholesky factor of A."""
    assert A.shape[0] == A.shape[1]
    R = la.cholesky(A, lower=False)
    if right_inv:
        def f(x):
            return la.cho_solve((R, False), x.T).T
    else:
        def f(x):
            return la.cho_solve((R, False), x)
    out = [f, R]
    if return_logdet:
        out.append(2.0 * np.sum(np.log(np.diag(R))))
    if return_inv:
        out.append(f(np.eye(A.shape[0])))
    return out


def invert_trimat(A, lower=False, right_inv=False, return_logdet=False, return_inv=False):
    """Returns f (or [f, (logdet), (inv)]) with f(v) = A^-1 v (or v A^-1) for triangular A."""
    if right_inv:
        def f(x):
            return la.solve_triangular(A.T, x.T, lower=not lower).T
    else:
        def f(x):
            return la.solve_triangular(A, x, lower=lower)
    if not (return_logdet or return_inv):
        return f
    out = [f]
    if return_logdet:
        out.append(np.sum(np.log(np.diag(A))))
    if return_inv:
        out.append(f(np.eye(A.shape[0])))
    return out


# ---- the rest of hyperion/utils/math.py:102-347: small host helpers on model-sized arrays --------------
def softmax(r, axis=-1):
    """exp(r) / sum exp(r) along axis, shifted by the maximum (math.py:102-110)."""
    e = np.exp(r - np.max(r, axis=axis, keepdims=True))
    return e / np.sum(e, axis=axis, keepdims=True)


def logsumexp(r, axis=-1):
    """log sum exp(r) along axis, shifted by the maximum; 1e-20 inside the log as the reference has it
    (math.py:113-120)."""
    m = np.max(r, axis=axis, keepdims=True)
    return np.log(np.sum(np.exp(r - m), axis=axis) + 1e-20) + np.squeeze(m, axis=axis)


def _below(x, y, value_of):
    """y with the entries where x < -100 replaced (exp(-x) has overflowed there)."""
    far = x < -100
    y[far] = value_of(x[far])
    return y


def logsigmoid(x):
    """log(1 / (1 + exp(-x))), = x far below zero (math.py:123-132)."""
    with np.errstate(over="ignore"):
        return _below(x, -np.log(1 + np.exp(-x)), lambda v: v)


def neglogsigmoid(x):
    """-log sigmoid(x) (math.py:135-144)."""
    with np.errstate(over="ignore"):
        return _below(x, np.log(1 + np.exp(-x)), lambda v: -v)


def sigmoid(x):
    """1 / (1 + exp(-x)), exactly 0 far below zero (math.py:147-156)."""
    with np.errstate(over="ignore"):
        return _below(x, 1 / (1 + np.exp(-x)), lambda v: 0.0)


def fisher_ratio(mu1, Sigma1, mu2, Sigma2):
    """(mu1 - mu2)^T (Sigma1 + Sigma2)^-1 (mu1 - mu2) (math.py:159-166)."""
    delta = mu1 - mu2
    return np.inner(delta, la.cho_solve(la.cho_factor(Sigma1 + Sigma2), delta))


def fisher_ratio_with_precs(mu1, Lambda1, mu2, Lambda2):
    """The same from the precision matrices (math.py:169-176)."""
    inv = lambda L: la.cho_solve(la.cho_factor(L), np.eye(L.shape[0]))  # noqa: E731
    return fisher_ratio(mu1, inv(Lambda1), mu2, inv(Lambda2))


def symmat2vec(A, lower=False, diag_factor=None):
    """Upper (or lower) triangle of A, row by row, the diagonal scaled by diag_factor
    (math.py:179-196)."""
    n = A.shape[0]
    if diag_factor is not None:
        A = A + (diag_factor - 1.0) * np.diag(np.diag(A))
    return A[np.tril_indices(n)] if lower else A[np.triu_indices(n)]


def _dim_of_triangle(v):
    return int((-1 + np.sqrt(1 + 8 * v.shape[0])) / 2)


def vec2trimat(v, lower=False):
    """Triangular matrix whose triangle, row by row, is v (math.py:242-259)."""
    from ..hyp_defs import float_cpu

    n = _dim_of_triangle(v)
    A = np.zeros((n, n), dtype=float_cpu())
    A[np.tril_indices(n) if lower else np.triu_indices(n)] = v
    return A


def vec2symmat(v, lower=False, diag_factor=None):
    """Symmetric matrix from one triangle; diag_factor scales the diagonal (math.py:199-224)."""
    T = vec2trimat(v, lower)
    A = T + T.T - np.diag(np.diag(T))
    if diag_factor is not None:
        A[np.diag_indices(A.shape[0])] *= diag_factor
    return A


def trimat2vec(A, lower=False):
    """math.py:227-239."""
    return symmat2vec(A, lower)


def _floor_unit_spectrum(T):
    """T with its singular values below 1 raised to 1 (T symmetric positive semi-definite)."""
    u, d, _ = la.svd(T, full_matrices=False)
    return np.dot(u * np.maximum(d, 1.0), u.T)


def fullcov_varfloor(S, F, F_is_chol=False, lower=False):
    """Covariance S floored by the covariance F (or the scalar F): in the coordinates that whiten F no
    direction of S may have variance below 1 (math.py:262-294).  F may be given by its Cholesky
    factor (upper unless lower=True)."""
    if not isinstance(F, np.ndarray):
        return F * _floor_unit_spectrum(S / F)
    R = la.cholesky(F, lower=False) if not F_is_chol else (F.T if lower else F)
    iR = la.solve_triangular(R, np.eye(R.shape[0]), lower=False)
    return np.dot(R.T, np.dot(_floor_unit_spectrum(np.dot(np.dot(iR.T, S), iR)), R))


def fullcov_varfloor_from_cholS(cholS, cholF, lower=False):
    """The same with S given and returned as Cholesky factors (math.py:297-327)."""
    if lower:
        cholS = cholS.T
    if isinstance(cholF, np.ndarray):
        R = cholF.T if lower else cholF
        T = np.dot(cholS, la.solve_triangular(R, np.eye(R.shape[0]), lower=False))
        S = np.dot(R.T, np.dot(_floor_unit_spectrum(np.dot(T.T, T)), R))
    else:
        T = cholS / cholF
        S = (cholF ** 2) * _floor_unit_spectrum(np.dot(T.T, T))
    return la.cholesky(S, lower=lower)


def int2onehot(class_ids, num_classes=None):
    """[n, num_classes] indicator matrix of integer class ids (math.py:330-346)."""
    from ..hyp_defs import float_cpu

    class_ids = np.asarray(class_ids)
    if num_classes is None:
        num_classes = np.max(class_ids) + 1
    p = np.zeros((len(class_ids), num_classes), dtype=float_cpu())
    p[np.arange(len(class_ids)), class_ids] = 1
    return p


def cosine_scoring(x1, x2):
    """Cosine of the angle between every row of x1 and every row of x2 (hyperion/utils/math.py:
    349-356).  On the device: rows scaled to unit length by the length-normalisation kernel, zero
    padded to the score kernels' row width, one matrix product.  numpy in gives a `float_cpu()` array
    out, device tensors in a device fp32 matrix; a zero row gives zeros (reference: NaN from 0/0)."""
    from .. import engine
    from ..hyp_defs import float_cpu
    from ..transforms.base import chain_plan
    from ..transforms.lnorm import LNorm

    torch = engine.require_gpu()
    on_device = isinstance(x1, torch.Tensor) and isinstance(x2, torch.Tensor)
    d = x1.shape[1]
    if x2.shape[1] != d:
        raise ValueError("dimensions differ: %d and %d" % (d, x2.shape[1]))
    ln = _unit.setdefault("lnorm", LNorm())
    plan = chain_plan([ln], d, ln)
    kp = engine.round_up(d, 8)

    def unit_rows(x):
        y = plan.transform(engine.upload(x))  # sqrt(d) x / |x|
        G = torch.zeros((y.shape[0], kp), dtype=torch.float32, device=y.device)
        G[:, :d] = y[:, :d] * (1.0 / np.sqrt(d))
        return G

    S = engine.score(unit_rows(x1), unit_rows(x2))
    return S if on_device else engine.to_host(S, float_cpu())


_unit = {}
