"""Small dense helpers with the call signatures of hyperion/utils/math.py:14-99.  They act on
model matrices (K x K or d x d) on the host in float64; embeddings and scores never pass through
them."""
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
