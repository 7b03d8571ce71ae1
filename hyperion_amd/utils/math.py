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
