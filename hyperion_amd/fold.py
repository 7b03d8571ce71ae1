"""Host-side (float64) folding of transform chains and PLDA models into the few small matrices the
device kernels consume.  Everything here acts on model parameters only (d x d or K x K), never on
embeddings or scores.

Closed form used by all three PLDA flavours (reference: hyperion/pdfs/plda/splda.py:275-316,
frplda.py:258-303, plda.py:430-472):

    g(x) = x At + bt      (gamma_tar)          h(x) = x An + bn      (gamma_non)
    q(x) = (|g|^2 - |h|^2) / 2
    llr(x1, x2) = g(x1) . g(x2) + q(x1) + q(x2) + c
"""
import numpy as np
import scipy.linalg as sla


def _chol_upper(A):
    return sla.cholesky(np.asarray(A, dtype=np.float64), lower=False)


def _right_solve(R, X):
    """X @ inv(R), R upper triangular (hyperion/utils/math.py:81-82)."""
    return sla.solve_triangular(R.T, X.T, lower=True).T


def _logdet_from_chol(R):
    return 2.0 * np.sum(np.log(np.diag(R)))


# ----------------------------------------------------------------------------- affine chains
def compose(a1, a2):
    """(A1, b1) then (A2, b2):  (x A1 + b1) A2 + b2.  None stands for identity / zero."""
    A1, b1 = a1
    A2, b2 = a2
    if A1 is None:
        A = A2
    elif A2 is None:
        A = A1
    else:
        A = A1 @ A2
    if b1 is None:
        b = b2
    else:
        b = b1 if A2 is None else b1 @ A2
        if b2 is not None:
            b = b + b2
    return A, b


def affine_from_mu_T(mu, T):
    """x -> (x - mu) T with T a matrix, a diagonal (1-D) or None (cent_whiten.py:25-33,
    lda.py:31-34, pca.py:39-42)."""
    A = None
    if T is not None:
        T = np.asarray(T, dtype=np.float64)
        A = np.diag(T) if T.ndim == 1 else T
    b = None
    if mu is not None:
        mu = np.asarray(mu, dtype=np.float64)
        b = -mu if A is None else -(mu @ A)
    return A, b


def fold_chain(steps, d_in):
    """steps: list of (A, b, lnorm) in application order.  Returns (segments, tail, d_out):
    segments = [(A, b, d_out)] each closed by a length normalisation, tail = (A, b) affine left
    over after the last normalisation."""
    segments = []
    cur = (None, None)
    d = d_in
    for A, b, lnorm in steps:
        if A is not None:
            assert A.shape[0] == d, "transform expects %d-dim input, got %d" % (A.shape[0], d)
            d = A.shape[1]
        if b is not None:
            assert b.shape[0] == d
        cur = compose(cur, (A, b))
        if lnorm:
            segments.append((cur[0], cur[1], d))
            cur = (None, None)
    return segments, cur, d


# ----------------------------------------------------------------------------- PLDA folds
def fold_from_WV(mu, WV, VV):
    """SPLDA / PLDA: At = WV Rtar^-1, An = WV Rnon^-1 with Rtar = chol(I + 2 VV),
    Rnon = chol(I + VV); biases -mu At, -mu An; c = logLnon - logLtar / 2
    (splda.py:281-315)."""
    K = VV.shape[0]
    I = np.eye(K)
    Rnon = _chol_upper(I + VV)
    Rtar = _chol_upper(I + 2 * VV)
    At = _right_solve(Rtar, WV)
    An = _right_solve(Rnon, WV)
    c = _logdet_from_chol(Rnon) - 0.5 * _logdet_from_chol(Rtar)
    return At, -(mu @ At), An, -(mu @ An), float(c)


def splda_WV(V, W):
    WV = W @ V.T
    return WV, V @ WV


def plda_WV(V, U, D):
    """Woodbury form of the within-class precision applied to V (plda.py:77-89)."""
    DV = V * D
    DU = U * D
    J = V @ DU.T
    Lz = np.eye(U.shape[0]) + DU @ U.T
    Rz = _chol_upper(Lz)
    DUiLz = sla.cho_solve((Rz, False), DU).T
    VW = DV.T - DUiLz @ J.T
    return VW, V @ VW


def fold_frplda(mu, B, W):
    """frplda.py:262-301: g = (x W + mu B / 2) Rtar^-1, h = (x W + mu B) Rnon^-1."""
    Rnon = _chol_upper(B + W)
    Rtar = _chol_upper(B + 2 * W)
    Bmu = mu @ B
    At = _right_solve(Rtar, W)
    An = _right_solve(Rnon, W)
    bt = _right_solve(Rtar, (0.5 * Bmu)[None, :])[0]
    bn = _right_solve(Rnon, Bmu[None, :])[0]
    c = (_logdet_from_chol(Rnon) - 0.5 * _logdet_from_chol(Rtar)
         - 0.5 * _logdet_from_chol(_chol_upper(B)) + 0.5 * float(Bmu @ mu))
    return At, bt, An, bn, float(c)


# ----------------------------------------------------------------------------- by-the-book folds
def book_basis_WV(WV, VV):
    """Eigen-basis in which every L(n) = I + n VV is diagonal: VV = E diag(lam) E^T.
    Returns (P = WV E, lam): with s = F~ P the statistics of a class with n segments give
    |gamma(n)|^2 = sum s_k^2 / (1 + n lam_k)  and  logdet L(n) = sum log(1 + n lam_k)
    (splda.py:334-375)."""
    lam, E = np.linalg.eigh(0.5 * (VV + VV.T))
    return WV @ E, lam


def book_basis_frplda(mu, B, W):
    """Generalised eigen-basis W u = lam B u (U^T B U = I, U^T W U = diag(lam)), in which
    (B + n W)^-1 = U diag(1 / (1 + n lam)) U^T and logdet(B + n W) = logdet B + sum log(1 + n lam)
    (frplda.py:320-364).  Returns (P = W U, beta = (mu B) U, lam, logdetB, muBmu)."""
    lam, U = sla.eigh(0.5 * (W + W.T), 0.5 * (B + B.T))
    Bmu = mu @ B
    return W @ U, Bmu @ U, lam, _logdet_from_chol(_chol_upper(B)), float(Bmu @ mu)
