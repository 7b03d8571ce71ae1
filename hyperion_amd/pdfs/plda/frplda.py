import numpy as np

from ... import engine, fold
from ...hyp_defs import float_cpu
from .plda_base import ClassStats, PLDABase


class FRPLDA(PLDABase):
    """Full-rank PLDA (two-covariance model): y ~ N(mu, B^-1), x ~ N(y, W^-1); B and W are
    [x_dim, x_dim] precisions and y_dim = x_dim (constructor as hyperion/pdfs/plda/frplda.py:14-39).
    """

    _param_names = ("mu", "B", "W")

    def __init__(self, mu=None, B=None, W=None, fullcov_W=True, update_mu=True, update_B=True,
                 update_W=True, **kwargs):
        super().__init__(mu=mu, update_mu=update_mu, **kwargs)
        if mu is not None:
            self.y_dim = mu.shape[0]
        self.B = B
        self.W = W
        self.fullcov_W = fullcov_W
        self.update_B = update_B
        self.update_W = update_W

    def validate(self):
        assert self.mu is not None and self.B is not None and self.W is not None
        assert self.mu.shape[0] == self.B.shape[0]
        assert self.mu.shape[0] == self.B.shape[1]
        assert self.mu.shape[0] == self.W.shape[0]
        assert self.mu.shape[0] == self.W.shape[1]
        self.x_dim = self.mu.shape[0]
        self.y_dim = self.x_dim

    @property
    def is_init(self):
        if not self._is_init and self.mu is not None and self.B is not None and self.W is not None:
            self.validate()
            self._is_init = True
        return self._is_init

    def _params(self):
        return self.mu, self.B, self.W

    def _f64(self):
        f = lambda a: np.asarray(a, dtype=np.float64)  # noqa: E731
        return f(self.mu), f(self.B), f(self.W)

    def _fold(self):
        return fold.fold_frplda(*self._f64())

    def _book_fold(self):
        P, beta, lam, _, muBmu = fold.book_basis_frplda(*self._f64())
        K = P.shape[1]
        return {"P": P, "row_bias": np.zeros(K), "shift_t": 0.5 * beta, "shift_n": beta, "lam": lam,
                "c_extra": 0.5 * muBmu}

    # ---- training (hyperion/pdfs/plda/frplda.py:49-236) ---------------------------------------
    @engine.small_host_algebra
    def initialize(self, D):
        """First model from the statistics (frplda.py:49-75): mu and the between-class covariance
        from the class means, the within-class covariance from the scatter around them
        (S - sum_i N_i m_i m_i^T, which is what the reference's S - F^T m - m^T F + sum_i N_i m_i m_i^T
        equals).  The two Gram matrices over the classes come from the device."""
        D = ClassStats.of(D)
        torch = engine.require_gpu()
        N = D.N
        M = D.F.shape[0]
        self.x_dim = self.y_dim = D.F.shape[1]
        means = D.F / torch.from_numpy(N).cuda()[:, None]
        G, colsum = engine.gram64(means)
        mu = colsum.cpu().numpy() / M
        iB = G.cpu().numpy() / M - np.outer(mu, mu)
        C = D.S_host - engine.gram64(means, N)[0].cpu().numpy()
        iW = 0.5 * (C + C.T) / np.sum(N)
        self.mu, self.B, self.W = mu, _inv_pd(iB), _inv_pd(iW)
        self._is_init = True

    def _posterior(self, D, want_logpy):
        """Device posterior means of the class means.  W Q = B Q diag(lam) with Q^T B Q = I makes
        every L_i = B + N_i W diagonal in the basis Q (L_i^-1 = Q diag(1 / (1 + N_i lam)) Q^T,
        log|L_i| = log|B| + sum_j log(1 + N_i lam_j)), so the reference's loop over distinct counts
        with a Cholesky each (frplda.py:121-146) is two products and one scaling pass.
        Returns (Z, y, logpy, logpy_sum, lam, Q): Z = [F | y] as one device buffer."""
        from scipy import linalg as sla

        torch = engine.require_gpu()
        mu, B, W = self._f64()
        lam, Q = sla.eigh(0.5 * (W + W.T), 0.5 * (B + B.T))
        lam = np.maximum(lam, 0.0)
        M, d = D.F.shape
        Z = torch.empty((M, 2 * d), dtype=torch.float64, device="cuda")
        Z[:, :d] = D.F
        T = engine.gemm64(D.F, np.ascontiguousarray((W @ Q).T), bias=mu @ B @ Q)  # (F W + mu B) Q
        logpy, total = engine.splda_posterior(T, D.N, lam)
        half_logB = np.sum(np.log(np.diag(np.linalg.cholesky(B))))  # 1/2 log|B|
        y = engine.gemm64(T, Q, out=Z[:, d:])  # back to the model's basis
        return Z, y, (logpy + half_logB if want_logpy else None), total + M * half_logB, lam, Q

    @engine.small_host_algebra
    def compute_py_g_x(self, D, return_cov=False, return_logpy_0=False, return_acc=False):
        """Posterior of the class mean of every class given its statistics (frplda.py:77-158):
        y [M, d] and on request the posterior covariances [M, d, d], the log normalisers logpy [M]
        and the accumulators (Ry, Py) — Ry comes back as zeros, as it does there (only Py gathers
        the posterior covariances, frplda.py:145-146).  Host arrays out."""
        D = ClassStats.of(D)
        _, y, logpy, _, lam, Q = self._posterior(D, return_logpy_0)
        r = [y.cpu().numpy().astype(float_cpu())]
        if return_cov:
            uniq, inverse = np.unique(D.N, return_inverse=True)
            cov = np.einsum("ak,uk,bk->uab", Q, 1.0 / (1.0 + uniq[:, None] * lam[None, :]), Q)
            r.append(cov[inverse].astype(float_cpu()))
        if return_logpy_0:
            r.append(logpy.cpu().numpy().astype(float_cpu()))
        if return_acc:
            r += [np.zeros((Q.shape[0], Q.shape[0]), dtype=float_cpu()), self._posterior_cov_sum(D.N, lam, Q)]
        return r[0] if len(r) == 1 else tuple(r)

    @staticmethod
    def _posterior_cov_sum(N, lam, Q):
        """sum_i L_i^-1 from the distinct counts."""
        uniq, cnt = np.unique(N, return_counts=True)
        return (Q * (cnt[:, None] / (1.0 + uniq[:, None] * lam[None, :])).sum(axis=0)) @ Q.T

    @engine.small_host_algebra
    def Estep(self, D):
        """Accumulators of one EM epoch (frplda.py:160-177):
        (N_tot, M, S, sum logpy, sum y, Ry = sum N_i y_i y_i^T, Cy = F^T y, Py).  The products over
        the classes are two Gram passes on the device, over [F | y] and over y weighted by N."""
        D = ClassStats.of(D)
        Z, y, _, total, lam, Q = self._posterior(D, False)
        d = D.F.shape[1]
        G, colsum = engine.gram64(Z)
        G, colsum = G.cpu().numpy(), colsum.cpu().numpy()
        Ry = engine.gram64(y, D.N)[0].cpu().numpy()
        Py = self._posterior_cov_sum(D.N, lam, Q) + G[d:, d:]
        return (np.sum(D.N), D.F.shape[0], D.S_host, float(total.item()), colsum[d:], Ry, G[:d, d:], Py)

    @engine.small_host_algebra
    def elbo(self, stats):
        """Evidence lower bound from the E-step accumulators (frplda.py:179-202)."""
        N, M, S, logpy_x = stats[:4]
        mu, B, W = self._f64()
        logdet = lambda A: 2.0 * np.sum(np.log(np.diag(np.linalg.cholesky(A))))  # noqa: E731
        logpx_y = 0.5 * (-N * self.x_dim * np.log(2 * np.pi) + N * logdet(W) - np.sum(W * S))
        logpy = 0.5 * M * (-self.y_dim * np.log(2 * np.pi) + logdet(B) - (mu @ B) @ mu)
        return logpx_y + logpy - logpy_x

    @engine.small_host_algebra
    def MstepML(self, stats):
        """Maximum-likelihood update of mu, B and W from the accumulators (frplda.py:216-233); host
        algebra on x_dim-sized matrices.  With update_mu off the reference subtracts the same
        un-transposed cross term twice (frplda.py:225-226); so does this."""
        N, M, S, _, y_acc, Ry, Cy, Py = stats
        ybar = y_acc / M
        if self.update_mu:
            self.mu = ybar
        if self.update_B:
            if self.update_mu:
                iB = Py / M - np.outer(self.mu, self.mu)
            else:
                muybar = np.outer(self.mu, ybar)
                iB = Py / M - muybar - muybar + np.outer(self.mu, self.mu)
            self.B = _inv_pd(iB)
        if self.update_W:
            iW = (S - Cy - Cy.T + Ry) / N
            self.W = _inv_pd(iW) if self.fullcov_W else np.diag(1 / np.diag(iW))

    def MstepMD(self, stats):
        """Nothing to do for the two-covariance model (frplda.py:235-236)."""

    # ---- sampling, adaptation (frplda.py:368-410) -----------------------------------------------
    def sample(self, num_classes, num_samples_per_class, rng=None, seed=1024, return_y=False):
        """Host draw with numpy's generator, in the reference's order (frplda.py:368-394)."""
        from scipy import linalg as sla

        assert self.is_init
        if rng is None:
            rng = np.random.RandomState(seed=seed)
        mu, B, W = self._f64()
        chol_Sb = sla.cholesky(_inv_pd(B), lower=False)
        chol_Sw = sla.cholesky(_inv_pd(W), lower=False)
        d = mu.shape[0]
        z = rng.normal(size=(num_classes * num_samples_per_class, d)).astype(float_cpu(), copy=False) @ chol_Sw
        y = rng.normal(size=(num_classes, d)).astype(float_cpu(), copy=False) @ chol_Sb + mu
        y = np.repeat(y, num_samples_per_class, axis=0)
        return (y + z, y) if return_y else y + z

    def weighted_avg_params(self, mu, B, W, w_mu, w_B, w_W):
        """Interpolation towards (mu, B, W) on the covariances (frplda.py:396-407)."""
        self.weigthed_avg_params(mu, w_mu)
        _, B0, W0 = self._f64()
        if w_B > 0:
            self.B = _inv_pd(w_B * _inv_pd(np.asarray(B, dtype=np.float64)) + (1 - w_B) * _inv_pd(B0))
        if w_W > 0:
            self.W = _inv_pd(w_W * _inv_pd(np.asarray(W, dtype=np.float64)) + (1 - w_W) * _inv_pd(W0))

    def weighted_avg_model(self, plda, w_mu, w_B, w_W):
        self.weighted_avg_params(plda.mu, plda.B, plda.W, w_mu, w_B, w_W)

    def get_config(self):
        config = {"update_W": self.update_W, "update_B": self.update_B, "fullcov_W": self.fullcov_W}
        base_config = super().get_config()
        return dict(list(base_config.items()) + list(config.items()))


def _inv_pd(A):
    """Inverse of a positive definite matrix through its Cholesky factor, upper triangle read
    (hyperion/utils/math.py:38-57 with return_inv=True)."""
    from scipy import linalg as sla

    return sla.cho_solve((sla.cholesky(A, lower=False), False), np.eye(A.shape[0]))
