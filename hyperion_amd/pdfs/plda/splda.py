import numpy as np

from ... import fold
from ... import engine
from ...hyp_defs import float_cpu
from .plda_base import ClassStats, PLDABase


class SPLDA(PLDABase):
    """Simplified PLDA: x = mu + y V + e, e ~ N(0, W^-1).  V is [y_dim, x_dim], W the within-class
    precision [x_dim, x_dim] (constructor as hyperion/pdfs/plda/splda.py:13-39)."""

    _param_names = ("mu", "V", "W")

    def __init__(self, y_dim=None, mu=None, V=None, W=None, fullcov_W=True, update_mu=True,
                 update_V=True, update_W=True, **kwargs):
        super().__init__(y_dim=y_dim, mu=mu, update_mu=update_mu, **kwargs)
        if V is not None:
            self.y_dim = V.shape[0]
        self.V = V
        self.W = W
        self.fullcov_W = fullcov_W
        self.update_V = update_V
        self.update_W = update_W

    def validate(self):
        assert self.mu is not None and self.V is not None and self.W is not None
        assert self.mu.shape[0] >= self.V.shape[0]
        assert self.mu.shape[0] == self.V.shape[1]
        assert self.mu.shape[0] == self.W.shape[0]
        assert self.mu.shape[0] == self.W.shape[1]
        self.x_dim = self.mu.shape[0]

    def _validate_for_scoring(self):
        """Shapes only: a model projected onto fewer dimensions than y_dim (project() after a PCA on
        a short recording) scores in the reference, whose llr_1vs1 never validates."""
        assert self.mu is not None and self.V is not None and self.W is not None
        d = self.mu.shape[0]
        assert self.V.shape[1] == d and self.W.shape[0] == d and self.W.shape[-1] == d
        self.x_dim = d

    @property
    def is_init(self):
        if not self._is_init and self.mu is not None and self.V is not None and self.W is not None:
            self.validate()
            self._is_init = True
        return self._is_init

    def _params(self):
        return self.mu, self.V, self.W

    def _WV(self):
        V = np.asarray(self.V, dtype=np.float64)
        W = np.asarray(self.W, dtype=np.float64)
        if W.ndim == 1:
            W = np.diag(W)
        return fold.splda_WV(V, W)

    def _fold(self):
        WV, VV = self._WV()
        return fold.fold_from_WV(np.asarray(self.mu, dtype=np.float64), WV, VV)

    def _book_fold(self):
        WV, VV = self._WV()
        P, lam = fold.book_basis_WV(WV, VV)
        mu = np.asarray(self.mu, dtype=np.float64)
        K = P.shape[1]
        return {"P": P, "row_bias": -(mu @ P), "shift_t": np.zeros(K), "shift_n": np.zeros(K),
                "lam": lam, "c_extra": 0.0}

    def project(self, T, delta_mu=None):
        """The model of y = (x - delta_mu) T, e.g. after a PCA fitted on the recording that is being
        diarized (hyperion/pdfs/plda/splda.py:421-431, called from
        hyperion/diarization/diar_ahc_plda.py:131-141).  Host arithmetic on d x d matrices; this
        object is left unchanged (the reference subtracts delta_mu from its own mu in place)."""
        from scipy import linalg as sla

        T = np.asarray(T, dtype=np.float64)
        mu = np.asarray(self.mu, dtype=np.float64)
        if delta_mu is not None:
            mu = mu - np.asarray(delta_mu, dtype=np.float64)
        W = np.asarray(self.W, dtype=np.float64)
        if W.ndim == 1:
            W = np.diag(W)
        Sw = sla.cho_solve(sla.cho_factor(W), np.eye(W.shape[0]))
        Sw_p = T.T @ Sw @ T
        W_p = sla.cho_solve(sla.cho_factor(Sw_p), np.eye(Sw_p.shape[0]))
        return SPLDA(mu=mu @ T, V=np.asarray(self.V, dtype=np.float64) @ T, W=0.5 * (W_p + W_p.T),
                     fullcov_W=True)

    # ---- training (hyperion/pdfs/plda/splda.py:50-242) ----------------------------------------
    def _W_full(self):
        W = np.asarray(self.W, dtype=np.float64)
        return np.diag(W) if W.ndim == 1 else W

    @engine.small_host_algebra
    def initialize(self, D):
        """First model from the statistics (splda.py:50-71): mu = mean of the class means, V = the
        leading principal directions of the class means scaled by their singular values, W from
        the scatter that the class means leave.  The singular vectors come from the d x d Gram
        matrix of the centred class means (device) instead of an SVD of the [classes, d] matrix:
        the rows of V can differ from the reference's in sign, which no score, posterior
        covariance or ELBO depends on."""
        from scipy import linalg as sla

        D = ClassStats.of(D)
        torch = engine.require_gpu()
        N = D.N
        self.x_dim = D.F.shape[1]
        Nd = torch.from_numpy(N).cuda()
        means = D.F / Nd[:, None]
        mu = means.mean(dim=0)
        G = engine.gram64(means - mu)[0].cpu().numpy()
        ev, U = sla.eigh(G)
        ev, U = ev[::-1][:self.y_dim], U[:, ::-1][:, :self.y_dim]
        V = np.sqrt(np.maximum(ev, 0.0))[:, None] * U.T
        C = (D.S_host - engine.gram64(means, N)[0].cpu().numpy()) / np.sum(N)
        if self.fullcov_W:
            W = sla.cho_solve(sla.cho_factor(C), np.eye(C.shape[0]))
        else:
            W = 1 / np.diag(C)
        self.mu, self.V, self.W = mu.cpu().numpy(), V, W

    @engine.small_host_algebra
    def Estep(self, D):
        """Accumulators of one EM epoch (splda.py:160-179):
        (N_tot, M, F_tot, S, sum logpy, sum y, sum N y, Ry, Cy = F^T y, Py).  The products over the
        classes are two Gram passes on the device, over [F | y] and over y weighted by N."""
        D = ClassStats.of(D)
        Z, y, _, total, lam, Q = self._posterior(D, False)
        d = D.F.shape[1]
        G, colsum = engine.gram64(Z)
        G, colsum = G.cpu().numpy(), colsum.cpu().numpy()
        GN, Ry1 = engine.gram64(y, D.N)
        Ry, Py = self._posterior_cov_sums(D.N, lam, Q)
        Ry = Ry + GN.cpu().numpy()
        Py = Py + G[d:, d:]
        return (np.sum(D.N), D.F.shape[0], colsum[:d], D.S_host, float(total.item()), colsum[d:],
                Ry1.cpu().numpy(), Ry, G[:d, d:], Py)

    @engine.small_host_algebra
    def elbo(self, stats):
        """Evidence lower bound from the E-step accumulators (splda.py:181-196)."""
        N, M, F, S, logpy_x = stats[:5]
        W, mu = self._W_full(), np.asarray(self.mu, dtype=np.float64)
        logW = 2.0 * np.sum(np.log(np.diag(np.linalg.cholesky(W))))
        Fmu = np.outer(F, mu)
        Shat = S - Fmu - Fmu.T + N * np.outer(mu, mu)
        logpx_y = 0.5 * (-N * self.x_dim * np.log(2 * np.pi) + N * logW - np.sum(W * Shat))
        return logpx_y - 0.5 * M * self.y_dim * np.log(2 * np.pi) - logpy_x

    @engine.small_host_algebra
    def MstepML(self, stats):
        """Maximum-likelihood update of (V, mu) and W from the accumulators (splda.py:198-230);
        host algebra on y_dim- and x_dim-sized matrices."""
        from scipy import linalg as sla

        N, M, F, S, _, y_acc, Ry1, Ry, Cy, Py = stats
        Rytilde = np.block([[Ry, Ry1[:, None]], [Ry1[None, :], np.array([[N]])]])
        Cytilde = np.hstack((Cy, F[:, None]))
        joint = self.update_mu and self.update_V
        if self.update_mu and not self.update_V:
            self.mu = (F - Ry1 @ self.V) / N
        if self.update_V and not self.update_mu:
            self.V = sla.cho_solve(sla.cho_factor(Ry), Cy.T - np.outer(Ry1, self.mu))
        if joint:
            Vtilde = sla.cho_solve(sla.cho_factor(Rytilde), Cytilde.T)
            self.V, self.mu = Vtilde[:-1, :], Vtilde[-1, :]
        if self.update_W:
            if joint:
                iW = (S - Cy @ self.V - np.outer(F, self.mu)) / N
            else:
                Vtilde = np.vstack((self.V, self.mu))
                CVt = Cytilde @ Vtilde
                iW = (S - CVt - CVt.T + Vtilde.T @ Rytilde @ Vtilde) / N
            if self.fullcov_W:
                self.W = sla.cho_solve(sla.cho_factor(iW), np.eye(iW.shape[0]))
            else:
                self.W = np.diag(1 / np.diag(iW))

    @engine.small_host_algebra
    def MstepMD(self, stats):
        """Minimum-divergence update: the factor prior is moved back to N(0, I) (splda.py:232-242)."""
        from scipy import linalg as sla

        N, M, F, S, _, y_acc, Ry1, Ry, Cy, Py = stats
        mu_y = y_acc / M
        if self.update_mu:
            self.mu = self.mu + mu_y @ self.V
        if self.update_V:
            self.V = sla.cholesky(Py / M - np.outer(mu_y, mu_y), lower=False) @ self.V

    # ---- sampling, adaptation (splda.py:381-419) ------------------------------------------------
    def log_probx_g_y(self, x, y):
        """log p(x_i | y_i) of every row, x [n, x_dim], y [n, y_dim] (splda.py:264-273):
        (-d log 2 pi + log|W| - (x - mu - y V) W (x - mu - y V)^T) / 2.  Residual and quadratic form
        on the device in float64, log|W| from the host Cholesky factor."""
        from ...utils.math import logdet_pdmat

        W = self._W_full()
        logp = self._row_logp(self._residual(x, (y, self.V)), W, logdet_pdmat(W))
        return engine.to_host(logp, float_cpu())

    def sample(self, num_classes, num_samples_per_class, rng=None, seed=1024):
        """Host draw with numpy's generator, in the reference's order (splda.py:381-399)."""
        from scipy import linalg as sla

        if rng is None:
            rng = np.random.RandomState(seed=seed)
        W, V, mu = self._W_full(), np.asarray(self.V, dtype=np.float64), np.asarray(self.mu, dtype=np.float64)
        chol_Sw = sla.cholesky(_inv_pd(W), lower=False)
        d = mu.shape[0]
        z = rng.normal(size=(num_classes * num_samples_per_class, d)).astype(float_cpu(), copy=False) @ chol_Sw
        y = rng.normal(size=(num_classes, self.y_dim)).astype(float_cpu(), copy=False) @ V + mu
        return np.repeat(y, num_samples_per_class, axis=0) + z

    def weighted_avg_params(self, mu, V, W, w_mu, w_B, w_W):
        """Interpolation towards (mu, V, W): between-class covariance V^T V refactored to y_dim
        directions, within-class covariance W^-1 (splda.py:401-416)."""
        from scipy import linalg as sla

        self.weigthed_avg_params(mu, w_mu)
        if w_B > 0:
            V0, V = np.asarray(self.V, dtype=np.float64), np.asarray(V, dtype=np.float64)
            w, U = sla.eigh(w_B * (V.T @ V) + (1 - w_B) * (V0.T @ V0))
            self.V = (np.sqrt(w[-self.y_dim:]) * U[:, -self.y_dim:]).T
        if w_W > 0:
            W = np.asarray(W, dtype=np.float64)
            W = np.diag(W) if W.ndim == 1 else W
            self.W = _inv_pd(w_W * _inv_pd(W) + (1 - w_W) * _inv_pd(self._W_full()))

    def weighted_avg_model(self, plda, w_mu, w_B, w_W):
        self.weighted_avg_params(plda.mu, plda.V, plda.W, w_mu, w_B, w_W)

    def get_config(self):
        config = {"update_W": self.update_W, "update_V": self.update_V, "fullcov_W": self.fullcov_W}
        base_config = super().get_config()
        return dict(list(base_config.items()) + list(config.items()))


def _inv_pd(A):
    """Inverse of a positive definite matrix through its Cholesky factor, upper triangle read
    (hyperion/utils/math.py:38-57 with return_inv=True)."""
    from scipy import linalg as sla

    return sla.cho_solve((sla.cholesky(A, lower=False), False), np.eye(A.shape[0]))
