import numpy as np

from ... import engine, fold
from ...hyp_defs import float_cpu
from .plda_base import ClassStats, PLDABase


class PLDA(PLDABase):
    """Full PLDA: x = mu + y V + z U + e, e ~ N(0, diag(D)^-1).  V [y_dim, x_dim], U [z_dim, x_dim],
    D [x_dim] diagonal precision (constructor as hyperion/pdfs/plda/plda.py:14-59).  The channel
    subspace only enters through the Woodbury form of the within-class precision, which the host
    folds into the same two projections SPLDA uses."""

    _param_names = ("mu", "V", "U", "D")

    def __init__(self, y_dim=None, z_dim=None, mu=None, V=None, U=None, D=None, floor_iD=1e-5,
                 update_mu=True, update_V=True, update_U=True, update_D=True, **kwargs):
        super().__init__(y_dim=y_dim, mu=mu, update_mu=update_mu, **kwargs)
        self.z_dim = z_dim
        if V is not None:
            self.y_dim = V.shape[0]
        if U is not None:
            self.z_dim = U.shape[0]
        self.V = V
        self.U = U
        self.D = D
        self.floor_iD = floor_iD
        self.update_V = update_V
        self.update_U = update_U
        self.update_D = update_D

    def validate(self):
        assert self.mu is not None and self.V is not None and self.U is not None and self.D is not None
        assert self.mu.shape[0] >= self.V.shape[0]
        assert self.mu.shape[0] == self.V.shape[1]
        assert self.mu.shape[0] >= self.U.shape[0]
        assert self.mu.shape[0] == self.U.shape[1]
        assert self.mu.shape[0] == self.D.shape[0]
        self.x_dim = self.mu.shape[0]

    @property
    def is_init(self):
        if not self._is_init and all(a is not None for a in (self.mu, self.V, self.U, self.D)):
            self.validate()
            self._is_init = True
        return self._is_init

    def _params(self):
        return self.mu, self.V, self.U, self.D

    def _WV(self):
        f = lambda a: np.asarray(a, dtype=np.float64)  # noqa: E731
        return fold.plda_WV(f(self.V), f(self.U), f(self.D))

    def _fold(self):
        WV, VV = self._WV()
        return fold.fold_from_WV(np.asarray(self.mu, dtype=np.float64), WV, VV)

    def _book_fold(self):
        # the reference's PLDA.llr_NvsM_book reads an attribute that is never set (plda.py:481);
        # the evident intent (the Woodbury product computed by compute_aux) is implemented
        WV, VV = self._WV()
        P, lam = fold.book_basis_WV(WV, VV)
        mu = np.asarray(self.mu, dtype=np.float64)
        K = P.shape[1]
        return {"P": P, "row_bias": -(mu @ P), "shift_t": np.zeros(K), "shift_n": np.zeros(K),
                "lam": lam, "c_extra": 0.0}

    # ---- training (hyperion/pdfs/plda/plda.py:77-394) ------------------------------------------
    def _aux(self):
        """The auxiliary values of compute_aux (plda.py:77-89) from the current parameters: DU, J,
        the upper Cholesky factor of Lz = I + (U D) U^T, log|Lz| and the within-class precision W."""
        from scipy import linalg as sla

        f = lambda a: np.asarray(a, dtype=np.float64)  # noqa: E731
        V, U, D = f(self.V), f(self.U), f(self.D)
        DU = U * D
        Rz = sla.cholesky(np.eye(U.shape[0]) + DU @ U.T, lower=False)
        DUiLz = sla.cho_solve((Rz, False), DU).T
        return {"DU": DU, "J": V @ DU.T, "Rz": Rz, "log_Lz": 2 * np.sum(np.log(np.diag(Rz))),
                "W": np.diag(D) - DUiLz @ DU}

    @engine.small_host_algebra
    def initialize(self, D):
        """First model from the statistics (plda.py:91-117): mu and V as for SPLDA (the singular
        vectors from the Gram matrix of the centred class means on the device: rows of V up to
        sign), U from the leading eigenvectors of the residual scatter, D from what is left on its
        diagonal, floored."""
        from scipy import linalg as sla

        D = ClassStats.of(D)
        torch = engine.require_gpu()
        N = D.N
        self.x_dim = D.F.shape[1]
        means = D.F / torch.from_numpy(N).cuda()[:, None]
        mu = means.mean(dim=0)
        G = engine.gram64(means - mu)[0].cpu().numpy()
        ev, E = sla.eigh(G)
        ev, E = ev[::-1][:self.y_dim], E[:, ::-1][:, :self.y_dim]
        V = np.sqrt(np.maximum(ev, 0.0))[:, None] * E.T
        C = (D.S_host - engine.gram64(means, N)[0].cpu().numpy()) / np.sum(N)
        w, E = sla.eigh(C)
        U = np.fliplr(E * np.sqrt(w))[:, :self.z_dim].T
        iD = np.diag(C - U.T @ U).copy()
        iD[iD < self.floor_iD] = self.floor_iD
        self.mu, self.V, self.U, self.D = mu.cpu().numpy(), V, U, 1 / iD

    @engine.small_host_algebra
    def Estep(self, D):
        """Accumulators of one EM epoch (plda.py:200-264): the speaker-factor ones from the device
        (posterior in the eigenbasis of V W V^T, Gram passes over [F | y] and over y weighted by
        N), the channel-factor ones (Rz1, Rz, Ryz, Cz) by host algebra on those.  In Cz the
        reference subtracts the inner product of the summed first-order statistic and mu from
        every entry of S (np.dot of two vectors, plda.py:214); so does this."""
        from scipy import linalg as sla

        D = ClassStats.of(D)
        Z, y, _, total, lam, Q = self._posterior(D, False)
        d = D.F.shape[1]
        G, colsum = engine.gram64(Z)
        G, colsum = G.cpu().numpy(), colsum.cpu().numpy()
        GN, Ry1 = engine.gram64(y, D.N)
        Ry1 = Ry1.cpu().numpy()
        Ry, Py = self._posterior_cov_sums(D.N, lam, Q)
        Ry = Ry + GN.cpu().numpy()
        Py = Py + G[d:, d:]
        Cy, F_tot, S, N_tot = G[:d, d:], colsum[:d], D.S_host, np.sum(D.N)
        mu = np.asarray(self.mu, dtype=np.float64)
        aux = self._aux()
        DU, J = aux["DU"], aux["J"]
        mult_iLz = lambda X: sla.cho_solve((aux["Rz"], False), X.T).T  # noqa: E731
        Cz = mult_iLz((S - F_tot @ mu) @ DU.T - Cy @ J)
        Fmu = np.outer(F_tot, mu)
        Sc = S - Fmu - Fmu.T + N_tot * np.outer(mu, mu)
        Rz1 = mult_iLz((F_tot - N_tot * mu) @ DU.T - Ry1 @ J)
        Cbary = Cy - np.outer(mu, Ry1)
        Ryz = mult_iLz(Cbary.T @ DU.T - Ry @ J)
        A = DU @ Cbary @ J
        B = mult_iLz(DU @ Sc @ DU.T - A - A.T + J.T @ Ry @ J)
        Rz = mult_iLz(B.T).T + N_tot * mult_iLz(np.eye(DU.shape[0]))
        return (N_tot, D.F.shape[0], F_tot, S, float(total.item()), colsum[d:], Ry1, Ry, Cy, Py, Rz1, Rz, Ryz, Cz)

    @engine.small_host_algebra
    def elbo(self, stats):
        """Evidence lower bound from the E-step accumulators (plda.py:266-282)."""
        N, M, F, S, logpy_x = stats[:5]
        mu = np.asarray(self.mu, dtype=np.float64)
        aux = self._aux()
        Fmu = np.outer(F, mu)
        Shat = S - Fmu - Fmu.T + N * np.outer(mu, mu)
        logD = np.sum(np.log(np.asarray(self.D, dtype=np.float64)))
        logpx_y = 0.5 * (-N * self.x_dim * np.log(2 * np.pi) + N * (logD - aux["log_Lz"]) - np.sum(aux["W"] * Shat))
        return logpx_y - 0.5 * M * self.y_dim * np.log(2 * np.pi) - logpy_x

    @engine.small_host_algebra
    def MstepML(self, stats):
        """Maximum-likelihood update of (V, U, mu) and D from the accumulators, branch by branch as
        the reference writes it (plda.py:284-362), the partial-update branches included; host
        algebra on (y_dim + z_dim + 1)- and x_dim-sized matrices."""
        from scipy import linalg as sla

        N, M, F, S, _, y_acc, Ry1, Ry, Cy, Py, Rz1, Rz, Ryz, Cz = stats
        f = lambda a: np.asarray(a, dtype=np.float64)  # noqa: E731
        mu, V, U = f(self.mu), f(self.V), f(self.U)
        solve = lambda A, C: sla.cho_solve((sla.cholesky(A, lower=False), False), C)  # noqa: E731
        upd = (bool(self.update_mu), bool(self.update_V), bool(self.update_U))
        if upd == (True, False, False):
            mu = (F - Ry1 @ V - Rz1 @ U) / N
        if upd == (False, True, False):
            V = solve(Ry, Cy.T - np.outer(Ry1, mu) - Ryz @ U)
        if upd == (False, False, True):
            U = solve(Rz, Cz.T - Ryz.T @ V - np.outer(Rz1, mu))
        if upd == (False, True, True):
            Vtilde = solve(np.block([[Ry, Ryz], [Ryz.T, Rz]]),
                           np.vstack((Cy.T - np.outer(Ry1, mu), Cz.T - np.outer(Rz1, mu))))
            V, U = Vtilde[:self.y_dim], Vtilde[self.y_dim:]
        if upd == (True, False, True):
            Rytilde = np.vstack((np.hstack((Rz, Rz1[:, None])), np.hstack((Rz1, N))))
            Vtilde = solve(Rytilde, np.vstack((Cz.T - np.outer(Ryz, V), F[:, None] - np.outer(Ry1, V))))
            U, mu = Vtilde[:-1], Vtilde[-1]
        if upd == (False, True, False):
            Rytilde = np.vstack((np.hstack((Ry, Ry1[:, None])), np.hstack((Ry1, N))))
            Vtilde = solve(Rytilde, np.vstack((Cy.T - Ryz @ U, F[:, None] - Rz1 @ U)))
            V, U = Vtilde[:-1], Vtilde[-1]
        Rytilde = np.vstack((np.hstack((Ry, Ryz, Ry1[:, None])), np.hstack((Ryz.T, Rz, Rz1[:, None])),
                             np.hstack((Ry1, Rz1, N))))
        Cytilde = np.hstack((Cy, Cz, F[:, None]))
        if upd == (True, True, True):
            Vtilde = solve(Rytilde, Cytilde.T)
            V, U, mu = Vtilde[:self.y_dim, :], Vtilde[self.y_dim:-1], Vtilde[-1]
        self.mu, self.V, self.U = mu, V, U
        if self.update_D:
            Vtilde = np.vstack((V, U, mu))
            CVt = Cytilde @ Vtilde
            iD = np.diag((S - CVt - CVt.T + Vtilde.T @ Rytilde @ Vtilde) / N).copy()
            iD[iD < self.floor_iD] = self.floor_iD
            self.D = 1 / iD

    @engine.small_host_algebra
    def MstepMD(self, stats):
        """Minimum-divergence update: the priors of both factors are moved back to N(0, I)
        (plda.py:364-394)."""
        from scipy import linalg as sla

        N, M, F, S, _, y_acc, Ry1, Ry, Cy, Py, Rz1, Rz, Ryz, Cz = stats
        f = lambda a: np.asarray(a, dtype=np.float64)  # noqa: E731
        mu, V, U = f(self.mu), f(self.V), f(self.U)
        mu_y = y_acc / M
        chol_Cov_y = sla.cholesky(Py / M - np.outer(mu_y, mu_y), lower=False)
        R = sla.cholesky(Ry - np.outer(Ry1, Ry1) / N, lower=False)
        H = sla.cho_solve((R, False), (Ryz.T - np.outer(Rz1, Ry1) / N).T).T
        mu_z = (Rz1 - Ry1 @ H.T) / N
        RzyH = Ryz.T @ H.T
        chol_Cov_z = sla.cholesky((Rz - RzyH - RzyH.T + H @ Ry @ H.T) / N - np.outer(mu_z, mu_z), lower=False)
        if self.update_mu:
            self.mu = mu + mu_y @ (V + H.T @ U) + mu_z @ U
        if self.update_V:
            self.V = chol_Cov_y @ (V + H.T @ U)
        if self.update_U:
            self.U = chol_Cov_z @ U

    # ---- sampling, adaptation (plda.py:541-589) --------------------------------------------------
    def compute_aux(self):
        """Sets the reference's auxiliary attributes from the current parameters (plda.py:77-89):
        _DU, _J, _Lz, _mult_iLz, _log_Lz, the within-class precision _W (Woodbury form of
        (diag(1/D) + U^T U)^-1) and the products _VW [x_dim, y_dim], _VWV.  Scoring and training here
        fold the parameters themselves (fingerprinted caches), so nothing reads these back; they are
        for callers that did."""
        from ...utils.math import invert_pdmat

        a = self._aux()
        f = lambda v: np.asarray(v, dtype=np.float64)  # noqa: E731
        self._DU, self._J, self._W = a["DU"], a["J"], a["W"]
        self._Lz = np.eye(a["DU"].shape[0]) + a["DU"] @ f(self.U).T
        self._mult_iLz, _, self._log_Lz = invert_pdmat(self._Lz, right_inv=True, return_logdet=True)
        self._VW, self._VWV = self._WV()

    def log_probx_g_y(self, x, y):
        """log p(x_i | y_i) with the channel factor integrated out (plda.py:407-417): the precision
        is W = (diag(1/D) + U^T U)^-1 and log|W| = sum log D - log|I + U D U^T|."""
        a = self._aux()
        logdet_W = float(np.sum(np.log(np.asarray(self.D, dtype=np.float64)))) - a["log_Lz"]
        logp = self._row_logp(self._residual(x, (y, self.V)), a["W"], logdet_W)
        return engine.to_host(logp, float_cpu())

    def log_probx_g_yz(self, x, y, z):
        """log p(x_i | y_i, z_i) (plda.py:419-428): diagonal precision D around mu + y V + z U."""
        D = np.asarray(self.D, dtype=np.float64)
        logp = self._row_logp(self._residual(x, (y, self.V), (z, self.U)), D, float(np.sum(np.log(D))))
        return engine.to_host(logp, float_cpu())

    def sample(self, num_classes, num_samples_per_class, rng=None, seed=1024):
        """Host draw with numpy's generator, in the reference's order; the residual is divided by D
        itself, as it is there (plda.py:541-562)."""
        if rng is None:
            rng = np.random.RandomState(seed=seed)
        f = lambda a: np.asarray(a, dtype=np.float64)  # noqa: E731
        mu, V, U, D = f(self.mu), f(self.V), f(self.U), f(self.D)
        n, d = num_classes * num_samples_per_class, mu.shape[0]
        z1 = rng.normal(size=(n, d)).astype(float_cpu(), copy=False) / D
        z2 = rng.normal(size=(n, self.z_dim)).astype(float_cpu(), copy=False) @ U
        y = rng.normal(size=(num_classes, self.y_dim)).astype(float_cpu(), copy=False) @ V + mu
        return np.repeat(y, num_samples_per_class, axis=0) + z1 + z2

    def weighted_avg_params(self, mu, V, U, D, w_mu, w_B, w_W):
        """Interpolation towards (mu, V, U, D): between-class covariance V^T V refactored to y_dim
        directions; within-class covariance U^T U + diag(1 / D) refactored to z_dim directions plus
        a floored diagonal (plda.py:564-589)."""
        from scipy import linalg as sla

        f = lambda a: np.asarray(a, dtype=np.float64)  # noqa: E731
        self.weigthed_avg_params(mu, w_mu)
        if w_B > 0:
            w, E = sla.eigh(w_B * (f(V).T @ f(V)) + (1 - w_B) * (f(self.V).T @ f(self.V)))
            self.V = (np.sqrt(w) * E)[:, -self.y_dim:].T
        if w_W > 0:
            Sw0 = f(self.U).T @ f(self.U) + np.diag(1 / f(self.D))
            Sw = w_W * (f(U).T @ f(U) + np.diag(1 / f(D))) + (1 - w_W) * Sw0
            w, E = sla.eigh(Sw)
            self.U = (np.sqrt(w) * E)[:, -self.z_dim:].T
            iD = np.diag(Sw - self.U.T @ self.U).copy()
            iD[iD < self.floor_iD] = self.floor_iD
            self.D = 1 / iD

    def weighted_avg_model(self, plda, w_mu, w_B, w_W):
        self.weighted_avg_params(plda.mu, plda.V, plda.U, plda.D, w_mu, w_B, w_W)

    def get_config(self):
        config = {"z_dim": self.z_dim, "update_V": self.update_V, "update_U": self.update_U,
                  "update_D": self.update_D, "floor_iD": self.floor_iD}
        base_config = super().get_config()
        return dict(list(base_config.items()) + list(config.items()))
