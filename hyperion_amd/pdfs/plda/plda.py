import numpy as np

from ... import fold
from ...hyp_defs import float_cpu
from .plda_base import PLDABase


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

    # ---- sampling, adaptation (plda.py:541-589) --------------------------------------------------
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
