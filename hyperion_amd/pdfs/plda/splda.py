import numpy as np

from ... import fold
from .plda_base import PLDABase


class SPLDA(PLDABase):
    """Simplified PLDA: x = mu + y V + e, e ~ N(0, W^-1).  V is [y_dim, x_dim], W the within-class
    precision [x_dim, x_dim] (constructor as hyperion/pdfs/plda/splda.py:13-39)."""

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

    def get_config(self):
        config = {"update_W": self.update_W, "update_V": self.update_V, "fullcov_W": self.fullcov_W}
        base_config = super().get_config()
        return dict(list(base_config.items()) + list(config.items()))
