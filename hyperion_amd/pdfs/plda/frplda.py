import numpy as np

from ... import fold
from .plda_base import PLDABase


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

    def get_config(self):
        config = {"update_W": self.update_W, "update_B": self.update_B, "fullcov_W": self.fullcov_W}
        base_config = super().get_config()
        return dict(list(base_config.items()) + list(config.items()))
