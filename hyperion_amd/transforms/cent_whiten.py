import numpy as np

from .base import AffineTransform


class CentWhiten(AffineTransform):
    """Centering and whitening, x -> (x - mu) T (hyperion/transforms/cent_whiten.py:15-33).
    T may be a matrix, a 1-D diagonal (the reference's 1-D branch raises NameError; the intended
    element-wise product is implemented) or None."""

    def __init__(self, mu=None, T=None, update_mu=True, update_T=True, **kwargs):
        super().__init__(mu=mu, T=T, **kwargs)
        self.update_mu = update_mu
        self.update_T = update_T

    def get_config(self):
        config = {"update_mu": self.update_mu, "update_t": self.update_T}
        base_config = super().get_config()
        return dict(list(base_config.items()) + list(config.items()))
