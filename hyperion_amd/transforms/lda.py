from .base import AffineTransform


class LDA(AffineTransform):
    """Linear discriminant projection x -> (x - mu) T (hyperion/transforms/lda.py:15-34)."""

    def __init__(self, mu=None, T=None, lda_dim=None, update_mu=True, update_T=True, **kwargs):
        super().__init__(mu=mu, T=T, **kwargs)
        self.lda_dim = lda_dim if T is None else T.shape[1]
        self.update_mu = update_mu
        self.update_T = update_T

    def _step(self):
        if self.T is None:
            raise ValueError("LDA.predict needs a projection matrix T")
        return super()._step()

    def get_config(self):
        config = {"lda_dim": self.lda_dim, "update_mu": self.update_mu, "update_t": self.update_T}
        base_config = super().get_config()
        return dict(list(base_config.items()) + list(config.items()))
