from .base import AffineTransform


class PCA(AffineTransform):
    """Principal component projection x -> (x - mu) T (hyperion/transforms/pca.py:14-42)."""

    def __init__(self, mu=None, T=None, update_mu=True, update_T=True, pca_dim=None,
                 pca_var_r=None, pca_min_dim=2, whiten=False, **kwargs):
        super().__init__(mu=mu, T=T, **kwargs)
        self.update_mu = update_mu
        self.update_T = update_T
        self.pca_dim = pca_dim
        self.pca_var_r = pca_var_r
        self.pca_min_dim = pca_min_dim
        self.whiten = whiten

    def _step(self):
        if self.T is None:
            raise ValueError("PCA.predict needs a projection matrix T")
        return super()._step()

    def get_config(self):
        config = {"update_mu": self.update_mu, "update_t": self.update_T, "pca_dim": self.pca_dim,
                  "pca_var_r": self.pca_var_r}
        base_config = super().get_config()
        return dict(list(base_config.items()) + list(config.items()))
