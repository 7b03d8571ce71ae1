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

    def fit(self, x=None, sample_weight=None, mu=None, S=None):
        """Mean and leading eigenvectors of the covariance, largest first, first row non-negative;
        pca_var_r picks the dimension that explains that share of the variance, whiten scales the
        directions to unit variance and drops those of zero variance (pca.py:61-106;
        sample_weight is unused there too).  Moments from the device, eigendecomposition on the
        host."""
        import numpy as np
        import scipy.linalg as la

        from .. import engine

        if x is not None:
            C, F = engine.gram64(x)
            n = x.shape[0]
            mu = F.cpu().numpy() / n
            S = C.cpu().numpy() / n - np.outer(mu, mu)
        if self.update_mu:
            self.mu = mu
        if not self.update_T:
            return
        d, V = la.eigh(S)
        d, V = np.flip(d), np.fliplr(V)
        V[:, V[0, :] < 0] *= -1
        if self.pca_var_r is not None:
            share = np.cumsum(d) / np.sum(d)
            self.pca_dim = max(int(np.where(share > self.pca_var_r)[0][0]), self.pca_min_dim)
        if self.whiten:
            dead = d <= 0
            if np.any(dead):
                keep = int(np.where(dead)[0][0])
                V = V[:, :keep] / np.sqrt(d[:keep])
                self.pca_dim = keep if self.pca_dim is None else min(keep, self.pca_dim)
            else:
                V = V / np.sqrt(d)
        if self.pca_dim is not None:
            assert self.pca_dim <= V.shape[1]
            V = V[:, :self.pca_dim]
        self.T = V

    @classmethod
    def load_params(cls, f, config):
        """hyperion/transforms/pca.py:120-129."""
        params = cls._load_params_to_dict(f, config["name"], ["mu", "T"])
        return cls(mu=params["mu"], T=params["T"], pca_dim=config.get("pca_dim"), name=config["name"])

    def get_config(self):
        config = {"update_mu": self.update_mu, "update_t": self.update_T, "pca_dim": self.pca_dim,
                  "pca_var_r": self.pca_var_r}
        base_config = super().get_config()
        return dict(list(base_config.items()) + list(config.items()))
