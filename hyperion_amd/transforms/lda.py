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

    def fit(self, x, y, mu=None, Sb=None, Sw=None):
        """Projection onto the leading generalised eigenvectors of (Sb, Sw), first row non-negative
        (lda.py:36-68); the covariances come from SbSw (device) unless given."""
        import numpy as np
        import scipy.linalg as la

        from .sb_sw import SbSw

        if mu is None or Sb is None or Sw is None:
            sbsw = SbSw()
            sbsw.fit(x, y)
            mu, Sb, Sw = sbsw.mu, sbsw.Sb, sbsw.Sw
        if self.update_mu:
            self.mu = mu
        if not self.update_T:
            return
        assert Sb.shape == Sw.shape
        try:
            d, V = la.eigh(Sb, Sw)
        except la.LinAlgError:  # Sw not positive definite: a ridge of 1 % of its largest variance
            alpha = 1e-2 * np.max(np.diag(Sw))
            d, V = la.eigh(Sb, alpha * np.eye(Sw.shape[0]) + Sw)
        V = np.fliplr(V)
        V[:, V[0, :] < 0] *= -1
        if self.lda_dim is not None:
            assert self.lda_dim <= V.shape[1]
            V = V[:, :self.lda_dim]
        self.T = V

    def get_config(self):
        config = {"lda_dim": self.lda_dim, "update_mu": self.update_mu, "update_t": self.update_T}
        base_config = super().get_config()
        return dict(list(base_config.items()) + list(config.items()))
