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

    def fit(self, x=None, sample_weight=None, mu=None, S=None):
        """Mean and whitening transform from data (or from a given mean and covariance):
        T = eigenvectors of S scaled by 1 / sqrt(eigenvalue), largest first, first row non-negative,
        directions of zero variance dropped (cent_whiten.py:35-62).  N, sum x and x^T x come from
        the device (float64 matrix cores); the d x d eigendecomposition stays on the host."""
        import scipy.linalg as la

        from .. import engine

        if x is not None:
            x = np.asarray(x) if not hasattr(x, "is_cuda") else x
            if x.shape[0] > x.shape[1]:
                C, F = engine.gram64(x, sample_weight)
                N = float(x.shape[0]) if sample_weight is None else float(np.sum(np.asarray(sample_weight)))
                mu = F.cpu().numpy() / N
                S = C.cpu().numpy() / N - np.outer(mu, mu)
            else:  # fewer vectors than dimensions: centre only (cent_whiten.py:43-45)
                mu = engine.gram64(x)[1].cpu().numpy() / x.shape[0]
                S = np.eye(x.shape[1])
        if self.update_mu:
            self.mu = mu
        if self.update_T:
            d, V = la.eigh(S)
            with np.errstate(divide="ignore", invalid="ignore"):
                V = V * np.sqrt(1 / d)
            V = np.fliplr(V)
            V[:, V[0, :] < 0] *= -1
            nonzero = d > 0
            if not np.all(nonzero):
                V = V[:, nonzero[::-1]]
            self.T = V

    def get_config(self):
        config = {"update_mu": self.update_mu, "update_t": self.update_T}
        base_config = super().get_config()
        return dict(list(base_config.items()) + list(config.items()))
