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

    @staticmethod
    def get_pca_dim_for_var_ratio(x, var_r=1, min_dim=2):
        """Number of directions that hold the share `var_r` of the (uncentred) energy of the rows
        of x, at least min_dim (pca.py:44-59).  The squared singular values of x are the eigenvalues
        of its Gram matrix, which comes from the device; with var_r = 1 the answer is the rank of x,
        counted like numpy's matrix_rank (singular values above max(n, d) eps times the largest) -
        on the Gram matrix only values above sqrt(eps) times the largest can be told from zero, so
        the count uses that threshold when it is the larger one."""
        import numpy as np

        from .. import engine

        n, d = x.shape
        C, _ = engine.gram64(x)
        ev = np.flip(np.linalg.eigvalsh(C.cpu().numpy()))
        ev = np.maximum(ev, 0.0)[:min(n, d)]
        if var_r == 1:
            eps = np.finfo(np.float64).eps
            tol = max(max(n, d) * eps, np.sqrt(d * eps)) * np.sqrt(ev[0]) if ev.size else 0.0
            rank = int(np.sum(np.sqrt(ev) > tol))
        else:
            share = np.cumsum(ev)
            share = share / share[-1]
            rank = int(np.where(share > var_r)[0][0])
        return max(min_dim, rank)

    _valid_args = ("update_mu", "update_T", "name", "pca_dim", "pca_var_r")

    @classmethod
    def _add_own_args(cls, parser, p1):
        """pca.py:166-178; the name option has no prefix there."""
        parser.add_argument(p1 + "pca-dim", default=None, type=int, help="output dimension of PCA")
        parser.add_argument(p1 + "pca-var-r", default=None, type=int,
                            help="proportion of variance to keep when choosing the PCA dimension")
        parser.add_argument("--name", dest="name", default="pca")

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
