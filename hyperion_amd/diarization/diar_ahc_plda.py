import numpy as np

from .. import engine
from ..clustering import AHC
from ..transforms import PCA, LNorm


class DiarAHCPLDA:
    """Diarization of one recording by clustering its segments on PLDA scores
    (hyperion/diarization/diar_ahc_plda.py:21-168): pre-processing, optionally a PCA fitted on the
    recording with the PLDA model projected onto it, the [segments x segments] self-score matrix,
    average-linkage AHC cut at `threshold`.  The vectors, the PCA moments and the score matrix are
    device work; the eigendecomposition, the model projection and the linkage are host work on
    small matrices.  With do_unsup_cal a two-Gaussian mixture with one shared variance is fitted to
    the scores above the diagonal (on the device); with use_bic a recording whose scores one
    Gaussian explains as well (negative BIC) comes back as a single cluster.  As in the reference,
    the calibrated scores themselves do not reach the clustering (there they replace the scores
    only when a histogram plot is asked for, diar_ahc_plda.py:151-154; plots are not made here)."""

    def __init__(self, plda_model, preproc=None, threshold=0, pca_var_r=1, do_unsup_cal=False, use_bic=False):
        self.plda_model = plda_model
        self.preproc = preproc
        self.threshold = threshold
        self.pca_var_r = pca_var_r
        self.do_unsup_cal = do_unsup_cal
        self.use_bic = use_bic and do_unsup_cal
        self._ahc = AHC()

    @staticmethod
    def _unsup_gmm_calibration(scores, var_floor=1e-3, epochs=20):
        """Unsupervised calibration of a self-score matrix (diar_ahc_plda.py:100-120): the scores
        above the diagonal are modelled by two Gaussians with one shared variance, started from the
        single Gaussian of all scores split at mean +- one deviation and trained by EM
        (hyperion/pdfs/mixtures/gmm_tied_diag_cov.py:53-121: component variances floored at
        var_floor times their mean, then pooled).  The log-likelihood ratio of the two components
        is linear in the score: returns (scale * scores + bias, bic, (pi, mu, Lambda)), bic being
        the mean log-likelihood gain of two Gaussians over one minus the penalty of the four extra
        parameters.  Device float64 throughout; `scores` may be a host array or a device tensor
        (the result is of the same kind)."""
        import math

        torch = engine.require_gpu()
        is_dev = isinstance(scores, torch.Tensor)
        S = scores.to(device="cuda", dtype=torch.float64) if is_dev else torch.from_numpy(
            np.ascontiguousarray(scores, dtype=np.float64)).cuda()
        n_all = S.shape[0]
        iu = torch.triu_indices(n_all, n_all, offset=1, device=S.device)
        x = S[iu[0], iu[1]]
        n = x.numel()
        assert n >= 1, "calibration needs at least two segments"
        m1 = x.mean()
        lam1 = 1.0 / ((x - m1) ** 2).mean()
        half_log_2pi = 0.5 * math.log(2 * math.pi)

        def comp_llk(pi, mu, lam):
            return torch.log(pi) + 0.5 * torch.log(lam) - half_log_2pi - 0.5 * lam * (x[:, None] - mu) ** 2

        def log_prob(pi, mu, lam):
            r = comp_llk(pi, mu, lam)
            mx = r.max(dim=-1, keepdim=True).values
            return torch.log(torch.exp(r - mx).sum(dim=-1) + 1e-20) + mx[:, 0]

        sd = 1.0 / torch.sqrt(lam1)
        pi = torch.full((2,), 0.5, dtype=torch.float64, device=S.device)
        mu = torch.stack((m1 + sd, m1 - sd))
        lam = lam1
        for _ in range(epochs):
            z = torch.softmax(comp_llk(pi, mu, lam), dim=-1)
            N = z.sum(dim=0)
            mu = (z.T @ x) / N
            var = (z.T @ (x * x)) / N - mu * mu
            var = torch.maximum(var, var_floor * var.mean())
            lam = 1.0 / ((N * var).sum() / N.sum())
            pi = N / N.sum()
        one = torch.ones((1,), dtype=torch.float64, device=S.device)
        bic = float((log_prob(pi, mu, lam) - log_prob(one, m1[None], lam1)).mean().item()) \
            - 4 * math.log(n) / 2 / n
        scale = (mu[0] - mu[1]) * lam
        bias = (mu[1] ** 2 - mu[0] ** 2) * lam / 2 + torch.log(pi[0]) - torch.log(pi[1])
        cal = scale * S + bias
        gmm = (pi.cpu().numpy(), mu.cpu().numpy(), float(lam.item()))
        return (cal.to(scores.dtype) if is_dev else cal.cpu().numpy()), bic, gmm

    def scores(self, x):
        """The self-score matrix the clustering works on, as a device tensor [n, n]."""
        x = engine.upload(x)
        if self.preproc is not None:
            x = self.preproc.predict(x)
        plda_model = self.plda_model
        if self.pca_var_r < 1:
            pca = PCA(pca_var_r=self.pca_var_r, whiten=True)
            pca.fit(x)
            x = LNorm().predict(pca.predict(x))
            plda_model = plda_model.project(pca.T, pca.mu)
        return plda_model.llr_1vs1(x, x, refine=True)

    def cluster(self, x, hist_file=None):
        """x [segments, dim] -> cluster id per segment (host int array).  hist_file (a plot in the
        reference) is ignored."""
        scores = self.scores(x)
        if self.do_unsup_cal:
            _, bic, _ = self._unsup_gmm_calibration(scores)
            if self.use_bic and bic < 0:  # one Gaussian is enough: target trials only
                return np.zeros(scores.shape[0], dtype=int)
        self._ahc.fit(scores)
        return self._ahc.get_flat_clusters(self.threshold)

    @staticmethod
    def filter_args(**kwargs):
        valid = ("threshold", "pca_var_r", "do_unsup_cal", "use_bic")
        return {k: kwargs[k] for k in valid if k in kwargs}

    @staticmethod
    def add_class_args(parser, prefix=None):
        """hyperion/diarization/diar_ahc_plda.py:186-203."""
        p1 = "--" if prefix is None else "--" + prefix + "."
        parser.add_argument(p1 + "threshold", default=0, type=float)
        parser.add_argument(p1 + "pca-var-r", default=1, type=float)
        parser.add_argument(p1 + "do-unsup-cal", default=False, action="store_true")
        parser.add_argument(p1 + "use-bic", default=False, action="store_true")

    add_argparse_args = add_class_args
