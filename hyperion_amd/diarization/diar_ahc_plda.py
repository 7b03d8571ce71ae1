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
    small matrices.  The unsupervised score calibration of the reference (a two-component GMM on
    the scores) is not available: do_unsup_cal=True raises."""

    def __init__(self, plda_model, preproc=None, threshold=0, pca_var_r=1, do_unsup_cal=False, use_bic=False):
        if do_unsup_cal:
            raise NotImplementedError("unsupervised GMM calibration of the scores is not part of this package")
        self.plda_model = plda_model
        self.preproc = preproc
        self.threshold = threshold
        self.pca_var_r = pca_var_r
        self.do_unsup_cal = False
        self.use_bic = False
        self._ahc = AHC()

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
        self._ahc.fit(self.scores(x))
        return self._ahc.get_flat_clusters(self.threshold)

    @staticmethod
    def filter_args(**kwargs):
        valid = ("threshold", "pca_var_r", "do_unsup_cal", "use_bic")
        return {k: kwargs[k] for k in valid if k in kwargs}
