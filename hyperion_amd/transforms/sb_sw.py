"""Between-class and within-class covariance of labelled vectors (hyperion/transforms/sb_sw.py)."""
import numpy as np

from .. import engine
from ..hyp_model import HypModel


class SbSw(HypModel):
    """mu: mean of the class means; Sb: covariance of the class means; Sw: mean over classes of the
    per-class covariance (sb_sw.py:15-50).  The per-class loop of the reference becomes two Gram
    matrices on the device: sum_i (1 / N_i) sum_{j in i} x_j x_j^T, with row weights 1 / N_class,
    and M^T M over the class means M."""

    def __init__(self, Sb=None, Sw=None, mu=None, num_classes=0, **kwargs):
        super().__init__(**kwargs)
        self.Sb = None
        self.Sw = None
        self.mu = None
        self.num_classes = num_classes

    def fit(self, x, class_ids, sample_weight=None, class_weights=None, normalize=True):
        """Adds the classes of (x, class_ids) to the accumulated sums; normalize() turns sums into
        the statistics (sb_sw.py:25-50; sample_weight and class_weights are accepted and unused
        there as well)."""
        engine.require_gpu()
        X = engine.upload64(x)
        n, dim = X.shape
        if self.Sb is None:
            self.Sb = np.zeros((dim, dim))
            self.Sw = np.zeros((dim, dim))
            self.mu = np.zeros((dim,))
            self.num_classes = 0
        u_ids, inv = np.unique(np.asarray(class_ids), return_inverse=True)
        inv = inv.reshape(-1)
        counts = np.bincount(inv, minlength=len(u_ids))
        row_w = 1.0 / counts[inv]  # every class weighs the same, whatever its size
        order = np.argsort(inv, kind="stable").astype(np.int32)
        offsets = np.zeros(len(u_ids) + 1, dtype=np.int64)
        np.cumsum(counts, out=offsets[1:])
        M = engine.pool64(X, order, offsets, row_w)  # class means: rows summed in list order
        within, _ = engine.gram64(X, row_w)
        between, mean_sum = engine.gram64(M)
        self.num_classes += len(u_ids)
        self.mu += mean_sum.cpu().numpy()
        self.Sb += between.cpu().numpy()
        self.Sw += (within - between).cpu().numpy()
        if normalize:
            self.normalize()

    def normalize(self):
        self.mu /= self.num_classes
        self.Sb = self.Sb / self.num_classes - np.outer(self.mu, self.mu)
        self.Sw /= self.num_classes

    @classmethod
    def accum_stats(cls, stats):
        """Sum of un-normalised statistics objects (sb_sw.py:52-66)."""
        out = cls()
        out.mu = sum(s.mu for s in stats)
        out.Sb = sum(s.Sb for s in stats)
        out.Sw = sum(s.Sw for s in stats)
        out.num_classes = sum(s.num_classes for s in stats)
        return out

    def get_config(self):
        config = {"num_classes": self.num_classes}
        base_config = super().get_config()
        return dict(list(base_config.items()) + list(config.items()))
