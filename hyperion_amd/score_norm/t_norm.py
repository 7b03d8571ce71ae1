from .. import engine
from ..hyp_defs import float_cpu
from .score_norm import (ScoreNorm, dev_mask, dev_matrix, norm_apply, row_meanstd, transpose,
                         transpose_mask)


class TNorm(ScoreNorm):
    """T-norm: each test column is normalised by the mean / std of the cohort-vs-test scores
    (hyperion/score_norm/t_norm.py:12-31).  The caller's arrays are never modified."""

    def predict(self, scores, scores_coh_test, mask=None):
        S = dev_matrix(scores)
        TC = transpose(dev_matrix(scores_coh_test))  # rows over the cohort
        assert TC.shape[0] == S.shape[1]
        m = transpose_mask(dev_mask(mask, (TC.shape[1], TC.shape[0]))) if mask is not None else None
        out = norm_apply(S, t=row_meanstd(TC, m, self.std_floor))
        return engine.to_host(out, float_cpu())
