from .. import engine
from ..hyp_defs import float_cpu
from .score_norm import ScoreNorm, dev_mask, dev_matrix, norm_apply, row_meanstd


class ZNorm(ScoreNorm):
    """Z-norm: each enrolment row is normalised by the mean / std of its cohort scores
    (hyperion/score_norm/z_norm.py:11-32).  The caller's arrays are never modified."""

    def predict(self, scores, scores_enr_coh, mask=None):
        S = dev_matrix(scores)
        EC = dev_matrix(scores_enr_coh)
        assert EC.shape[0] == S.shape[0]
        m = dev_mask(mask, EC.shape) if mask is not None else None
        out = norm_apply(S, z=row_meanstd(EC, m, self.std_floor))
        return engine.to_host(out, float_cpu())
