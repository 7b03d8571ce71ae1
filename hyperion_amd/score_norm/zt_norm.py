from .. import engine
from ..hyp_defs import float_cpu
from .score_norm import (ScoreNorm, dev_mask, dev_matrix, norm_apply, row_meanstd, transpose,
                         transpose_mask)
from .t_norm import TNorm
from .z_norm import ZNorm


class _TwoStageNorm(ScoreNorm):
    """ZT- and TZ-norm (hyperion/score_norm/zt_norm.py:14-45, tz_norm.py:14-45).  In the reference
    neither class can be constructed (their __init__ name SNorm, which the modules do not import);
    what their predict bodies state is implemented: two chained normalisations with the second
    one's cohort scores normalised the same way as the trial scores.  All stages stay on the
    device; the caller's arrays are never modified."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.t_norm = TNorm(**kwargs)
        self.z_norm = ZNorm(**kwargs)

    def _z(self, S, EC, mask):
        """Rows of S normalised by the statistics of the matching rows of EC."""
        assert EC.shape[0] == S.shape[0]
        m = dev_mask(mask, EC.shape) if mask is not None else None
        return norm_apply(S, z=row_meanstd(EC, m, self.z_norm.std_floor))

    def _t(self, S, CT, mask):
        """Columns of S normalised by the statistics of the matching columns of CT."""
        TC = transpose(CT)
        assert TC.shape[0] == S.shape[1]
        m = transpose_mask(dev_mask(mask, (TC.shape[1], TC.shape[0]))) if mask is not None else None
        return norm_apply(S, t=row_meanstd(TC, m, self.t_norm.std_floor))


class ZTNorm(_TwoStageNorm):
    """Z-norm of the scores and of the cohort-vs-test scores (by the cohort-vs-cohort scores),
    then T-norm of the first by the second.  The reference's body hands mask_enr_coh to the
    Z-norm of the cohort-vs-test scores (zt_norm.py:38-40), whose shape only fits when there are as
    many enrolment models as cohort members; mask_coh_coh, the mask of that stage's cohort scores,
    is used here."""

    def predict(self, scores, scores_coh_test, scores_enr_coh, scores_coh_coh, mask_coh_test=None,
                mask_enr_coh=None, mask_coh_coh=None):
        S, CT = dev_matrix(scores), dev_matrix(scores_coh_test)
        s_z = self._z(S, dev_matrix(scores_enr_coh), mask_enr_coh)
        ct_z = self._z(CT, dev_matrix(scores_coh_coh), mask_coh_coh)
        return engine.to_host(self._t(s_z, ct_z, mask_coh_test), float_cpu())


class TZNorm(_TwoStageNorm):
    """T-norm of the scores and of the enrol-vs-cohort scores (by the cohort-vs-cohort scores),
    then Z-norm of the first by the second (tz_norm.py:34-43)."""

    def predict(self, scores, scores_coh_test, scores_enr_coh, scores_coh_coh, mask_coh_test=None,
                mask_enr_coh=None, mask_coh_coh=None):
        S, EC = dev_matrix(scores), dev_matrix(scores_enr_coh)
        s_t = self._t(S, dev_matrix(scores_coh_test), mask_coh_test)
        ec_t = self._t(EC, dev_matrix(scores_coh_coh), mask_coh_coh)
        return engine.to_host(self._z(s_t, ec_t, mask_enr_coh), float_cpu())
