from .. import engine
from ..hyp_defs import float_cpu
from .score_norm import (ScoreNorm, dev_mask, dev_matrix, norm_apply, row_meanstd, transpose,
                         transpose_mask)
from .t_norm import TNorm
from .z_norm import ZNorm


class SNorm(ScoreNorm):
    """Symmetric normalisation (z + t) / sqrt(2) (hyperion/score_norm/s_norm.py:14-34)."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.t_norm = TNorm(**kwargs)
        self.z_norm = ZNorm(**kwargs)

    def predict(self, scores, scores_coh_test, scores_enr_coh, mask_coh_test=None, mask_enr_coh=None):
        S = dev_matrix(scores)
        EC = dev_matrix(scores_enr_coh)
        TC = transpose(dev_matrix(scores_coh_test))
        assert EC.shape[0] == S.shape[0] and TC.shape[0] == S.shape[1]
        mz = dev_mask(mask_enr_coh, EC.shape) if mask_enr_coh is not None else None
        mt = None
        if mask_coh_test is not None:
            mt = transpose_mask(dev_mask(mask_coh_test, (TC.shape[1], TC.shape[0])))
        out = norm_apply(S, z=row_meanstd(EC, mz, self.z_norm.std_floor),
                         t=row_meanstd(TC, mt, self.t_norm.std_floor))
        return engine.to_host(out, float_cpu())
