"""Detection metrics: for trial sets that stay on the device (trial_hist.TrialHistogram) and, with the
reference's function names, for explicit score arrays (detection)."""
from .detection import (compute_act_dcf, compute_dcf, compute_eer, compute_min_dcf, compute_prbep,  # noqa: F401
                        compute_roc, compute_rocch, effective_prior, fast_eval_dcf_eer, opt_loglr, pavx,
                        rocch2eer)
from .trial_hist import TrialHistogram, rocch_from_counts, eer_from_rocch, min_dcf_from_rocch  # noqa: F401
