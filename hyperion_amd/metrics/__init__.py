"""Detection metrics for trial sets that stay on the device (see trial_hist.TrialHistogram)."""
from .trial_hist import TrialHistogram, rocch_from_counts, eer_from_rocch, min_dcf_from_rocch  # noqa: F401
