from .cent_whiten import CentWhiten


class LNorm(CentWhiten):
    """Length normalisation: centre/whiten, then sqrt(D) x / (|x| + 1e-10) with D the output
    dimension (hyperion/transforms/lnorm.py:11-17)."""

    _lnorm = True
