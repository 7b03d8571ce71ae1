"""Score normalisation with the hyperion.score_norm call signatures, executed on the GPU."""
from .score_norm import ScoreNorm
from .t_norm import TNorm
from .z_norm import ZNorm
from .s_norm import SNorm
from .zt_norm import TZNorm, ZTNorm
from .adapt_s_norm import AdaptSNorm

__all__ = ["ScoreNorm", "TNorm", "ZNorm", "SNorm", "ZTNorm", "TZNorm", "AdaptSNorm"]
