"""Embedding pre-processing with the hyperion.transforms call signatures, executed on the GPU."""
from .cent_whiten import CentWhiten
from .lnorm import LNorm
from .lda import LDA
from .pca import PCA
from .sb_sw import SbSw
from .transform_list import TransformList

__all__ = ["CentWhiten", "LNorm", "LDA", "PCA", "SbSw", "TransformList"]
