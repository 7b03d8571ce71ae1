from .plda_base import PLDABase
from .splda import SPLDA
from .frplda import FRPLDA
from .plda import PLDA

__all__ = ["PLDABase", "SPLDA", "FRPLDA", "PLDA"]
