"""hyperion_amd: MI355X (gfx950) implementation of Hyperion's PLDA back-end scoring path.

Same call signatures as hyperion.transforms / hyperion.pdfs.plda / hyperion.score_norm (numpy in,
numpy out); the arithmetic on embeddings and scores runs in hand-written HIP kernels behind the
C ABI of include/hypmi.h.  There is no CPU fallback."""
__version__ = "0.1.0"

from . import hyp_defs  # noqa: F401
from .hyp_defs import float_cpu, set_float_cpu  # noqa: F401
