"""Base class and device helpers shared by the score normalisers."""
import numpy as np

from .. import _lib, engine
from ..hyp_model import HypModel


class ScoreNorm(HypModel):
    """Base class (hyperion/score_norm/score_norm.py:11-18).  Keyword arguments are forwarded by
    name; the reference forwards them with a single star, which turns the keys into positional
    arguments."""

    def __init__(self, std_floor=1e-5, **kwargs):
        super().__init__(**kwargs)
        self.std_floor = std_floor


def dev_matrix(x, copy=False):
    """Score matrix -> device fp32 [r, c] with a row pitch that keeps rows 16-byte aligned."""
    t = engine.upload(x)
    if copy and isinstance(x, type(t)):
        t = t.clone()
    return t


def dev_mask(mask, shape):
    """Boolean mask -> device uint8 [r, c] (1 = keep)."""
    torch = engine.require_gpu()
    if isinstance(mask, torch.Tensor):
        m = mask.to(device="cuda", dtype=torch.uint8).contiguous()
    else:
        m = torch.from_numpy(np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)).cuda()
    if tuple(m.shape) != tuple(shape):
        raise AssertionError("mask shape %s does not match scores %s" % (tuple(m.shape), tuple(shape)))
    return m


def zero_masked(X, mask_u8):
    lib = _lib.load()
    _lib.check(lib.hypmi_mask_zero(engine._ptr(X), X.shape[0], X.shape[1], X.stride(0),
                                   engine._ptr(mask_u8), mask_u8.stride(0), engine.stream_ptr()))


def transpose(X):
    """Device [r, c] fp32 -> device [c, r] through the library's tiled transpose."""
    torch = engine.require_gpu()
    lib = _lib.load()
    r, c = X.shape
    ld = engine.round_up(max(r, 1), 4)
    out = torch.empty((c, ld), dtype=torch.float32, device=X.device)
    _lib.check(lib.hypmi_transpose(engine._ptr(X), r, c, X.stride(0), engine._ptr(out), ld,
                                   engine.stream_ptr()))
    return out[:, :r]


def transpose_mask(mask_u8):
    """uint8 [r, c] -> uint8 [c, r] (through the fp32 transpose kernel)."""
    torch = engine.require_gpu()
    return transpose(mask_u8.to(torch.float32)).to(torch.uint8).contiguous()


def row_meanstd(X, mask_u8, std_floor):
    """Per-row mean and 1 / max(std, floor) over all columns: two device vectors."""
    torch = engine.require_gpu()
    lib = _lib.load()
    n, nc = X.shape
    mean = torch.empty(n, dtype=torch.float32, device=X.device)
    istd = torch.empty(n, dtype=torch.float32, device=X.device)
    _lib.check(lib.hypmi_row_meanstd(engine._ptr(X), n, nc, X.stride(0),
                                     engine._ptr(mask_u8) if mask_u8 is not None else None,
                                     mask_u8.stride(0) if mask_u8 is not None else 0,
                                     float(std_floor), engine._ptr(mean), engine._ptr(istd),
                                     engine.stream_ptr()))
    return mean, istd


def norm_apply(S, z=None, t=None):
    torch = engine.require_gpu()
    lib = _lib.load()
    ne, nt = S.shape
    ld = engine.round_up(max(nt, 1), 32)
    out = torch.empty((ne, ld), dtype=torch.float32, device=S.device)
    p = lambda v: engine._ptr(v) if v is not None else None  # noqa: E731
    mz, iz = z if z is not None else (None, None)
    mt, it = t if t is not None else (None, None)
    _lib.check(lib.hypmi_norm_apply(engine._ptr(S), S.stride(0), ne, nt, p(mz), p(iz), p(mt), p(it),
                                    engine._ptr(out), ld, engine.stream_ptr()))
    return out[:, :nt]
