"""Device engine: thin Python over the hypmi C ABI.  torch provides device memory and streams only;
all arithmetic on embeddings and scores happens in the HIP kernels of libhypmi.so."""
import ctypes
import threading

import numpy as np

from . import _lib
from ._lib import SIDE_ENROL, SIDE_TEST  # noqa: F401


def _torch():
    import torch

    return torch


def require_gpu():
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError(
            "hyperion_amd needs a HIP device (built for gfx950 / MI355X); there is no CPU fallback")
    return torch


def stream_ptr():
    return ctypes.c_void_p(_torch().cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def round_up(v, m):
    return (v + m - 1) // m * m


def upload(x, dtype=None):
    """numpy [n, d] (any float) -> device fp32 [n, ld] with ld = d rounded up to 4, pad zero.
    A torch tensor already on the device is passed through after the same layout checks."""
    torch = require_gpu()
    if isinstance(x, torch.Tensor):
        t = x
        if t.device.type != "cuda":
            t = t.cuda()
        if t.dtype != torch.float32:
            t = t.float()
        if t.dim() != 2:
            raise ValueError("expected a 2-D array of embeddings")
        if t.stride(1) != 1 or t.stride(0) % 4 != 0 or t.data_ptr() % 16 != 0 or \
                t.stride(0) < round_up(t.shape[1], 4):
            n, d = t.shape
            buf = torch.zeros((n, round_up(d, 4)), dtype=torch.float32, device=t.device)
            buf[:, :d] = t
            t = buf[:, :d]
        return t
    x = np.asarray(x)
    if x.ndim != 2:
        raise ValueError("expected a 2-D array of embeddings, got shape %s" % (x.shape,))
    n, d = x.shape
    ld = round_up(d, 4)
    if (ld == d and x.dtype != np.float32 and x.size >= (1 << 20) and x.flags.c_contiguous
            and x.dtype.isnative and x.dtype.kind == "f" and x.dtype.itemsize in (2, 8)):
        # a wide host type: narrowed by torch's threaded copy into a page-locked buffer and sent from
        # there (numpy's serial astype plus a pageable copy take four times as long)
        return _upload_staged(x, torch.float32)
    host = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    if ld == d:
        return host.cuda()
    buf = torch.zeros((n, ld), dtype=torch.float32, device="cuda")
    buf[:, :d] = host.cuda()
    return buf[:, :d]


def upload64(x):
    """numpy / device array -> contiguous device float64 [n, d] (input of the refinement path)."""
    torch = require_gpu()
    if isinstance(x, torch.Tensor):
        return x.to(device="cuda", dtype=torch.float64).contiguous()
    x = np.asarray(x)
    if x.ndim != 2:
        raise ValueError("expected a 2-D array of embeddings, got shape %s" % (x.shape,))
    x = np.ascontiguousarray(x, dtype=np.float64)
    if x.size < (1 << 20):
        return torch.from_numpy(x).cuda()
    return _upload_staged(x, torch.float64)


# Page-locked staging: at most this many bytes per buffer (two buffers per direction and thread), whatever
# the size of the array that passes through; larger arrays go in row slices.
STAGE_BYTES = 256 << 20


def _upload_staged(x, dtype):
    """C-contiguous host array [n, d] -> device tensor of `dtype`, through two page-locked buffers of at
    most STAGE_BYTES each in row slices: the conversion of slice i + 1 (torch's threaded copy) runs
    while slice i is on the link; an event per buffer says when it may be filled again."""
    torch = _torch()
    n, d = x.shape
    out = torch.empty((n, d), dtype=dtype, device="cuda")
    item = out.element_size()
    rows = max(1, min(n, STAGE_BYTES // max(1, d * item)))
    src = _as_tensor(x)
    stream = torch.cuda.current_stream()
    bufs = [_pinned("up%d_%d" % (item, i), rows * d, dtype) for i in range(2 if rows < n else 1)]
    busy = [None] * len(bufs)
    for i, r0 in enumerate(range(0, n, rows)):
        r1 = min(n, r0 + rows)
        b = i % len(bufs)
        if busy[b] is not None:
            busy[b].synchronize()
        stage = bufs[b][:(r1 - r0) * d].view(r1 - r0, d)
        stage.copy_(src[r0:r1])
        out[r0:r1].copy_(stage, non_blocking=True)
        busy[b] = torch.cuda.Event()
        busy[b].record(stream)
    for e in busy:
        if e is not None:
            e.synchronize()  # the buffers are reused by the next call
    return out


def _as_tensor(x):
    """torch view of a C-contiguous host array that is only read.  torch warns about writes through a
    view of a read-only array; such an array is therefore aliased through its address (no copy, no
    process-wide warning filter, nothing shared between threads).  The caller keeps `x` alive."""
    if not x.flags.writeable and x.nbytes:
        assert x.flags.c_contiguous
        raw = (ctypes.c_byte * x.nbytes).from_address(x.ctypes.data)
        x = np.frombuffer(raw, dtype=x.dtype).reshape(x.shape)
    return _torch().from_numpy(x)


def _pinned(key, numel, dtype):
    """Cached page-locked host buffer of at least `numel` elements (one set per calling thread, never
    more than STAGE_BYTES each; drop_staging() releases them)."""
    torch = _torch()
    staging = _staging.__dict__
    buf = staging.get(key)
    if buf is None or buf.numel() < numel or buf.dtype != dtype:
        buf = torch.empty(max(numel, 1 << 20), dtype=dtype, pin_memory=True)
        staging[key] = buf
    return buf


def drop_staging():
    """Releases the calling thread's page-locked staging buffers (they are made again on demand)."""
    _staging.__dict__.clear()


_blas_pools = None


def _blas_controller():
    """threadpoolctl controller over the BLAS libraries of numpy and scipy, looked up once (the
    look-up walks the loaded libraries: 1.3 ms, against 0.02 ms for a limit on a known controller)."""
    global _blas_pools
    if _blas_pools is None:
        try:
            import scipy.linalg  # noqa: F401  (its BLAS must be loaded before the look-up)
            from threadpoolctl import ThreadpoolController

            _blas_pools = ThreadpoolController()
        except ImportError:
            _blas_pools = False
    return _blas_pools


def small_host_algebra(fn):
    """Decorator for host code whose LAPACK / BLAS calls work on y_dim- and x_dim-sized matrices:
    they run on one thread.  Waking a 16-thread pool costs ~10 ms per call on a many-core host, a
    hundred times the call itself (measured: 131 ms -> 5 ms per training epoch)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        pools = _blas_controller()
        if not pools:
            return fn(*args, **kwargs)
        with pools.limit(limits=1, user_api="blas"):
            return fn(*args, **kwargs)

    return wrapped


def _rows64(x):
    """Device float64 rows with unit column stride (a column slice of a wider buffer is taken as it
    is, anything else goes through upload64)."""
    torch = require_gpu()
    if (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float64 and x.dim() == 2
            and (x.shape[1] == 1 or x.stride(1) == 1) and (x.shape[0] <= 1 or x.stride(0) >= x.shape[1])):
        return x
    return upload64(x)


def _vec64(w, n):
    torch = require_gpu()
    W = w if isinstance(w, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(w, dtype=np.float64))
    W = W.to(device="cuda", dtype=torch.float64).contiguous()
    assert W.shape == (n,)
    return W


def gram64(x, w=None):
    """(sum_s w_s x_s x_s^T, sum_s w_s x_s) as device float64 tensors [d, d] and [d]; x numpy or
    device [n, d], w optional row weights [n]."""
    torch = require_gpu()
    X = _rows64(x)
    n, d = X.shape
    W = None if w is None else _vec64(w, n)
    C = torch.empty((d, d), dtype=torch.float64, device="cuda")
    F = torch.empty((d,), dtype=torch.float64, device="cuda")
    lib = _lib.load()
    nbytes = lib.hypmi_gram_f64_workspace_bytes(n, d)
    ws = torch.empty(max(nbytes // 8, 1), dtype=torch.float64, device="cuda")
    # partial blocks are added in a fixed order: the statistics (and what is trained from them) are the
    # same bits from run to run
    _lib.check(lib.hypmi_gram_f64_ordered(_ptr(X), n, X.stride(0) if n > 1 else d, d,
                                          ctypes.c_void_p(0) if W is None else _ptr(W), _ptr(C), d, _ptr(F),
                                          _ptr(ws), nbytes, stream_ptr()))
    # with row weights the two triangles of a diagonal block round x_f (w x_g) and x_g (w x_f) differently
    return 0.5 * (C + C.T), F


def gemm64(a, b, bias=None, out=None):
    """a [m, k] @ b[n, k]^T (+ bias [n]) in float64 on the device; `out` may be a column slice of a
    wider float64 buffer."""
    torch = require_gpu()
    A, B = _rows64(a), _rows64(b)
    m, k = A.shape
    n = B.shape[0]
    assert B.shape[1] == k
    if out is None:
        out = torch.empty((m, n), dtype=torch.float64, device="cuda")
    assert out.shape == (m, n) and out.dtype == torch.float64 and (n == 1 or out.stride(1) == 1)
    bv = None if bias is None else _vec64(bias, n)
    _lib.check(_lib.load().hypmi_gemm_nt_f64(_ptr(A), m, A.stride(0) if m > 1 else k, _ptr(B), n,
                                             B.stride(0) if n > 1 else k, k,
                                             ctypes.c_void_p(0) if bv is None else _ptr(bv), _ptr(out),
                                             out.stride(0) if m > 1 else n, stream_ptr()))
    return out


def pool64(x, order, offsets, w=None):
    """Float64 class sums: out[g] = sum of w_r x_r over rows order[offsets[g]:offsets[g + 1]]."""
    torch = require_gpu()
    X = _rows64(x)
    n, d = X.shape
    C = len(offsets) - 1
    out = torch.zeros((C, d), dtype=torch.float64, device="cuda")
    if C and n:
        o_dev = torch.from_numpy(np.ascontiguousarray(order, dtype=np.int32)).cuda()
        f_dev = torch.from_numpy(np.ascontiguousarray(offsets, dtype=np.int64)).cuda()
        W = None if w is None else _vec64(w, n)
        _lib.check(_lib.load().hypmi_pool_rows_f64(_ptr(X), X.stride(0) if n > 1 else d, d, _ptr(o_dev),
                                                   _ptr(f_dev), C, ctypes.c_void_p(0) if W is None else _ptr(W),
                                                   _ptr(out), d, stream_ptr()))
    return out


def splda_posterior(T, N, lam):
    """In place T[i, j] /= 1 + N_i lam_j on a device float64 [m, K] buffer; returns the per-class
    log-normalisers (device [m]) and their sum (device scalar), see hypmi_splda_posterior."""
    torch = require_gpu()
    m, K = T.shape
    assert T.dtype == torch.float64 and T.is_cuda and (K == 1 or T.stride(1) == 1)
    Nd, ld = _vec64(N, m), _vec64(lam, K)
    logpy = torch.empty((m,), dtype=torch.float64, device="cuda")
    total = torch.zeros((1,), dtype=torch.float64, device="cuda")
    _lib.check(_lib.load().hypmi_splda_posterior(_ptr(T), m, T.stride(0) if m > 1 else K, K, _ptr(Nd),
                                                 _ptr(ld), _ptr(logpy), _ptr(total), stream_ptr()))
    return logpy, total


# fp32 error model of one score: |err| <= ERR_COEFF * (|g1| |g2| + |q1| + |q2|), 8 units of 2^-24
ERR_COEFF = 8.0 * 2.0 ** -24


def _as_c(a):
    """float64 C-contiguous copy + ctypes pointer (None stays NULL)."""
    if a is None:
        return None, ctypes.cast(None, _lib.c_double_p)
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_lib.c_double_p)


class Plan:
    """Device-resident folded model (hypmi_plan).  `segments` is a list of (A, b, d_out) affine
    maps each followed by a length normalisation; `tail` = (A, b) is applied after the last
    normalisation; `proj` = (At, bt, An, bn, c) is the PLDA projection (None for a
    transform-only plan, in which case the tail becomes a final un-normalised segment).
    `shared` = dict(P, pb, shift_t, scale_t, shift_n, scale_n) is the same projection in the basis
    where g and h are per-column maps of one K-column projection s = y P + pb (see
    hypmi_plan_desc): the device then projects onto K columns instead of 2K."""

    def __init__(self, d_in, segments=(), tail=(None, None), proj=None, shared=None):
        require_gpu()
        lib = _lib.load()
        from .fold import compose

        keep = []
        desc = _lib.PlanDesc()
        desc.d_in = int(d_in)
        segs = [(A, b, d, 1) for (A, b, d) in segments]
        d_last = segs[-1][2] if segs else d_in
        if proj is None:
            A, b = tail
            if A is not None or b is not None:
                d_last = A.shape[1] if A is not None else d_last
                segs.append((A, b, d_last, 0))
            desc.K = 0
        else:
            At, bt, An, bn, c = proj
            if shared is not None:
                (At2, bt2) = compose(tail, (shared["P"], shared["pb"]))
                for name in ("shift_t", "scale_t", "shift_n", "scale_n"):
                    a, ptr = _as_c(shared[name])
                    assert a.shape == (At2.shape[1],)
                    setattr(desc, name, ptr)
                    keep.append(a)
            else:
                (At2, bt2) = compose(tail, (At, bt))
                (An2, bn2) = compose(tail, (An, bn))
                a, desc.An = _as_c(An2)
                keep.append(a)
                a, desc.bn = _as_c(bn2)
                keep.append(a)
            a, desc.At = _as_c(At2)
            keep.append(a)
            a, desc.bt = _as_c(bt2)
            keep.append(a)
            desc.K = int(At2.shape[1])
            desc.c = float(c)
            if At2.shape[0] != d_last:
                raise ValueError("PLDA model expects %d-dim vectors, pre-processing gives %d"
                                 % (At2.shape[0], d_last))
        if len(segs) > _lib.HYPMI_MAX_SEG:
            raise ValueError("at most %d length-normalised segments are supported" % _lib.HYPMI_MAX_SEG)
        desc.n_seg = len(segs)
        for i, (A, b, d, ln) in enumerate(segs):
            a, desc.seg[i].A = _as_c(A)
            keep.append(a)
            a, desc.seg[i].b = _as_c(b)
            keep.append(a)
            desc.seg[i].d_out = int(d)
            desc.seg[i].lnorm = int(ln)
        handle = ctypes.c_void_p()
        _lib.check(lib.hypmi_plan_create(ctypes.byref(desc), ctypes.byref(handle)))
        self._h = handle
        self._lib = lib
        self.kp = lib.hypmi_plan_kp(handle)
        self.k = lib.hypmi_plan_k(handle)
        self.d_in = lib.hypmi_plan_d_in(handle)
        self.d_out = lib.hypmi_plan_d_out(handle)
        self.c = float(proj[4]) if proj is not None else 0.0
        del keep

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.hypmi_plan_destroy(h)
            except Exception:
                pass
            self._h = None

    def _workspace(self, n):
        torch = _torch()
        nbytes = self._lib.hypmi_prologue_workspace_bytes(self._h, n)
        return torch.empty((max(nbytes, 16) + 3) // 4, dtype=torch.float32, device="cuda"), nbytes

    def _check_x(self, x):
        if x.shape[1] != self.d_in:
            raise ValueError("expected %d-dim embeddings, got %d" % (self.d_in, x.shape[1]))

    def prologue(self, x, side):
        """x: device fp32 [n, d_in] (see upload) -> packed rows [n, kp]."""
        torch = _torch()
        self._check_x(x)
        n = x.shape[0]
        G = torch.empty((n, self.kp), dtype=torch.float32, device=x.device)
        if n == 0:
            return G
        ws, nbytes = self._workspace(n)
        _lib.check(self._lib.hypmi_prologue(self._h, _ptr(x), n, x.stride(0), int(side), _ptr(G),
                                            _ptr(ws), nbytes, stream_ptr()))
        return G

    def prologue_split(self, x, side, fmt=None, keep_rows=False, bounds=False):
        """x -> the operand of the 16-bit score kernels (SplitRows).  Plans with a fused shared-basis
        tail and the default fp16-pair format do it in the prologue kernel itself (one launch, the
        packed fp32 rows never reach memory unless keep_rows asks for them too: then (split, G)
        is returned); other plans, formats or widths convert the packed rows in a second kernel.
        bounds (only with pairs_in_prologue): also the device [n, 2] floats (|g|, |q|) of the rows, as
        the last element of the result."""
        torch = _torch()
        fmt = fmt or DEFAULT_SPLIT
        self._check_x(x)
        n = x.shape[0]
        if fmt == "pair" and n and self._lib.hypmi_prologue_pairs_supported(self._h):
            H = torch.empty(self._lib.hypmi_pair_bytes(self.kp, n) // 4, dtype=torch.int32, device=x.device)
            G = torch.empty((n, self.kp), dtype=torch.float32, device=x.device) if keep_rows else None
            bnd = torch.empty((n, 2), dtype=torch.float32, device=x.device) if bounds else None
            ws, nbytes = self._workspace(n)
            _lib.check(self._lib.hypmi_prologue_pairs_bounds(self._h, _ptr(x), n, x.stride(0), int(side), _ptr(H),
                                                             None, _ptr(G) if keep_rows else None,
                                                             _ptr(bnd) if bounds else None, _ptr(ws), nbytes,
                                                             stream_ptr()))
            split = SplitRows(H, n, self.kp, fmt)
            res = (split, G) if keep_rows else (split,)
            if bounds:
                res = res + (bnd,)
            return res if len(res) > 1 else split
        if bounds:
            raise ValueError("bounds come from the prologue kernel that writes fp16 pairs (pairs_in_prologue)")
        G = self.prologue(x, side)
        split = split_rows(G, fmt)
        if split is None:  # no 16-bit kernel for this width: the fp32 rows are the operand
            return (G, G) if keep_rows else G
        return (split, G) if keep_rows else split

    def pairs_in_prologue(self, fmt=None):
        """Whether the prologue kernel of this plan writes the fp16-pair operand itself."""
        return (fmt or DEFAULT_SPLIT) == "pair" and bool(self._lib.hypmi_prologue_pairs_supported(self._h))

    def prologue_split_both(self, x, fmt=None, bounds=False):
        """x -> (enrol-side, test-side) score operands of the same vectors (adaptive S-norm uses a test
        vector as a row of the cohort panel and as a column of the trial matrix).  One pass over the
        chain where the plan's prologue kernel writes fp16 pairs itself, two otherwise.  bounds (only
        with pairs_in_prologue): a third result, device [n, 2] floats (|g|, |q|) of the enrol-side rows
        (see prologue_f64)."""
        torch = _torch()
        fmt = fmt or DEFAULT_SPLIT
        self._check_x(x)
        n = x.shape[0]
        if self.pairs_in_prologue(fmt) and n:
            nb = self._lib.hypmi_pair_bytes(self.kp, n) // 4
            He = torch.empty(nb, dtype=torch.int32, device=x.device)
            Ht = torch.empty(nb, dtype=torch.int32, device=x.device)
            bnd = torch.empty((n, 2), dtype=torch.float32, device=x.device) if bounds else None
            ws, nbytes = self._workspace(n)
            _lib.check(self._lib.hypmi_prologue_pairs_bounds(self._h, _ptr(x), n, x.stride(0), SIDE_ENROL, _ptr(He),
                                                             _ptr(Ht), None, _ptr(bnd) if bounds else None,
                                                             _ptr(ws), nbytes, stream_ptr()))
            res = SplitRows(He, n, self.kp, fmt), SplitRows(Ht, n, self.kp, fmt)
            return res + (bnd,) if bounds else res
        if bounds:
            raise ValueError("bounds come from the prologue kernel that writes fp16 pairs (pairs_in_prologue)")
        return self.prologue_split(x, SIDE_ENROL, fmt), self.prologue_split(x, SIDE_TEST, fmt)

    def prologue_f64(self, x64, side):
        """x64: device float64 [n, d_in] -> (G64 [n, K + 2] rows [g, q(+c), |g|], bnd [n, 2] floats)."""
        torch = _torch()
        self._check_x(x64)
        n = x64.shape[0]
        G = torch.empty((n, self.k + 2), dtype=torch.float64, device=x64.device)
        bnd = torch.empty((n, 2), dtype=torch.float32, device=x64.device)
        if n:
            nbytes = self._lib.hypmi_prologue_f64_workspace_bytes(self._h, n)
            ws = torch.empty(max(nbytes // 8, 2), dtype=torch.float64, device=x64.device)
            _lib.check(self._lib.hypmi_prologue_f64(self._h, _ptr(x64), n, x64.stride(0), int(side),
                                                    _ptr(G), G.stride(0), _ptr(bnd), _ptr(ws), nbytes,
                                                    stream_ptr()))
        return G, bnd

    def transform(self, x):
        """x: device fp32 [n, d_in] -> device fp32 [n, d_out] (pre-processing chain only)."""
        torch = _torch()
        self._check_x(x)
        n = x.shape[0]
        ld = round_up(self.d_out, 4)
        y = torch.empty((n, ld), dtype=torch.float32, device=x.device)
        if ld != self.d_out:
            y.zero_()
        if n:
            ws, nbytes = self._workspace(n)
            _lib.check(self._lib.hypmi_transform(self._h, _ptr(x), n, x.stride(0), _ptr(y), ld,
                                                 _ptr(ws), nbytes, stream_ptr()))
        return y[:, :self.d_out]


# Operand formats of the 16-bit matrix-core score kernels: C entry points by format name.
SPLIT_FORMATS = {
    "pair": ("hypmi_pair_row_bytes", "hypmi_pair_bytes", "hypmi_pair_rows", "hypmi_score_pair"),
    "triple": ("hypmi_split_row_bytes", "hypmi_split_bytes", "hypmi_split_rows", "hypmi_score_split"),
}
DEFAULT_SPLIT = "pair"  # scaled fp16 pairs; "triple": exact bf16 triples


class SplitRows:
    """Packed rows converted for a 16-bit matrix-core score kernel (fmt: "pair" or "triple")."""

    def __init__(self, H, n, kp, fmt):
        self.H, self.n, self.kp, self.fmt = H, n, kp, fmt

    @property
    def shape(self):
        return (self.n, self.kp)

    @property
    def device(self):
        return self.H.device

    def rows(self, r0, r1):
        """Rows r0..r1-1 as SplitRows on the same storage (rows are stored one after the other; the
        kernels may read up to 1 KiB past the last row of a tile, which stays inside the parent)."""
        assert 0 <= r0 <= r1 <= self.n
        if r0 == 0 and r1 == self.n:
            return self
        row_i32 = getattr(_lib.load(), SPLIT_FORMATS[self.fmt][0])(self.kp) // 4
        return SplitRows(self.H[r0 * row_i32:], r1 - r0, self.kp, self.fmt)


def split_rows(G, fmt=None):
    """Packed fp32 rows [n, kp] -> SplitRows, or None when kp has no split kernel."""
    torch = _torch()
    lib = _lib.load()
    fmt = fmt or DEFAULT_SPLIT
    row_bytes_fn, bytes_fn, rows_fn, _ = SPLIT_FORMATS[fmt]
    n, kp = G.shape
    if getattr(lib, row_bytes_fn)(kp) == 0:
        return None
    if G.stride(1) != 1 or G.stride(0) != kp:
        G = G.contiguous()
    H = torch.empty(getattr(lib, bytes_fn)(kp, n) // 4, dtype=torch.int32, device=G.device)
    if n:
        _lib.check(getattr(lib, rows_fn)(_ptr(G), n, kp, _ptr(H), stream_ptr()))
    return SplitRows(H, n, kp, fmt)


def score(G1, G2, out=None, split=True):
    """Packed enrol rows x packed test rows -> device fp32 score matrix [n1, n2].  Either side may
    already be SplitRows; with split=True (or a format name) fp32 rows are converted when their
    width allows it (16-bit matrix cores at fp32 accuracy), otherwise the fp32 kernel runs."""
    torch = _torch()
    lib = _lib.load()
    n1, kp = G1.shape
    n2, kp2 = G2.shape
    if kp != kp2:
        raise ValueError("packed widths differ: %d vs %d" % (kp, kp2))
    given = [g.fmt for g in (G1, G2) if isinstance(g, SplitRows)]
    fmt = given[0] if given else (split if isinstance(split, str) else DEFAULT_SPLIT)
    if len(given) == 2 and given[0] != given[1]:
        raise ValueError("the two sides are split in different formats: %s and %s" % tuple(given))
    if given or (split and getattr(lib, SPLIT_FORMATS[fmt][0])(kp) != 0 and n1 and n2):
        H1 = G1 if isinstance(G1, SplitRows) else split_rows(G1, fmt)
        H2 = G2 if isinstance(G2, SplitRows) else split_rows(G2, fmt)
        if H1 is None or H2 is None:
            raise ValueError("packed width %d has no split kernel" % kp)
        if out is None:
            ld = round_up(max(n2, 1), 32)
            out = torch.empty((n1, ld), dtype=torch.float32, device=H1.device)[:, :n2]
        if n1 and n2:
            _lib.check(getattr(lib, SPLIT_FORMATS[fmt][3])(_ptr(H1.H), n1, _ptr(H2.H), n2, kp, _ptr(out),
                                                          out.stride(0), stream_ptr()))
        return out
    if out is None:
        ld = round_up(max(n2, 1), 32)  # full 128-byte lines per stored row segment
        buf = torch.empty((n1, ld), dtype=torch.float32, device=G1.device)
        out = buf[:, :n2]
    if n1 and n2:
        _lib.check(lib.hypmi_score(_ptr(G1), n1, _ptr(G2), n2, kp, _ptr(out), out.stride(0),
                                   stream_ptr()))
    return out


def refine_scores(S, G1_64, bnd1, G2_64, bnd2, rel_tol=1e-4, err_coeff=ERR_COEFF, count=False):
    """Replace in place the entries of S whose fp32 error bound is not below rel_tol * |s| by their
    float64 value.  Returns the number of refined entries when count is set."""
    torch = _torch()
    lib = _lib.load()
    n1, n2 = S.shape
    K = G1_64.shape[1] - 2
    counter = torch.zeros(1, dtype=torch.int64, device=S.device) if count else None
    if n1 and n2:
        _lib.check(lib.hypmi_refine_scores(_ptr(S), S.stride(0), n1, n2, _ptr(G1_64), _ptr(G2_64),
                                           G1_64.stride(0), K, _ptr(bnd1), _ptr(bnd2), float(rel_tol),
                                           float(err_coeff), _ptr(counter) if count else None,
                                           stream_ptr()))
    return int(counter.item()) if count else None


_staging = threading.local()  # page-locked buffers, one set per calling thread


def to_host(t, dtype):
    """Device tensor -> numpy array of the host float type.  Large results go through page-locked
    staging buffers of at most STAGE_BYTES (pageable device-to-host copies run at a fraction of the link
    rate), in row slices, and are widened to the host type by torch's threaded copy rather than numpy's
    serial astype; the widening of slice i overlaps the transfer of slice i + 1."""
    torch = _torch()
    n = t.numel()
    if n < (1 << 16) or t.dtype != torch.float32 or t.dim() != 2:
        return t.cpu().numpy().astype(dtype, copy=False)
    r, c = t.shape
    out = np.empty((r, c), dtype=dtype)
    dst = torch.from_numpy(out)
    rows = max(1, min(r, STAGE_BYTES // max(1, 4 * c)))
    bufs = [_pinned("down32_%d" % i, rows * c, torch.float32) for i in range(2 if rows < r else 1)]
    stream = torch.cuda.current_stream()
    pending = None  # (buffer view, destination rows, event) of the slice in flight
    for i, r0 in enumerate(range(0, r, rows)):
        r1 = min(r, r0 + rows)
        stage = bufs[i % len(bufs)][:(r1 - r0) * c].view(r1 - r0, c)
        stage.copy_(t[r0:r1], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream)
        if pending is not None:
            pending[2].synchronize()
            dst[pending[1][0]:pending[1][1]].copy_(pending[0])
        pending = (stage, (r0, r1), ev)
    pending[2].synchronize()
    dst[pending[1][0]:pending[1][1]].copy_(pending[0])
    return out
