"""Target / non-target score histograms accumulated on the device, and ROC convex hull metrics from
them.

The reference evaluates a trial set by pulling the target and the non-target scores out of the score
matrix with the boolean masks of a TrialKey (hyperion/utils/trial_scores.py:342-357 `get_tar_non`),
concatenating and sorting them (hyperion/metrics/roc.py:67-72) and running pool-adjacent-violators
over the sorted labels (hyperion/metrics/utils.py:30-94) to get the ROC convex hull, from which EER
(hyperion/metrics/roc.py:105-140, eer.py:11-22) and minimum DCF (dcf.py:36-65) follow.  For 1e10 and
more trials the matrix cannot be kept, let alone sorted on the host; here every block of scores is
reduced on the device to two fixed-size histograms (hypmi_score_hist) and the same convex hull is
built from the bin counts: it is the reference's result for scores rounded to their bin, targets
before non-targets inside a bin exactly as the reference's stable sort of `[tar, non]` orders ties.
"""
import ctypes

import numpy as np

from .. import _lib, engine


def rocch_from_counts(tar_counts, non_counts):
    """ROC convex hull vertices (p_miss, p_fa) of binned scores; bins in ascending score order.

    Sorted label sequence of the reference (roc.py:67-72): within a bin the targets (label 1) come
    first, then the non-targets (label 0).  Pool-adjacent-violators on that sequence (utils.py:30-94:
    merge while the level of the previous block >= the level of the current one) is run on
    run-length blocks with exact integer comparisons; the vertices are the cumulative miss / false
    alarm rates at the block boundaries (roc.py:81-100)."""
    tar_counts = np.asarray(tar_counts, dtype=np.int64)
    non_counts = np.asarray(non_counts, dtype=np.int64)
    assert tar_counts.shape == non_counts.shape and tar_counts.ndim == 1
    n_tar, n_non = int(tar_counts.sum()), int(non_counts.sum())
    assert n_tar > 0 and n_non > 0, "a ROC needs target and non-target trials"
    width, ones = [], []  # stack of pooled blocks
    for t, n in zip(tar_counts.tolist(), non_counts.tolist()):
        for w, o in ((t, t), (n, 0)):
            if w == 0:
                continue
            width.append(w)
            ones.append(o)
            # level[-2] >= level[-1]  <=>  ones[-2] * width[-1] >= ones[-1] * width[-2]
            while len(width) >= 2 and ones[-2] * width[-1] >= ones[-1] * width[-2]:
                width[-2] += width[-1]
                ones[-2] += ones[-1]
                width.pop()
                ones.pop()
    left_tar = np.concatenate(([0], np.cumsum(np.asarray(ones, dtype=np.int64))))
    left_all = np.concatenate(([0], np.cumsum(np.asarray(width, dtype=np.int64))))
    p_miss = left_tar / float(n_tar)
    p_fa = (n_non - (left_all - left_tar)) / float(n_non)
    return p_miss, p_fa


def eer_from_rocch(p_miss, p_fa):
    """Equal error rate of a convex hull (roc.py:105-140): for every hull segment that is neither
    horizontal nor vertical, the point of the line through its end points where p_miss = p_fa; the
    maximum over segments."""
    x0, x1, y0, y1 = p_fa[:-1], p_fa[1:], p_miss[:-1], p_miss[1:]
    live = (x0 != x1) & (y0 != y1)
    if not np.any(live):
        return 0.0
    det = x0[live] * y1[live] - y0[live] * x1[live]
    den = (y1[live] - y0[live]) + (x0[live] - x1[live])
    return float(max(0.0, np.max(det / den)))


def min_dcf_from_rocch(p_miss, p_fa, prior, normalize=True):
    """Minimum detection cost over the hull vertices for a target prior (dcf.py:36-65)."""
    dcf = prior * p_miss + (1.0 - prior) * p_fa
    if normalize:
        dcf = dcf / min(prior, 1.0 - prior)
    i = int(np.argmin(dcf))
    return float(dcf[i]), float(p_miss[i]), float(p_fa[i])


class TrialHistogram:
    """Accumulates score blocks on the device.  lo / hi: score range covered by `nbins` equal bins
    (scores outside fall into the first / last bin)."""

    def __init__(self, lo, hi, nbins=4096):
        torch = engine.require_gpu()
        assert hi > lo and 1 <= nbins <= 8192
        self.lo, self.hi, self.nbins = float(lo), float(hi), int(nbins)
        self.hist = torch.zeros((2, self.nbins), dtype=torch.int64, device="cuda")
        self.mom = torch.zeros((2, 4), dtype=torch.float64, device="cuda")

    @staticmethod
    def _keys(k, n):
        torch = engine.require_gpu()
        if isinstance(k, torch.Tensor):
            t = k.to(device="cuda", dtype=torch.int32).contiguous()
        else:
            t = torch.from_numpy(np.ascontiguousarray(k, dtype=np.int32)).cuda()
        assert t.dim() == 1 and t.shape[0] == n, "one key per row / column of the score block"
        return t

    @staticmethod
    def _mask(m, shape):
        torch = engine.require_gpu()
        if not isinstance(m, torch.Tensor):
            m = torch.from_numpy(np.ascontiguousarray(m).astype(np.uint8, copy=False))
        t = m.to(device="cuda")
        t = (t != 0).to(torch.uint8) if t.dtype != torch.uint8 else t
        assert tuple(t.shape) == tuple(shape), "mask and score block differ in shape"
        return t if t.stride(1) == 1 else t.contiguous()

    def add(self, S, key_enrol=None, key_test=None, tar=None, non=None):
        """Adds the device score block S [n_enrol, n_test].  Labels: integer keys per enrol row and
        test column (target iff equal, negative = not a trial), or TrialKey-style boolean matrices
        `tar` (and `non`), or nothing (all scores counted as non-targets)."""
        torch = engine.require_gpu()
        assert isinstance(S, torch.Tensor) and S.is_cuda and S.dtype == torch.float32 and S.dim() == 2
        if S.stride(1) != 1:
            S = S.contiguous()
        ne, nt = S.shape
        ke = kt = tm = nm = None
        ldm = 0
        if key_enrol is not None or key_test is not None:
            assert key_enrol is not None and key_test is not None and tar is None and non is None
            ke, kt = self._keys(key_enrol, ne), self._keys(key_test, nt)
        elif tar is not None:
            tm = self._mask(tar, S.shape)
            ldm = tm.stride(0) if ne > 1 else max(nt, tm.stride(0))
            if non is not None:
                nm = self._mask(non, S.shape)
                if ne > 1 and nm.stride(0) != ldm:
                    tm, nm = tm.contiguous(), nm.contiguous()
                    ldm = nt
        else:
            assert non is None, "a non-target mask needs a target mask"
        null = ctypes.c_void_p(0)
        p = lambda t: null if t is None else engine._ptr(t)  # noqa: E731
        lds = S.stride(0) if ne > 1 else max(nt, S.stride(0))
        _lib.check(_lib.load().hypmi_score_hist(
            engine._ptr(S), lds, ne, nt, p(ke), p(kt), p(tm), p(nm), ldm, self.lo, self.hi, self.nbins,
            engine._ptr(self.hist), engine._ptr(self.mom), engine.stream_ptr()))
        return self

    def sink(self, key_enrol, key_test):
        """A `reduce` callback for PLDABackend.score_snorm: block [n_enrol, chunk] of the tests
        starting at column c0."""
        torch = engine.require_gpu()
        kt_all = key_test if isinstance(key_test, torch.Tensor) else torch.from_numpy(
            np.ascontiguousarray(key_test, dtype=np.int32)).cuda()

        def reduce(c0, block):
            self.add(block, key_enrol, kt_all[c0:c0 + block.shape[1]])

        return reduce

    def all_reduce(self, group=None):
        """Sums the histograms and moments of all ranks (a few kB; the only collective a sharded
        evaluation needs after scoring)."""
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.hist, group=group)
            dist.all_reduce(self.mom, group=group)
        return self

    # ---- host side (synchronises) ---------------------------------------------------------------
    def counts(self):
        """(non_counts, tar_counts) as int64 numpy arrays of nbins entries."""
        h = self.hist.cpu().numpy()
        return h[0].copy(), h[1].copy()

    def moments(self):
        """Per class: number of finite scores, their mean and population standard deviation, and the
        number of non-finite scores that were left out."""
        m = self.mom.cpu().numpy()
        out = {}
        for name, row in (("non", m[0]), ("tar", m[1])):
            n = row[0]
            mean = row[1] / n if n > 0 else float("nan")
            var = row[2] / n - mean * mean if n > 0 else float("nan")
            out[name] = {"count": int(n), "mean": float(mean), "std": float(np.sqrt(max(var, 0.0))),
                         "nonfinite": int(row[3])}
        return out

    def rocch(self):
        non, tar = self.counts()
        return rocch_from_counts(tar, non)

    def eer(self):
        return eer_from_rocch(*self.rocch())

    def min_dcf(self, prior, normalize=True):
        return min_dcf_from_rocch(*self.rocch(), prior, normalize)
