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


def _pool(runs):
    """Pool-adjacent-violators on run-length blocks (width, ones): merges while the level of the
    previous block >= the level of the current one (utils.py:30-94), with exact integer
    comparisons.  Returns the pooled (width, ones) lists and, per pooled block, the index of its
    first run."""
    width, ones, first = [], [], []
    for i, (w, o) in enumerate(runs):
        if w == 0:
            continue
        width.append(w)
        ones.append(o)
        first.append(i)
        # level[-2] >= level[-1]  <=>  ones[-2] * width[-1] >= ones[-1] * width[-2]
        while len(width) >= 2 and ones[-2] * width[-1] >= ones[-1] * width[-2]:
            width[-2] += width[-1]
            ones[-2] += ones[-1]
            width.pop()
            ones.pop()
            first.pop()
    return width, ones, first


def _hull_from_pooled(width, ones, n_tar, n_non):
    """Cumulative miss / false-alarm rates at the block boundaries (roc.py:81-100)."""
    left_tar = np.concatenate(([0], np.cumsum(np.asarray(ones, dtype=np.int64))))
    left_all = np.concatenate(([0], np.cumsum(np.asarray(width, dtype=np.int64))))
    return left_tar / float(n_tar), (n_non - (left_all - left_tar)) / float(n_non)


def _exact_runs(tar_scores, non_scores):
    """Label runs (width, ones) of the two sorted score lists merged in ascending order, targets
    before non-targets at equal scores (the stable sort of roc.py:67-72)."""
    if len(tar_scores) == 0:
        return [(len(non_scores), 0)]
    if len(non_scores) == 0:
        return [(len(tar_scores), len(tar_scores))]
    # a non-target comes after every target <= it; counts of non-targets between consecutive targets
    pos = np.searchsorted(tar_scores, non_scores, side="right")
    between = np.bincount(pos, minlength=len(tar_scores) + 1)
    runs = []
    start = 0  # targets are emitted in maximal runs between non-empty groups of non-targets
    for i in np.flatnonzero(between):
        if i > start:
            runs.append((int(i - start), int(i - start)))
        runs.append((int(between[i]), 0))
        start = i
    if len(tar_scores) > start:
        runs.append((int(len(tar_scores) - start), int(len(tar_scores) - start)))
    return runs


def rocch_from_counts(tar_counts, non_counts, refined=None, optimistic=False, return_open=False):
    """ROC convex hull vertices (p_miss, p_fa) of binned scores; bins in ascending score order.

    Sorted label sequence of the reference (roc.py:67-72): within a bin the targets (label 1) come
    first, then the non-targets (label 0).  Pool-adjacent-violators on that sequence is run on
    run-length blocks; the vertices are the cumulative miss / false alarm rates at the block
    boundaries (roc.py:81-100).

    refined: {bin: (sorted target scores, sorted non-target scores) or the label runs [(width,
    ones), ...] themselves} for bins whose scores are known exactly; they contribute their true
    label sequence.  optimistic: the other bins put their
    non-targets first (the best ordering a bin can have); return_open then adds the bins where that
    ordering leaves a vertex inside the bin, the only ones whose true order can change the hull."""
    tar_counts = np.asarray(tar_counts, dtype=np.int64)
    non_counts = np.asarray(non_counts, dtype=np.int64)
    assert tar_counts.shape == non_counts.shape and tar_counts.ndim == 1
    n_tar, n_non = int(tar_counts.sum()), int(non_counts.sum())
    assert n_tar > 0 and n_non > 0, "a ROC needs target and non-target trials"
    refined = refined or {}
    runs, inner = [], {}  # inner: bin -> index of the run that follows its internal boundary
    for b, (t, n) in enumerate(zip(tar_counts.tolist(), non_counts.tolist())):
        if b in refined:
            exact = refined[b] if isinstance(refined[b], list) else _exact_runs(*refined[b])
            assert sum(o for _, o in exact) == t and sum(w - o for w, o in exact) == n, \
                "refined scores of bin %d do not match its counts" % b
            runs += exact
        elif optimistic:
            if t and n:
                inner[b] = len(runs) + 1
            runs += [(n, 0), (t, t)]
        else:
            runs += [(t, t), (n, 0)]
    width, ones, first = _pool(runs)
    hull = _hull_from_pooled(width, ones, n_tar, n_non)
    if not return_open:
        return hull
    starts = set(first)
    return hull + (np.array(sorted(b for b, i in inner.items() if i in starts), dtype=np.int64),)


def bins_that_can_hold_a_vertex(tar_counts, non_counts, refined=None):
    """Bins whose true score order can change the hull: those with both classes whose best possible
    ordering (all non-targets below all targets) gives a point strictly better than the hull of the
    worst ordering.  The exact hull lies between the two, so once these bins are known exactly no
    other bin can contribute a vertex: ONE second pass makes the hull exact."""
    tar_counts = np.asarray(tar_counts, dtype=np.int64)
    non_counts = np.asarray(non_counts, dtype=np.int64)
    refined = refined or {}
    p_miss, p_fa = rocch_from_counts(tar_counts, non_counts, refined)  # achievable everywhere
    cum_t = np.concatenate(([0], np.cumsum(tar_counts)))
    cum_n = np.concatenate(([0], np.cumsum(non_counts)))
    best_miss = cum_t[:-1] / float(cum_t[-1])        # none of the bin's targets missed ...
    best_fa = 1.0 - cum_n[1:] / float(cum_n[-1])     # ... with all of its non-targets rejected
    hull_miss = np.interp(best_fa, p_fa[::-1], p_miss[::-1])  # p_fa falls along the hull
    mixed = (tar_counts > 0) & (non_counts > 0)
    done = np.zeros(len(tar_counts), dtype=bool)
    done[list(refined)] = True
    return np.flatnonzero(mixed & ~done & (best_miss < hull_miss))


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


def _p(t):
    return ctypes.c_void_p(0) if t is None else engine._ptr(t)


class ExactPass:
    """Second pass of a TrialHistogram over the same score blocks: collects the scores of the bins
    that can hold a vertex of the ROC convex hull, so that the hull becomes the reference's own.
    Use like the histogram (add / sink), then finish()."""

    def __init__(self, hist, bins):
        torch = engine.require_gpu()
        self.h = hist
        self.bins = np.asarray(bins, dtype=np.int64)
        non, tar = hist.counts()
        self.expect = (int(non[self.bins].sum()), int(tar[self.bins].sum()))
        self.cap = max(1, max(self.expect))
        want = np.zeros((2, hist.nbins), dtype=np.uint8)
        want[:, self.bins] = 1
        self.want = torch.from_numpy(want).cuda()
        self.out = torch.empty((2, self.cap), dtype=torch.float32, device="cuda")
        self.count = torch.zeros((2,), dtype=torch.int64, device="cuda")

    def add(self, S, key_enrol=None, key_test=None, tar=None, non=None):
        h = self.h
        S, lds, ke, kt, tm, nm, ldm = h._labels(S, key_enrol, key_test, tar, non)
        _lib.check(_lib.load().hypmi_score_extract(
            engine._ptr(S), lds, S.shape[0], S.shape[1], _p(ke), _p(kt), _p(tm), _p(nm), ldm, h.lo, h.hi, h.nbins,
            engine._ptr(self.want), engine._ptr(self.out), self.cap, engine._ptr(self.count), engine.stream_ptr()))
        return self

    def sink(self, key_enrol, key_test):
        return TrialHistogram.sink(self, key_enrol, key_test)

    def _bin_of(self, s):
        """The kernel's bin rule in float32: floor((s - lo) * (nbins / (hi - lo))), clamped."""
        torch = engine.require_gpu()
        h = self.h
        lo32 = np.float32(h.lo)
        scale32 = np.float32(h.nbins) / (np.float32(h.hi) - lo32)
        return torch.floor((s - float(lo32)) * float(scale32)).clamp_(0, h.nbins - 1).to(torch.int64)

    def all_gather(self, group=None):
        """Several GPUs: every rank ends up with the scores all ranks collected (call after the pass,
        before finish; the histogram must have been all-reduced before exact_pass)."""
        import torch.distributed as dist

        torch = engine.require_gpu()
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return self
        from ..backend import settle_for_gloo

        world = dist.get_world_size(group)
        counts = torch.empty((world, 2), dtype=torch.int64, device="cuda")
        settle_for_gloo(self.out, group)
        dist.all_gather_into_tensor(counts, self.count.view(1, 2), group=group)
        lists = torch.empty((world,) + tuple(self.out.shape), dtype=torch.float32, device="cuda")
        dist.all_gather_into_tensor(lists, self.out.unsqueeze(0).contiguous(), group=group)
        counts_h = counts.cpu().numpy()
        merged = torch.empty_like(self.out)
        for c in range(2):
            parts = [lists[r, c, :min(int(counts_h[r, c]), self.cap)] for r in range(world)]
            cat = torch.cat(parts)[:self.cap]
            merged[c, :cat.numel()] = cat
        self.out = merged
        self.count = counts.sum(dim=0)
        return self

    def finish(self):
        """Sorts what was collected (on the device) and hands the histogram the exact label runs
        of the refined bins.  The blocks of the second pass must be the blocks of the first."""
        torch = engine.require_gpu()
        got = tuple(int(c) for c in self.count.cpu())
        if got != self.expect:
            raise RuntimeError("second pass collected %s (non-target, target) scores where the histogram has %s: "
                               "the two passes did not see the same scores" % (got, self.expect))
        non_s = torch.sort(self.out[0, :got[0]])[0]
        tar_s = torch.sort(self.out[1, :got[1]])[0]
        counts_non, counts_tar = self.h.counts()
        t_b = counts_tar[self.bins]
        # slot of a non-target: (its refined bin, number of that bin's targets <= it)
        rank_of_bin = torch.full((self.h.nbins,), -1, dtype=torch.int64, device="cuda")
        rank_of_bin[torch.from_numpy(self.bins).cuda()] = torch.arange(len(self.bins), device="cuda")
        first_tar = torch.from_numpy(np.concatenate(([0], np.cumsum(t_b)))[:-1]).cuda()
        slot0 = torch.from_numpy(np.concatenate(([0], np.cumsum(t_b + 1)))[:-1]).cuda()
        r = rank_of_bin[self._bin_of(non_s)]
        local = torch.searchsorted(tar_s, non_s, right=True) - first_tar[r]
        between = torch.bincount(slot0[r] + local, minlength=int((t_b + 1).sum())).cpu().numpy()
        at = 0
        for b, t in zip(self.bins.tolist(), t_b.tolist()):
            zeros = between[at:at + t + 1]
            at += t + 1
            runs = [(int(zeros[0]), 0)]
            gaps = np.flatnonzero(zeros[1:]) + 1          # targets after which non-targets follow
            prev = 0
            for g in gaps.tolist():
                runs += [(g - prev, g - prev), (int(zeros[g]), 0)]
                prev = g
            if t > prev:
                runs.append((t - prev, t - prev))
            self.h.refined[b] = [x for x in runs if x[0] > 0]
        return self.h


class TrialHistogram:
    """Accumulates score blocks on the device.  lo / hi: score range covered by `nbins` equal bins
    (scores outside fall into the first / last bin)."""

    def __init__(self, lo, hi, nbins=4096):
        torch = engine.require_gpu()
        assert hi > lo and 1 <= nbins <= 8192
        self.lo, self.hi, self.nbins = float(lo), float(hi), int(nbins)
        self.hist = torch.zeros((2, self.nbins), dtype=torch.int64, device="cuda")
        self.mom = torch.zeros((2, 4), dtype=torch.float64, device="cuda")
        self.refined = {}  # bin -> exact label runs, filled by a second pass (see exact_pass)

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

    def _labels(self, S, key_enrol, key_test, tar, non):
        """Device arguments of one block: (S, lds, ke, kt, tar mask, non mask, ldm)."""
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
        lds = S.stride(0) if ne > 1 else max(nt, S.stride(0))
        return S, lds, ke, kt, tm, nm, ldm

    def add(self, S, key_enrol=None, key_test=None, tar=None, non=None):
        """Adds the device score block S [n_enrol, n_test].  Labels: integer keys per enrol row and
        test column (target iff equal, negative = not a trial), or TrialKey-style boolean matrices
        `tar` (and `non`), or nothing (all scores counted as non-targets)."""
        S, lds, ke, kt, tm, nm, ldm = self._labels(S, key_enrol, key_test, tar, non)
        _lib.check(_lib.load().hypmi_score_hist(
            engine._ptr(S), lds, S.shape[0], S.shape[1], _p(ke), _p(kt), _p(tm), _p(nm), ldm, self.lo, self.hi,
            self.nbins, engine._ptr(self.hist), engine._ptr(self.mom), engine.stream_ptr()))
        return self

    def sink(self, key_enrol, key_test):
        """A `reduce` callback for PLDABackend.score_snorm: block [rows, chunk] of the enrolment rows
        starting at r0 against the tests starting at column c0."""
        torch = engine.require_gpu()
        kt_all = key_test if isinstance(key_test, torch.Tensor) else torch.from_numpy(
            np.ascontiguousarray(key_test, dtype=np.int32)).cuda()
        ke_all = key_enrol if isinstance(key_enrol, torch.Tensor) else torch.from_numpy(
            np.ascontiguousarray(key_enrol, dtype=np.int32)).cuda()

        def reduce(c0, block, r0=0):
            self.add(block, ke_all[r0:r0 + block.shape[0]], kt_all[c0:c0 + block.shape[1]])

        return reduce

    def all_reduce(self, group=None):
        """Sums the histograms and moments of all ranks (a few kB; the only collective a sharded
        evaluation needs after scoring)."""
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            from ..backend import settle_for_gloo

            settle_for_gloo(self.hist, group)
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

    def exact_pass(self):
        """The second pass that makes rocch / eer / min_dcf exact, or None when the histograms
        already determine the hull (see bins_that_can_hold_a_vertex).  With several GPUs call
        all_reduce() first, run the pass on every rank's own blocks, then ExactPass.all_gather()."""
        non, tar = self.counts()
        todo = bins_that_can_hold_a_vertex(tar, non, self.refined)
        return ExactPass(self, todo) if len(todo) else None

    def is_exact(self):
        non, tar = self.counts()
        return len(bins_that_can_hold_a_vertex(tar, non, self.refined)) == 0

    def rocch(self):
        non, tar = self.counts()
        return rocch_from_counts(tar, non, self.refined)

    def eer(self):
        return eer_from_rocch(*self.rocch())

    def min_dcf(self, prior, normalize=True):
        return min_dcf_from_rocch(*self.rocch(), prior, normalize)
