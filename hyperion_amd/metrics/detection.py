"""Detection metrics of explicit target / non-target score arrays, with the reference's function
names and return values (hyperion/metrics/roc.py:54-140, eer.py:11-39, dcf.py:11-175,
utils.py:13-27).

The reference sorts the concatenated scores on the host and runs pool-adjacent-violators over
every score in a Python loop.  Here the two arrays are sorted on the device (float64, so the order
is the reference's), merged into label runs — targets before non-targets at equal scores, the order
the reference's stable sort of `[tar, non]` gives — and pooled with exact integer comparisons
(trial_hist._pool): the hull vertices are the reference's, except that a point lying exactly on a
hull segment is never kept (the reference sometimes keeps one, through a floating-point comparison
of equal block means; EER, DCF and PRBEP do not depend on it).  For trial sets that do not fit as
arrays use TrialHistogram, which takes the score blocks as they are produced.
"""
import numpy as np

from .. import engine
from .trial_hist import _exact_runs, _hull_from_pooled, _pool, eer_from_rocch


def _sorted(scores):
    """float64 copy in ascending order, sorted on the device."""
    torch = engine.require_gpu()
    if isinstance(scores, torch.Tensor):
        t = scores.detach().reshape(-1).to(device="cuda", dtype=torch.float64)
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(scores).reshape(-1), dtype=np.float64)).cuda()
    return torch.sort(t).values.cpu().numpy()


def _rocch_sorted(tar, non):
    assert len(tar) > 0 and len(non) > 0, "a ROC needs target and non-target trials"
    width, ones, _ = _pool(_exact_runs(tar, non))
    return _hull_from_pooled(width, ones, len(tar), len(non))


def compute_rocch(tar_scores, non_scores):
    """ROC convex hull: (p_miss, p_fa) at its vertices, from the lowest threshold up (roc.py:54-102)."""
    return _rocch_sorted(_sorted(tar_scores), _sorted(non_scores))


def rocch2eer(p_miss, p_fa):
    """Equal error rate of a hull (roc.py:105-140)."""
    return eer_from_rocch(np.asarray(p_miss, dtype=np.float64), np.asarray(p_fa, dtype=np.float64))


def compute_eer(tar, non):
    """eer.py:11-22."""
    return rocch2eer(*compute_rocch(tar, non))


def compute_prbep(tar, non):
    """Precision-recall break-even point, where #misses = #false alarms (eer.py:25-39)."""
    p_miss, p_fa = compute_rocch(tar, non)
    return rocch2eer(p_miss * np.size(tar), p_fa * np.size(non))


def effective_prior(p_tar, c_miss, c_fa):
    """Prior that carries the two costs (utils.py:13-27)."""
    beta = p_tar * c_miss / (1 - p_tar) / c_fa
    return beta / (1 + beta)


def compute_dcf(p_miss, p_fa, prior, normalize=True):
    """prior p_miss + (1 - prior) p_fa for every prior and point: [len(prior), len(p_miss)], or
    [len(p_miss)] for a scalar prior (dcf.py:11-33)."""
    prior = np.asarray(prior)
    if prior.ndim == 1:
        prior = prior[:, None]
    dcf = prior * p_miss + (1 - prior) * p_fa
    if normalize:
        dcf = dcf / np.minimum(prior, 1 - prior)
    return dcf


def _min_dcf(p_miss, p_fa, prior, normalize):
    dcf = compute_dcf(p_miss, p_fa, prior, normalize)
    idx = np.argmin(dcf, axis=-1)
    if dcf.ndim == 1:
        return dcf[idx], p_miss[idx], p_fa[idx]
    return dcf[np.arange(dcf.shape[0]), idx], p_miss[idx], p_fa[idx]


def compute_min_dcf(tar, non, prior, normalize=True):
    """(minimum DCF, p_miss there, p_fa there) for a prior or a vector of priors (dcf.py:36-65)."""
    return _min_dcf(*compute_rocch(tar, non), prior, normalize)


def _act_dcf_sorted(tar, non, prior, normalize):
    prior = np.asarray(prior)
    if prior.ndim == 1:
        assert np.all(prior == np.sort(prior, kind="mergesort")), "priors must be in ascending order"
    else:
        prior = prior[None]
    t = -np.log(prior) + np.log(1 - prior)
    # the reference ranks the thresholds inside stable sorts of [thresholds, scores] (dcf.py:99-113):
    # a score equal to a threshold sorts after it, so it is not a miss and it is a false alarm
    p_miss = np.searchsorted(tar, t, side="left") / len(tar)
    p_fa = (len(non) - np.searchsorted(non, t, side="left")) / len(non)
    act_dcf = prior * p_miss + (1 - prior) * p_fa
    if normalize:
        act_dcf = act_dcf / np.minimum(prior, 1 - prior)
    if len(act_dcf) == 1:
        act_dcf = act_dcf[0]
    return act_dcf, p_miss, p_fa


def compute_act_dcf(tar, non, prior, normalize=True):
    """Detection cost at the Bayes thresholds log((1 - prior) / prior), the scores taken as
    log-likelihood ratios: (actual DCF, p_miss, p_fa) (dcf.py:68-136)."""
    return _act_dcf_sorted(_sorted(tar), _sorted(non), prior, normalize)


def fast_eval_dcf_eer(tar, non, prior, normalize_dcf=True, return_probs=False):
    """(min DCF, actual DCF, EER, PRBEP) from one sort and one hull; with return_probs also the
    operating points of the two costs (dcf.py:139-175)."""
    tar, non = _sorted(tar), _sorted(non)
    p_miss, p_fa = _rocch_sorted(tar, non)
    eer = rocch2eer(p_miss, p_fa)
    prbep = rocch2eer(p_miss * len(tar), p_fa * len(non))
    dcf = compute_dcf(p_miss, p_fa, prior, normalize_dcf)
    min_dcf = np.min(dcf, axis=-1)
    act_dcf, act_pmiss, act_pfa = _act_dcf_sorted(tar, non, prior, normalize_dcf)
    if not return_probs:
        return min_dcf, act_dcf, eer, prbep
    idx = np.argmin(dcf, axis=-1)
    return min_dcf, act_dcf, eer, prbep, p_miss[idx], p_fa[idx], act_pmiss, act_pfa


def _merged_order(tar, non):
    """(labels in ascending score order, the sorting permutation) of the concatenation [tar, non],
    stable - the order numpy's mergesort gives the reference - sorted on the device in float64."""
    torch = engine.require_gpu()
    f = lambda a: (a.detach().reshape(-1).to(device="cuda", dtype=torch.float64) if isinstance(a, torch.Tensor)  # noqa: E731
                   else torch.from_numpy(np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=np.float64)).cuda())
    t, n = f(tar), f(non)
    order = torch.sort(torch.cat((t, n)), stable=True).indices
    return order < t.numel(), order


def compute_roc(true_scores, false_scores):
    """The observed ROC: (p_miss, p_fa) after every score in ascending order, preceded by the point
    (0, 1) of the threshold below all scores (roc.py:13-51).  Sort and running counts on the
    device."""
    torch = engine.require_gpu()
    is_tar, _ = _merged_order(true_scores, false_scores)
    num_true = int(is_tar.sum().item())
    num_false = is_tar.numel() - num_true
    assert num_true > 0 and num_false > 0
    # running counts on the device as integers; the two divisions on the host, where they round as
    # the reference's do
    sumtrue = torch.cumsum(is_tar.to(torch.int64), dim=0).cpu().numpy()
    sumfalse = num_false - (np.arange(1, sumtrue.size + 1) - sumtrue)
    p_miss = np.zeros(sumtrue.size + 1)
    p_fa = np.ones(sumtrue.size + 1)
    p_miss[1:] = sumtrue / num_true
    p_fa[1:] = sumfalse / num_false
    return p_miss, p_fa


def pavx(y):
    """Pool-adjacent-violators: the non-decreasing sequence closest to y in squared error, as
    (ghat [n], width of every constant block, its height) (utils.py:30-94).  A block joins its left
    neighbour while that neighbour is not lower; the joined height is updated as
    left + (width / new_width) (height - left), the reference's own arithmetic, so equal inputs give
    bit-equal outputs.  Sequential by nature: host code."""
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    assert y.size > 0
    heights, widths = [], []
    for v in y.tolist():
        w = 1
        while heights and heights[-1] >= v:
            left, lw = heights.pop(), widths.pop()
            v = left + (w / (lw + w)) * (v - left)
            w += lw
        heights.append(v)
        widths.append(w)
    width = np.array(widths, dtype=int)
    height = np.array(heights, dtype=np.float64)
    return np.repeat(height, width), width, height


def opt_loglr(tar, non, method="laplace"):
    """Non-parametric score to log-likelihood-ratio mapping (utils.py:97-150): isotonic regression of
    the target indicator over the sorted scores, with one target and one non-target added at each end
    for method "laplace"; returns (tar_llr, non_llr) in the callers' order.  The sort runs on the
    device, the pooling on the host."""
    ntar, nnon = np.size(tar), np.size(non)
    n = ntar + nnon
    is_tar, order = _merged_order(tar, non)
    p_ideal = is_tar.cpu().numpy().astype(np.float64)
    order = order.cpu().numpy()
    if method == "laplace":
        p_ideal = np.concatenate(([1, 0], p_ideal, [1, 0]))
    p_opt = pavx(p_ideal)[0]
    if method == "laplace":
        p_opt = p_opt[2:-2]
    with np.errstate(divide="ignore"):
        llr = np.log(p_opt) - np.log(1 - p_opt) - np.log(ntar / nnon)
    llr += 1e-6 * np.arange(n) / n
    out = np.empty(n, dtype=np.float64)
    out[order] = llr
    return out[:ntar], out[ntar:]
