"""Helpers on lists of string ids (same results as hyperion/utils/list_utils.py, built on hashing
and array operations instead of per-element Python loops so that million-entry lists stay cheap)."""
import numpy as np


def list2ndarray(a, dtype=None):
    """A Python list becomes an ndarray; an ndarray passes through (list_utils.py:13-18)."""
    if isinstance(a, np.ndarray):
        return a
    assert isinstance(a, (list, tuple))
    return np.asarray(a, dtype=dtype)


def ismember(a, b):
    """For every element of `a`: is it in `b`, and the lowest index in `b` where it occurs
    (list_utils.py:21-40; int32 positions, the int32 minimum where absent)."""
    absent = np.iinfo(np.int32).min
    first = {}
    for i, v in enumerate(b):
        first.setdefault(v, i)
    loc = np.fromiter((first.get(v, absent) for v in a), dtype=np.int32, count=len(a))
    return loc != absent, loc


def sort(a, reverse=False, return_index=False):
    """Sorted copy, optionally with the permutation that sorts (list_utils.py:43-66)."""
    a = np.asarray(a)
    order = np.argsort(a)
    if reverse:
        order = order[::-1]
    return (a[order], order) if return_index else a[order]


def intersect(a, b, assume_unique=False, return_index=False):
    """Sorted common elements, optionally with their positions in `a` and in `b`
    (list_utils.py:69-92)."""
    c = np.intersect1d(a, b, assume_unique)
    if not return_index:
        return c
    return c, ismember(c, a)[1], ismember(c, b)[1]


def split_bounds(n, idx, num_parts):
    """Half-open range of part `idx` (1-based) of `num_parts` over n items: the contiguous floor
    partition of list_utils.py:108-111."""
    n = float(n)
    return int(np.floor((idx - 1) * n / num_parts)), int(np.floor(idx * n / num_parts))


def split_list(a, idx, num_parts):
    """Part `idx` of `num_parts` of a list and the positions it covers (list_utils.py:95-112)."""
    a = np.asarray(a)
    lo, hi = split_bounds(len(a), idx, num_parts)
    loc = np.arange(lo, hi, dtype=np.int64)
    return a[loc], loc


def split_list_group_by_key(a, idx, num_parts, key=None):
    """As split_list, but the parts are made over the distinct (sorted) values of `key` and every
    element follows its key (list_utils.py:115-144): elements come out grouped by key value."""
    a = np.asarray(a)
    ids = np.unique(a if key is None else key, return_inverse=True)[1]
    lo, hi = split_bounds(int(ids.max()) + 1, idx, num_parts)
    order = np.argsort(ids, kind="stable")
    sorted_ids = ids[order]
    loc = order[np.searchsorted(sorted_ids, lo, "left"):np.searchsorted(sorted_ids, hi, "left")].astype(np.int64)
    return a[loc], loc
