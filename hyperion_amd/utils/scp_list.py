"""Script lists: `key path[:offset][[first:last]]` lines that index archives
(hyperion/utils/scp_list.py)."""
from copy import deepcopy

import numpy as np

from .list_utils import list2ndarray, sort, split_list, split_list_group_by_key


class SCPList(object):
    """key: record names; file_path: archive of each record; offset: byte position in the archive
    (None for plain file lists); range_spec: [first_row, num_rows] per record (num_rows 0 = to the
    end) or None."""

    def __init__(self, key, file_path, offset=None, range_spec=None):
        self.key = list2ndarray(key)
        self.file_path = list2ndarray(file_path, dtype=object)
        assert len(self.key) == len(self.file_path)
        if offset is not None:
            offset = np.asarray(offset, dtype=np.int64)
            assert len(offset) == len(self.key)
        if range_spec is not None:
            if len(range_spec) == 0:
                range_spec = None
            else:
                range_spec = np.asarray(range_spec, dtype=np.int64)
                assert range_spec.shape == (len(self.key), 2)
        self.offset = offset
        self.range_spec = range_spec
        self._index = None

    def validate(self):
        """Re-checks attributes that were assigned after construction (scp_list.py:36-64): array
        types, equal lengths, an empty range list read as none."""
        self.__init__(self.key, self.file_path, self.offset, self.range_spec)

    def __cmp__(self, other):
        """0 when equal, 1 otherwise (scp_list.py:436-440; a Python 2 protocol nothing calls)."""
        return 0 if self.__eq__(other) else 1

    def copy(self):
        return deepcopy(self)

    def __len__(self):
        return len(self.key)

    len = __len__

    def _positions(self):
        if self._index is None:
            self._index = {k: i for i, k in enumerate(self.key)}
        return self._index

    def get_index(self, key):
        return self._positions()[key]

    def __contains__(self, key):
        return key in self._positions()

    def __getitem__(self, key):
        """scp['name'] -> (file_path, offset, range_spec); scp[i] -> (key, file_path, offset,
        range_spec) (scp_list.py:85-112)."""
        by_name = isinstance(key, str)
        i = self.get_index(key) if by_name else key
        rec = (self.file_path[i], None if self.offset is None else self.offset[i],
               None if self.range_spec is None else self.range_spec[i])
        return rec if by_name else (self.key[i],) + rec

    def add_prefix_to_filepath(self, prefix):
        self.file_path = np.array([prefix + p for p in self.file_path], dtype=object)

    def sort(self):
        self.key, order = sort(self.key, return_index=True)
        self._take(order)

    def _take(self, loc):
        self.file_path = self.file_path[loc]
        if self.offset is not None:
            self.offset = self.offset[loc]
        if self.range_spec is not None:
            self.range_spec = self.range_spec[loc]
        self._index = None

    @staticmethod
    def _range_text(r):
        if r[0] == 0 and r[1] == 0:
            return ""
        return "[%d:]" % r[0] if r[1] == 0 else "[%d:%d]" % (r[0], r[0] + r[1] - 1)

    def save(self, file_path, sep=" ", offset_sep=":"):
        """One line per record (scp_list.py:128-157)."""
        with open(file_path, "w") as f:
            for i, (k, p) in enumerate(zip(self.key, self.file_path)):
                pos = "" if self.offset is None else "%s%d" % (offset_sep, self.offset[i])
                rng = "" if self.range_spec is None else self._range_text(self.range_spec[i])
                f.write("%s%s%s%s%s\n" % (k, sep, p, pos, rng))

    @staticmethod
    def parse_script(script, offset_sep):
        """Second column of an scp file -> (file paths, offsets or None, range_spec or None)
        (scp_list.py:159-201): offsets are expected on every line when the first line has one; a
        range `[a:b]` is inclusive, `[a:]` runs to the end."""
        paths, offsets, ranges = [], [], []
        any_range = False
        with_offset = None
        for s in script:
            body, _, rng = s.partition("[")
            path, has_off, off = body.partition(offset_sep)
            if with_offset is None:
                with_offset = bool(has_off)
            if with_offset:
                if not has_off:
                    raise ValueError("Missing data position for %s" % path)
                offsets.append(int(off))
            paths.append(path)
            first, count = 0, 0
            if rng:
                any_range = True
                a, _, b = rng.rstrip("]").partition(":")
                first = int(a) if a.isdecimal() else 0
                count = int(b) - first + 1 if b.isdecimal() else 0
            ranges.append((first, count))
        return (paths, offsets if with_offset else None,
                np.asarray(ranges, dtype=np.int64) if any_range else None)

    @classmethod
    def load(cls, file_path, sep=" ", offset_sep=":", is_wav=False):
        """Reads `key<sep>script` lines (scp_list.py:203-229)."""
        key, script = [], []
        with open(file_path, "r") as f:
            for line in f:
                k, s = line.rstrip().split(sep=sep, maxsplit=1)
                key.append(k)
                script.append(s)
        if is_wav:
            return cls(key, script)
        return cls(key, *cls.parse_script(script, offset_sep))

    def split(self, idx, num_parts, group_by_key=True):
        """Part `idx` of `num_parts` (scp_list.py:231-259)."""
        picker = split_list_group_by_key if group_by_key else split_list
        return self._subset(picker(self.key, idx, num_parts)[1])

    def _subset(self, loc):
        return SCPList(self.key[loc], self.file_path[loc], None if self.offset is None else self.offset[loc],
                       None if self.range_spec is None else self.range_spec[loc])

    @classmethod
    def merge(cls, scp_lists):
        """Concatenation of several lists (scp_list.py:261-289)."""
        cat = lambda name: (None if getattr(scp_lists[0], name) is None  # noqa: E731
                            else np.concatenate([getattr(s, name) for s in scp_lists]))
        return cls(cat("key"), cat("file_path"), cat("offset"), cat("range_spec"))

    def filter(self, filter_key, keep=True):
        """All records whose key is in filter_key (or is not, keep=False), in list order; every
        requested key must exist (scp_list.py:291-317)."""
        return self._filter_by(self.key, filter_key, keep)

    def filter_paths(self, filter_key, keep=True):
        """As filter, by archive path (scp_list.py:319-346)."""
        return self._filter_by(self.file_path, filter_key, keep)

    def _filter_by(self, column, wanted, keep):
        if not keep:
            wanted = np.setdiff1d(column, wanted)
        have = set(column.tolist())
        assert all(w in have for w in wanted), "a requested entry is not in the list"
        wanted = set(np.asarray(wanted).tolist())
        return self._subset(np.fromiter((c in wanted for c in column.tolist()), dtype=bool, count=len(column)))

    def filter_index(self, index, keep=True):
        """Records at the given positions, in that order (scp_list.py:348-374)."""
        if not keep:
            index = np.setdiff1d(np.arange(len(self.key), dtype=np.int64), index)
        return self._subset(np.asarray(index, dtype=np.int64))

    def shuffle(self, seed=1024, rng=None):
        """Random order in place; returns the permutation (scp_list.py:376-399)."""
        rng = np.random.RandomState(seed=seed) if rng is None else rng
        index = np.arange(len(self.key))
        rng.shuffle(index)
        self.key = self.key[index]
        self._take(index)
        return index

    def __eq__(self, other):
        if len(self.key) == 0 and len(other.key) == 0:  # empty lists are equal whatever they carry
            return True
        same = lambda a, b: (a is None) == (b is None) and (a is None or (a.shape == b.shape and np.all(a == b)))  # noqa: E731
        return (same(self.key, other.key) and same(self.file_path, other.file_path) and
                same(self.offset, other.offset) and same(self.range_spec, other.range_spec))

    def __ne__(self, other):
        return not self.__eq__(other)
