"""Tables `key info1 [info2 ...]` such as utt2spk or enrolment lists (hyperion/utils/utt2info.py).
Columns are kept as numpy string arrays."""
from copy import deepcopy

import numpy as np

from .list_utils import ismember, split_list, split_list_group_by_key


class Utt2Info(object):
    """key: [n] record names; fields: [n, num_info_fields] strings."""

    def __init__(self, key, fields):
        self._key = np.asarray(key)
        fields = np.asarray(fields)
        self._fields = fields[:, None] if fields.ndim == 1 else fields
        assert self._fields.ndim == 2 and len(self._key) == len(self._fields) and self._fields.shape[1] >= 1
        self._index = None

    def validate(self):
        """A key column and at least one info column of the same length (utt2info.py:32-36)."""
        assert self._fields.ndim == 2 and len(self._key) == len(self._fields) and self._fields.shape[1] >= 1

    def __cmp__(self, other):
        """0 when equal, 1 otherwise (utt2info.py:282-286)."""
        return 0 if self.__eq__(other) else 1

    @classmethod
    def create(cls, key, info):
        return cls(key, info)

    @property
    def num_info_fields(self):
        return self._fields.shape[1]

    @property
    def key(self):
        return self._key

    @property
    def info(self):
        """[n] when the table has one info column, [n, fields] otherwise (utt2info.py:59-64)."""
        return self._fields[:, 0] if self._fields.shape[1] == 1 else self._fields

    def copy(self):
        return deepcopy(self)

    def __len__(self):
        return len(self._key)

    len = __len__

    def _positions(self):
        if self._index is None:
            self._index = {k: i for i, k in enumerate(self._key)}
        return self._index

    def get_index(self, key):
        return self._positions()[key]

    def __contains__(self, key):
        return key in self._positions()

    def __getitem__(self, key):
        """table['name'] -> info of that key; table[i] -> (key, info) (utt2info.py:96-124)."""
        if isinstance(key, str):
            row = self._fields[self.get_index(key)]
            return row[0] if len(row) == 1 else row
        row = self._fields[key]
        return self._key[key], (row[0] if len(row) == 1 else row)

    def _take(self, loc):
        return Utt2Info(self._key[loc], self._fields[loc])

    def sort(self, field=0):
        order = np.argsort(self._key if field == 0 else self._fields[:, field - 1], kind="stable")
        self._key, self._fields, self._index = self._key[order], self._fields[order], None

    def save(self, file_path, sep=" "):
        with open(file_path, "w") as f:
            for k, row in zip(self._key, self._fields):
                f.write(sep.join([str(k)] + [str(v) for v in row]) + "\n")

    @classmethod
    def load(cls, file_path, sep=" ", dtype=None):
        """Every line `key<sep>info...`; all columns are read as strings (utt2info.py:144-157 reads
        columns 0 and 1 as str by default)."""
        rows = []
        with open(file_path, "r") as f:
            for line in f:
                line = line.rstrip("\n").rstrip("\r")
                if line:
                    rows.append(line.split(sep))
        width = len(rows[0]) if rows else 2
        assert width >= 2 and all(len(r) == width for r in rows), "every line needs the same number of fields"
        table = np.asarray(rows, dtype=str).reshape(len(rows), width)
        return cls(table[:, 0], table[:, 1:])

    def split(self, idx, num_parts, group_by_field=0):
        if group_by_field == 0:
            _, loc = split_list(self._key, idx, num_parts)
        else:
            _, loc = split_list_group_by_key(self._fields[:, group_by_field - 1], idx, num_parts)
        return self._take(loc)

    @classmethod
    def merge(cls, info_lists):
        return cls(np.concatenate([u.key for u in info_lists]), np.concatenate([u._fields for u in info_lists]))

    def filter(self, filter_key, keep=True):
        """Rows with the given keys, in the order asked for (utt2info.py:195-209)."""
        if not keep:
            filter_key = np.setdiff1d(self._key, filter_key)
        return self._take(np.array([self.get_index(k) for k in filter_key], dtype=np.int64))

    def filter_info(self, filter_key, field=1, keep=True):
        """Rows whose info column `field` is in filter_key, in table order; every requested value
        must occur (utt2info.py:211-233)."""
        column = self._fields[:, field - 1]
        if not keep:
            filter_key = np.setdiff1d(column, filter_key)
        filter_key = np.asarray(filter_key)
        found, _ = ismember(filter_key, column)
        if not np.all(found):
            raise Exception("not all keys were found in field %d: %s" % (field, list(filter_key[~found][:5])))
        wanted, _ = ismember(column, filter_key)
        return self._take(np.flatnonzero(wanted))

    def filter_index(self, index, keep=True):
        if not keep:
            index = np.setdiff1d(np.arange(len(self), dtype=np.int64), index)
        return self._take(np.asarray(index, dtype=np.int64))

    def shuffle(self, seed=1024, rng=None):
        rng = np.random.RandomState(seed=seed) if rng is None else rng
        index = np.arange(len(self))
        rng.shuffle(index)
        self._key, self._fields, self._index = self._key[index], self._fields[index], None
        return index

    def __eq__(self, other):
        if len(self) == 0 and len(other) == 0:
            return True
        return (self._key.shape == other._key.shape and self._fields.shape == other._fields.shape and
                bool(np.all(self._key == other._key)) and bool(np.all(self._fields == other._fields)))

    def __ne__(self, other):
        return not self.__eq__(other)
