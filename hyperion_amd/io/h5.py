"""HDF5 vector / matrix files: one dataset per key (hyperion/io/h5_data_reader.py:403-760,
h5_data_writer.py), read and written through the package's own HDF5 parser and writer (hdf5_lite,
hdf5_write; h5py is not needed).  Keys with "/" are nested groups, as h5py stores them.  Datasets
with the Kaldi compression attributes ("data_format") are not supported."""
import os

import numpy as np

from ..hyp_defs import float_cpu, float_save
from .ark import RandomAccessArkDataReader
from .hdf5_lite import Hdf5File


def _rows(x, row_offset, num_rows):
    """hyperion/io/h5_data_reader.py:37-46."""
    if row_offset > 0:
        x = x[row_offset:] if num_rows == 0 else x[row_offset:row_offset + num_rows]
    elif num_rows > 0:
        x = x[:num_rows]
    assert num_rows == 0 or x.shape[0] == num_rows
    return x


class _H5ReaderBase(object):
    def __init__(self, transform=None, permissive=False):
        self.transform = transform
        self.permissive = permissive

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        pass

    def _dataset(self, key):
        """-> (Hdf5File, name inside it) or None for a missing key."""
        raise NotImplementedError

    def _get(self, key):
        hit = self._dataset(key)
        if hit is None:
            if self.permissive:
                return None
            raise Exception("Key %s not found" % key)
        return hit

    def read_shapes(self, keys, assert_same_dim=True):
        if isinstance(keys, str):
            keys = [keys]
        shapes = []
        for key in keys:
            hit = self._get(key)
            shapes.append((0,) if hit is None else tuple(hit[0].shape(hit[1])))
        if assert_same_dim and shapes:
            dims = {s[-1] for s in shapes if s != (0,)}
            assert len(dims) <= 1, "records of different dimension: %s" % sorted(dims)
        return shapes

    def read_num_rows(self, keys, assert_same_dim=True):
        shapes = self.read_shapes(keys, assert_same_dim)
        return np.array([s[0] if len(s) == 2 else 1 for s in shapes], dtype=int)

    def read_dims(self, keys, assert_same_dim=True):
        return np.array([s[-1] for s in self.read_shapes(keys, assert_same_dim)], dtype=np.int32)

    def read(self, keys, squeeze=False, row_offset=0, num_rows=0):
        """Records of `keys` as float_cpu() arrays: a list, or with squeeze one stacked array
        (h5_data_reader.py:520-574)."""
        if isinstance(keys, str):
            keys = [keys]
        per_key = lambda v, i: v[i] if isinstance(v, (list, np.ndarray)) else v  # noqa: E731
        for v in (row_offset, num_rows):
            assert not isinstance(v, (list, np.ndarray)) or len(v) == len(keys)
        data = []
        for i, key in enumerate(keys):
            hit = self._get(key)
            if hit is None:
                data.append(np.array([], dtype=float_cpu()))
                continue
            x = _rows(hit[0][hit[1]], per_key(row_offset, i), per_key(num_rows, i))
            x = np.asarray(x, dtype=float_cpu())
            if self.transform is not None:
                x = self.transform.predict(x)
            data.append(x)
        return RandomAccessArkDataReader._squeeze(data, self.permissive) if squeeze else data

    def read_device(self, keys):
        """Vectors of `keys` as a device float32 tensor [n, dim] (one transfer from page-locked memory)."""
        from .. import engine

        torch = engine.require_gpu()
        x = self.read(keys, squeeze=True)
        assert x.ndim == 2, "read_device handles vector records"
        ld = engine.round_up(x.shape[1], 4)
        host = torch.zeros((x.shape[0], ld), dtype=torch.float32).pin_memory()
        host[:, :x.shape[1]] = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        return host.to("cuda", non_blocking=True)[:, :x.shape[1]]


class RandomAccessH5FileDataReader(_H5ReaderBase):
    """All keys in one HDF5 file (h5_data_reader.py:454-574)."""

    def __init__(self, file_path, transform=None, permissive=False, **kwargs):
        super().__init__(transform, permissive)
        self.file_path = file_path
        self.f = Hdf5File(file_path)

    def keys(self):
        return self.f.keys()

    def _dataset(self, key):
        return (self.f, key) if key in self.f else None


class RandomAccessH5ScriptDataReader(_H5ReaderBase):
    """Script lines `key file.h5[first:last]`: every key is a dataset of that name in the file of
    its line (h5_data_reader.py:576-760)."""

    def __init__(self, file_path, path_prefix=None, scp_sep=" ", transform=None, permissive=False, **kwargs):
        super().__init__(transform, permissive)
        from ..utils.scp_list import SCPList

        self.scp = SCPList.load(file_path, sep=scp_sep)
        if path_prefix is not None:
            self.scp.add_prefix_to_filepath(path_prefix)
        self._where = {k: i for i, k in enumerate(self.scp.key)}
        self._files = {}

    def keys(self):
        return self.scp.key

    def _dataset(self, key):
        i = self._where.get(key)
        if i is None:
            return None
        path = self.scp.file_path[i]
        if path not in self._files:
            self._files[path] = Hdf5File(path)
        f = self._files[path]
        return (f, key) if key in f else None

    def read(self, keys, squeeze=False, row_offset=0, num_rows=0):
        if isinstance(keys, str):
            keys = [keys]
        if self.scp.range_spec is None:
            return super().read(keys, squeeze, row_offset, num_rows)
        from .ark import _combine_ranges

        per_key = lambda v, i: v[i] if isinstance(v, (list, np.ndarray)) else v  # noqa: E731
        first, count = [], []
        for i, key in enumerate(keys):
            p = self._where.get(key)
            a, b = _combine_ranges(None if p is None else self.scp.range_spec[p], per_key(row_offset, i),
                                   per_key(num_rows, i))
            first.append(a)
            count.append(b)
        return super().read(keys, squeeze, first, count)


class H5DataWriter(object):
    """One float_save() dataset per key and, optionally, a script line `key file.h5` for each
    (hyperion/io/h5_data_writer.py:17-130); uncompressed only."""

    def __init__(self, archive_path, script_path=None, flush=False, compress=False, compression_method="auto",
                 scp_sep=" ", **kwargs):
        if compress:
            raise NotImplementedError("Kaldi-compressed HDF5 datasets are not written")
        from .hdf5_write import File

        self.archive_path, self.script_path, self.scp_sep, self._flush = archive_path, script_path, scp_sep, flush
        for p in (archive_path, script_path):
            d = os.path.dirname(p) if p else ""
            if d and not os.path.exists(d):
                os.makedirs(d)
        self.f = File(archive_path, "w")
        self.f_script = open(script_path, "w") if script_path is not None else None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self.f is not None:
            self.f.close()
            self.f = None
        if self.f_script is not None:
            self.f_script.close()
            self.f_script = None

    def flush(self):
        self.f.flush()
        if self.f_script is not None:
            self.f_script.flush()

    def write(self, keys, data):
        """keys: names (no white space); data: list of vectors / matrices, or a 2-D / 3-D array
        whose first axis runs over the keys."""
        if isinstance(keys, str):
            keys, data = [keys], [data]
        for key, x in zip(keys, data):
            assert len(key) > 0 and not any(c.isspace() for c in key), "Token %s not valid" % key
            if not isinstance(x, np.ndarray):
                raise ValueError("Data is not ndarray")
            self.f.create_dataset(key, data=x.astype(float_save(), copy=False))
            if self.f_script is not None:
                self.f_script.write("%s%s%s\n" % (key, self.scp_sep, self.archive_path))
        if self._flush:
            self.flush()
