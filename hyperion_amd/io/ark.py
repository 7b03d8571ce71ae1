"""Kaldi binary archives of vectors and matrices, indexed by scp files
(hyperion/io/ark_data_reader.py, ark_data_writer.py, hyperion/utils/kaldi_matrix.py).

Reading the x-vectors of a trial list is the step in front of the scoring path: vectors asked for
with squeeze=True go through hypmi_ark_read_vectors (threaded pread straight into one buffer,
page-locked when a HIP device is present, so that the device copy is a single transfer).  Matrices,
row ranges and un-squeezed reads are parsed record by record on the host; compressed matrices (feature
archives) are decoded and encoded by kaldi_compressed.py.  Text archives are not supported."""
import ctypes
import os
import struct

import numpy as np

from .. import _lib
from ..hyp_defs import float_cpu, float_save
from . import kaldi_compressed
from ..utils.scp_list import SCPList

_MARK = b"\0B"


def _combine_ranges(read_range, row_offset, num_rows):
    """scp range [first, count] (count 0 = to the end) combined with a caller's row_offset /
    num_rows inside it (hyperion/io/data_reader.py:85-128)."""
    if read_range is None:
        return row_offset, num_rows
    first, count = int(read_range[0]), int(read_range[1])
    if count == 0:
        return first + row_offset, num_rows
    if num_rows == 0:
        num_rows = count - row_offset
    assert row_offset + num_rows <= count, "rows %d..%d outside the range of %d rows in the script" % (
        row_offset, row_offset + num_rows, count)
    return first + row_offset, num_rows


def _read_header(f):
    """(dtype, is_matrix, rows, cols) of the plain record at the file position, or
    ("compressed", format) with the position behind the token of a compressed matrix
    (kaldi_matrix.py:63-105, 196-230, 714-727)."""
    if f.read(2) != _MARK:
        raise ValueError("not a binary Kaldi record (text archives are not supported)")
    token = kaldi_compressed.read_token(f)
    if token[:1] == b"C":
        if token not in kaldi_compressed.TOKENS:
            raise ValueError("Unexpected token %s" % token.decode("ascii", "replace"))
        return "compressed", kaldi_compressed.TOKENS[token]
    if len(token) != 2 or token[:1] not in (b"F", b"D") or token[1:2] not in (b"V", b"M"):
        raise ValueError("unknown record token %r" % token)
    dtype = np.dtype("<f4") if token[:1] == b"F" else np.dtype("<f8")
    is_matrix = token[1:2] == b"M"
    dims = []
    for _ in range(2 if is_matrix else 1):
        size, value = struct.unpack("<bi", f.read(5))
        assert size == 4
        dims.append(value)
    rows, cols = (dims[0], dims[1]) if is_matrix else (1, dims[0])
    return dtype, is_matrix, rows, cols


def _read_record(f, row_offset=0, num_rows=0):
    head = _read_header(f)
    if head[0] == "compressed":
        return kaldi_compressed.read(f, head[1], row_offset, num_rows)
    dtype, is_matrix, rows, cols = head
    if not is_matrix:
        return np.frombuffer(f.read(cols * dtype.itemsize), dtype=dtype)
    assert row_offset <= rows, "row_offset (%d) > num_rows (%d)" % (row_offset, rows)
    avail = rows - row_offset
    if num_rows == 0:
        num_rows = avail
    assert num_rows <= avail, "requested rows (%d) > available rows (%d)" % (num_rows, avail)
    f.seek(row_offset * cols * dtype.itemsize, 1)
    return np.frombuffer(f.read(num_rows * cols * dtype.itemsize), dtype=dtype).reshape(num_rows, cols)


class RandomAccessArkDataReader(object):
    """file_path: scp file; path_prefix: prepended to every archive path; transform: object with
    .predict applied to every record; permissive: unknown keys give empty records instead of an
    exception (ark_data_reader.py:403-436)."""

    def __init__(self, file_path, path_prefix=None, transform=None, permissive=False, scp_sep=" ",
                 num_threads=None):
        self.file_path = file_path
        self.transform = transform
        self.permissive = permissive
        self.scp = SCPList.load(file_path, sep=scp_sep)
        if path_prefix is not None:
            self.scp.add_prefix_to_filepath(path_prefix)
        assert self.scp.offset is not None, "random access to an archive needs byte offsets in the script"
        self.archives, self.archive_idx = np.unique(self.scp.file_path.astype(str), return_inverse=True)
        self.archive_idx = np.asarray(self.archive_idx, dtype=np.int32).reshape(-1)
        self.num_threads = int(num_threads or min(16, os.cpu_count() or 1))
        self._files = {}

    @property
    def keys(self):
        return self.scp.key

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        for f in self._files.values():
            f.close()
        self._files = {}

    def _file(self, a):
        if a not in self._files:
            self._files[a] = open(self.archives[a], "rb")
        return self._files[a]

    def _lookup(self, keys):
        """Positions of the keys in the script; -1 for unknown keys when permissive."""
        if isinstance(keys, str):
            keys = [keys]
        index = self.scp._positions()
        pos = np.fromiter((index.get(k, -1) for k in keys), dtype=np.int64, count=len(keys))
        if not self.permissive and len(pos) and pos.min() < 0:
            raise Exception("Key %s not found" % keys[int(np.argmin(pos))])
        return keys, pos

    # ---- shapes -------------------------------------------------------------------------------
    def read_shapes(self, keys, assert_same_dim=True):
        """Shape of every record: (dim,) for vectors, (rows, cols) for matrices, (0,) for unknown
        keys when permissive (ark_data_reader.py:508-552)."""
        keys, pos = self._lookup(keys)
        shapes = []
        for p in pos:
            if p < 0:
                shapes.append((0,))
                continue
            f = self._file(self.archive_idx[p])
            f.seek(self.scp.offset[p], 0)
            head = _read_header(f)
            if head[0] == "compressed":
                is_matrix, (rows, cols) = True, kaldi_compressed.read_shape(f, head[1], skip=False)
            else:
                _, is_matrix, rows, cols = head
            if is_matrix:
                first, count = _combine_ranges(None if self.scp.range_spec is None else self.scp.range_spec[p], 0, 0)
                rows = count if count > 0 else rows - first
                shapes.append((rows, cols))
            else:
                shapes.append((cols,))
        if assert_same_dim and shapes:
            assert len({s[-1] for s in shapes}) == 1
        return shapes

    def read_num_rows(self, keys, assert_same_dim=True):
        return np.array([s[0] if len(s) == 2 else 1 for s in self.read_shapes(keys, assert_same_dim)], dtype=int)

    def read_dims(self, keys, assert_same_dim=True):
        dims = np.array([s[-1] for s in self.read_shapes(keys, False)], dtype=int)
        if assert_same_dim:
            assert np.all(dims == dims[0])
        return dims

    # ---- data ---------------------------------------------------------------------------------
    def _vector_fast_path(self, pos, row_offset, num_rows):
        """Record width in bytes and dimension when every key is known, no ranges are involved and
        the first record is a vector (the native loader then checks every record); else None."""
        if len(pos) == 0 or np.any(pos < 0) or self.scp.range_spec is not None:
            return None
        if not (np.isscalar(row_offset) and row_offset == 0 and np.isscalar(num_rows) and num_rows == 0):
            return None
        f = self._file(self.archive_idx[pos[0]])
        f.seek(self.scp.offset[pos[0]], 0)
        if f.read(3)[2:3] == b"C":  # a compressed matrix: decoded record by record
            return None
        lib = _lib.load()
        rows, cols, width, is_mat = (ctypes.c_int32() for _ in range(4))
        _lib.check(lib.hypmi_ark_probe(self.archives[self.archive_idx[pos[0]]].encode(), int(self.scp.offset[pos[0]]),
                                       ctypes.byref(rows), ctypes.byref(cols), ctypes.byref(width),
                                       ctypes.byref(is_mat)))
        if is_mat.value:
            return None
        return width.value, cols.value

    def read_into(self, keys, out):
        """Vectors of `keys` into the rows of `out`, a C-contiguous-row float32 or float64 numpy
        array [len(keys), >= dim] (e.g. a view of page-locked memory); all keys must exist."""
        return self._read_positions(self._lookup(keys)[1], out)

    def _read_positions(self, pos, out):
        assert not np.any(pos < 0), "read_into needs every key"
        assert out.ndim == 2 and out.shape[0] == len(pos) and out.strides[1] == out.itemsize
        assert out.dtype in (np.float32, np.float64) and out.strides[0] % out.itemsize == 0
        names = (ctypes.c_char_p * len(self.archives))(*[a.encode() for a in self.archives])
        # records are handed over in file order (neighbours share a read) with their output rows
        rows = np.lexsort((self.scp.offset[pos], self.archive_idx[pos])).astype(np.int64)
        which = np.ascontiguousarray(self.archive_idx[pos][rows], dtype=np.int32)
        where = np.ascontiguousarray(self.scp.offset[pos][rows], dtype=np.int64)
        _lib.check(_lib.load().hypmi_ark_read_vectors(
            ctypes.cast(names, ctypes.c_void_p), len(self.archives), which.ctypes.data, where.ctypes.data,
            rows.ctypes.data, len(pos), out.shape[1], int(out.dtype == np.float64), out.ctypes.data,
            out.strides[0] // out.itemsize, self.num_threads))
        return out

    def read(self, keys, squeeze=False, row_offset=0, num_rows=0):
        """Records of `keys`: a list of arrays, or with squeeze one stacked array
        (ark_data_reader.py:554-622).  Stored precision is kept (float32 or float64)."""
        keys, pos = self._lookup(keys)
        if squeeze and self.transform is None:
            fast = self._vector_fast_path(pos, row_offset, num_rows)
            if fast is not None:
                out = np.empty((len(keys), fast[1]), dtype=np.float32 if fast[0] == 4 else np.float64)
                return self._read_positions(pos, out)
        per_key = lambda v, i: v[i] if isinstance(v, (list, np.ndarray)) else v  # noqa: E731
        data = []
        for i, p in enumerate(pos):
            if p < 0:
                data.append(np.array([], dtype=float_cpu()))
                continue
            first, count = _combine_ranges(None if self.scp.range_spec is None else self.scp.range_spec[p],
                                           per_key(row_offset, i), per_key(num_rows, i))
            f = self._file(self.archive_idx[p])
            f.seek(self.scp.offset[p], 0)
            x = _read_record(f, first, count)
            assert count == 0 or x.shape[0] == count
            if self.transform is not None:
                x = self.transform.predict(x)
            data.append(x)
        return self._squeeze(data, self.permissive) if squeeze else data

    @staticmethod
    def _squeeze(data, permissive=False):
        """List of equally shaped records -> one array with a leading record axis; empty records
        become zeros when permissive (data_reader.py:59-83)."""
        shape = data[0].shape
        rows = []
        for x in data:
            if len(x) == 0:
                if permissive:
                    rows.append(np.zeros((1,) + shape, dtype=float_cpu()))
                continue
            assert x.ndim == len(shape) and x.shape[-1] == shape[-1]
            rows.append(x[None])
        return np.concatenate(rows, axis=0)

    def read_device(self, keys):
        """Vectors of `keys` as a device float32 tensor [n, dim]: read into page-locked memory and
        copied with one asynchronous transfer on the current stream."""
        from .. import engine

        torch = engine.require_gpu()
        keys, pos = self._lookup(keys)
        fast = self._vector_fast_path(pos, 0, 0)
        assert fast is not None, "read_device handles plain vector records of known keys"
        ld = engine.round_up(fast[1], 4)
        host = torch.zeros((len(keys), ld), dtype=torch.float32).pin_memory()
        self._read_positions(pos, host.numpy()[:, :fast[1]])
        return host.to("cuda", non_blocking=True)[:, :fast[1]]


class ArkDataWriter(object):
    """Writes records `key \\0B<token><dims><data>` and, optionally, the scp line of each
    (ark_data_writer.py:15-120; kaldi_matrix.py:163-178).  Arrays are stored as float_save()
    (float32) like the reference does; with compress, matrices are stored in one of Kaldi's compressed
    formats (compression_method: see kaldi_compressed.METHOD_FORMAT, or "auto").  Binary only."""

    def __init__(self, archive_path, script_path=None, binary=True, flush=False, compress=False,
                 compression_method="auto", scp_sep=" "):
        if not binary:
            raise NotImplementedError("only binary archives are written")
        self.compress, self.compression_method = bool(compress), compression_method
        self.archive_path, self.script_path, self.scp_sep, self._flush = archive_path, script_path, scp_sep, flush
        for p in (archive_path, script_path):
            d = os.path.dirname(p) if p else ""
            if d and not os.path.exists(d):
                os.makedirs(d)
        self.f = open(archive_path, "wb")
        self.f_script = open(script_path, "w") if script_path is not None else None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        self.f.close()
        if self.f_script is not None:
            self.f_script.close()

    def flush(self):
        self.f.flush()
        if self.f_script is not None:
            self.f_script.flush()

    def write(self, keys, data):
        """keys: names (no white space); data: list of vectors / matrices, or a 2-D / 3-D array
        whose first axis runs over the keys."""
        if isinstance(keys, str):
            keys, data = [keys], [data]
        pos = self.f.tell()
        chunks, lines = [], []
        for key, x in zip(keys, data):
            assert len(key) > 0 and not any(c.isspace() for c in key), "Token %s not valid" % key
            x = np.ascontiguousarray(x, dtype=float_save())
            assert x.ndim in (1, 2)
            head = key.encode() + b" "
            if self.compress:
                body = _MARK + kaldi_compressed.compress(x, self.compression_method)
            else:
                token = (b"F" if x.dtype == np.float32 else b"D") + (b"V " if x.ndim == 1 else b"M ")
                body = _MARK + token + b"".join(struct.pack("<bi", 4, n) for n in x.shape) + x.tobytes()
            lines.append("%s%s%s:%d\n" % (key, self.scp_sep, self.archive_path, pos + len(head)))
            chunks += [head, body]
            pos += len(head) + len(body)
        self.f.write(b"".join(chunks))
        if self.f_script is not None:
            self.f_script.writelines(lines)
        if self._flush:
            self.flush()
