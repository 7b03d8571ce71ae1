"""Minimal HDF5 writer (no h5py): the file structures h5py's default settings produce for the files
the reference writes — trial index / key / score files (hyperion/utils/trial_ndx.py:61-72,
trial_key.py:98-126, trial_scores.py:71-83), model files (hyperion/hyp_model.py:44-75) and vector
files with one dataset per key (hyperion/io/h5_data_writer.py:92-130).

What is written: superblock version 0, old-style groups (symbol-table message, version 1 B-tree of
any depth, local heap, symbol-table nodes), version 1 object headers, contiguous datasets of
little-endian floats, integers and fixed-length (null-padded) strings, scalar or n-dimensional.
Not written: attributes, chunked / compressed / extendible datasets, variable-length types.

`File(path, "w")` offers the two calls the reference's save code makes on an h5py file:
`create_dataset(name, data=...)` (names with "/" create the groups on the way) and use as a context
manager.  Dataset bytes go to the file as they arrive; groups and the superblock follow at close.

Files are read back by `hdf5_lite.Hdf5File`, which is pinned on HDF5 files written by the HDF5
library itself; no HDF5 library is available to this package's tests, so that the library accepts
these files is by construction from the format specification, not a tested fact.
"""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
_LEAF_K = 4       # symbol-table nodes hold up to 2 K entries
_INTERNAL_K = 16  # B-tree nodes hold up to 2 K children
_SNOD_BYTES = 8 + 2 * _LEAF_K * 40
_TREE_BYTES = 24 + (2 * _INTERNAL_K + 1) * 8 + 2 * _INTERNAL_K * 8
_HEAP_FREE_NULL = 1  # "no free block" in a local heap header


def _msg(kind, data):
    data = bytes(data) + b"\0" * (-len(data) % 8)
    if len(data) > 0xFFFF:
        raise ValueError("header message too large")
    return struct.pack("<HHB3x", kind, len(data), 0) + data


def _datatype(dtype):
    """Datatype message (version 1) of a numpy dtype."""
    dtype = np.dtype(dtype)
    if dtype.kind == "S":
        # class 3 string, null-padded, ASCII
        return struct.pack("<B3BI", 0x13, 0x01, 0, 0, max(dtype.itemsize, 1))
    if dtype.kind == "f" and dtype.itemsize == 8:
        return struct.pack("<B3BIHHBBBBI", 0x11, 0x20, 0x3F, 0, 8, 0, 64, 52, 11, 0, 52, 1023)
    if dtype.kind == "f" and dtype.itemsize == 4:
        return struct.pack("<B3BIHHBBBBI", 0x11, 0x20, 0x1F, 0, 4, 0, 32, 23, 8, 0, 23, 127)
    if dtype.kind in "iu":
        return struct.pack("<B3BIHH", 0x10, 0x08 if dtype.kind == "i" else 0x00, 0, 0, dtype.itemsize, 0,
                           8 * dtype.itemsize)
    raise TypeError("no HDF5 datatype written for numpy dtype %s" % dtype)


def _storable(a):
    """The array as the little-endian, C-contiguous array that goes to the file."""
    a = np.asarray(a)
    if a.dtype.kind == "U":
        a = np.char.encode(a, "utf-8") if a.size else a.astype("S1")
    elif a.dtype.kind == "b":
        a = a.astype(np.uint8)
    elif a.dtype.kind == "f" and a.dtype.itemsize == 2:
        a = a.astype(np.float32)
    if a.dtype.kind == "S" and a.dtype.itemsize == 0:
        a = a.astype("S1")
    if a.dtype.kind not in "Sfiu":
        raise TypeError("cannot store numpy dtype %s" % a.dtype)
    return np.asarray(a.astype(a.dtype.newbyteorder("<"), copy=False), order="C")  # keeps 0-d arrays 0-d


class File(object):
    def __init__(self, path, mode="w"):
        if mode != "w":
            raise ValueError("hdf5_write.File only writes (mode 'w'); read with hdf5_lite.Hdf5File")
        self._f = open(path, "wb")
        self._f.write(b"\0" * 96)  # superblock + root symbol-table entry, filled in at close
        self._pos = 96
        self._root = {}

    # ---- context manager / h5py-like surface ---------------------------------------------------
    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        self.close()

    def flush(self):
        if self._f is not None:
            self._f.flush()

    def __contains__(self, name):
        node = self._root
        for p in name.strip("/").split("/"):
            if not isinstance(node, dict) or p not in node:
                return False
            node = node[p]
        return True

    def create_dataset(self, name, data=None, dtype=None, **unused):
        if self._f is None:
            raise ValueError("file is closed")
        if data is None:
            raise ValueError("create_dataset needs data (extendible datasets are not written)")
        a = _storable(np.asarray(data) if dtype is None else np.asarray(data).astype(dtype))
        parts = [p for p in name.split("/") if p]
        if not parts:
            raise ValueError("empty dataset name")
        node = self._root
        for p in parts[:-1]:
            node = node.setdefault(p, {})
            if not isinstance(node, dict):
                raise ValueError("%s: a dataset is in the way of group %s" % (name, p))
        if parts[-1] in node:
            raise ValueError("name already exists: %s" % name)
        node[parts[-1]] = self._dataset(a)

    def close(self):
        if self._f is None:
            return
        root_hdr, tree, heap = self._group(self._root)
        eof = self._pos
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, _LEAF_K, _INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        # root entry: cached symbol-table information (B-tree and heap) in the scratch pad
        sb += struct.pack("<QQIIQQ", 0, root_hdr, 1, 0, tree, heap)
        assert len(sb) == 96
        self._f.seek(0)
        self._f.write(sb)
        self._f.close()
        self._f = None

    # ---- file pieces ---------------------------------------------------------------------------
    def _put(self, data):
        pad = -self._pos % 8
        if pad:
            self._f.write(b"\0" * pad)
            self._pos += pad
        addr = self._pos
        self._f.write(data)
        self._pos += len(data)
        return addr

    def _header(self, msgs):
        body = b"".join(msgs)
        return self._put(struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body)

    def _dataset(self, a):
        nbytes = a.size * a.dtype.itemsize
        addr = self._put(a.reshape(-1).view(np.uint8)) if nbytes else UNDEF
        space = struct.pack("<BBB5x", 1, a.ndim, 0) + b"".join(struct.pack("<Q", d) for d in a.shape)
        fill = struct.pack("<BBBBI", 2, 2, 2, 1, 0)  # version 2, late allocation, written if set, default value
        layout = struct.pack("<BBQQ", 3, 1, addr, nbytes)
        return self._header([_msg(0x01, space), _msg(0x03, _datatype(a.dtype)), _msg(0x05, fill),
                             _msg(0x08, layout)])

    def _group(self, members):
        """Writes the members' groups first, then this group -> (object header, B-tree, heap)."""
        addrs = {}
        for n, v in members.items():
            addrs[n.encode("utf-8")] = self._group(v)[0] if isinstance(v, dict) else v
        names = sorted(addrs)
        heap = bytearray(8)  # offset 0: the empty name
        offs = {}
        for n in names:
            offs[n] = len(heap)
            heap += n + b"\0" * (8 - len(n) % 8)
        heap_data = self._put(bytes(heap))
        heap_addr = self._put(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), _HEAP_FREE_NULL, heap_data))
        # leaves: symbol-table nodes of up to 2 K entries; (address, offset of the largest name in it)
        level_nodes = []
        for i in range(0, len(names), 2 * _LEAF_K):
            part = names[i:i + 2 * _LEAF_K]
            snod = b"SNOD" + struct.pack("<BBH", 1, 0, len(part))
            snod += b"".join(struct.pack("<QQII16x", offs[n], addrs[n], 0, 0) for n in part)
            level_nodes.append((self._put(snod + b"\0" * (_SNOD_BYTES - len(snod))), offs[part[-1]]))
        level = 0
        while True:
            groups = [level_nodes[i:i + 2 * _INTERNAL_K] for i in range(0, len(level_nodes), 2 * _INTERNAL_K)] or [[]]
            base = self._pos + (-self._pos % 8)  # nodes of one level are consecutive: siblings are known
            out = []
            for gi, g in enumerate(groups):
                left = base + (gi - 1) * _TREE_BYTES if gi > 0 else UNDEF
                right = base + (gi + 1) * _TREE_BYTES if gi + 1 < len(groups) else UNDEF
                node = b"TREE" + struct.pack("<BBHQQ", 0, level, len(g), left, right)
                # key 0 of the leftmost node is the empty name; otherwise the largest name to the left
                first_key = 0 if gi == 0 else groups[gi - 1][-1][1]
                node += struct.pack("<Q", first_key)
                node += b"".join(struct.pack("<QQ", child, last) for child, last in g)
                addr = self._put(node + b"\0" * (_TREE_BYTES - len(node)))
                assert addr == base + gi * _TREE_BYTES
                out.append((addr, g[-1][1] if g else 0))
            if len(out) == 1:
                tree_addr = out[0][0]
                break
            level_nodes = out
            level += 1
        hdr = self._header([_msg(0x11, struct.pack("<QQ", tree_addr, heap_addr))])
        return hdr, tree_addr, heap_addr
