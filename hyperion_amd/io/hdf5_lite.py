"""Read-only HDF5 for the files the back-end meets: trial lists and keys (hyperion/utils/trial_ndx.py:
110-124, trial_key.py load_h5), vector files written one dataset per key (hyperion/io/h5_data_writer.py)
and model parameter files.  h5py is not a dependency of this package; this module parses the subset
of the format that h5py's defaults produce for such files:

  superblock version 0 - 3, old-style groups (symbol-table B-tree + local heap) and new-style compact
  link messages, version 1 / 2 object headers, contiguous, compact and chunked (version 1 B-tree)
  layouts, deflate and shuffle filters, integer / IEEE float / fixed-length string / enum (h5py's
  bool) datatypes.  Variable-length strings are read when they sit in the global heap.

Anything else raises Hdf5Error naming the feature.  Nothing in a file is executed.
"""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(ValueError):
    pass


class _Dataset:
    def __init__(self, f, msgs):
        self._f = f
        self.shape = ()
        self.dtype = None
        self._vlen_str = False
        self._layout = None
        self._filters = []
        for t, data in msgs:
            if t == 0x01:
                self.shape = _dataspace(data)
            elif t == 0x03:
                self.dtype, self._vlen_str = _datatype(data)
            elif t == 0x08:
                self._layout = data
            elif t == 0x0B:
                self._filters = _filters(data)
        if self._layout is None or self.dtype is None:
            raise Hdf5Error("object is not a dataset")

    def read(self):
        f = self._f
        lay = self._layout
        ver, cls = lay[0], lay[1]
        if ver != 3:
            raise Hdf5Error("data layout message version %d" % ver)
        n = int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1
        item = self.dtype.itemsize
        if cls == 0:  # compact
            size = struct.unpack_from("<H", lay, 2)[0]
            raw = lay[4:4 + size]
        elif cls == 1:  # contiguous
            addr, size = struct.unpack_from("<QQ", lay, 2)
            raw = b"\0" * (n * item) if addr == UNDEF else f.buf[addr:addr + size]
        elif cls == 2:  # chunked
            raw = self._read_chunked(lay, n * item)
        else:
            raise Hdf5Error("data layout class %d" % cls)
        if len(raw) < n * item:
            raise Hdf5Error("dataset is truncated")
        if self._vlen_str:
            return self._read_vlen(raw, n)
        a = np.frombuffer(raw, dtype=self.dtype, count=n).reshape(self.shape)
        return a.copy()

    def _read_vlen(self, raw, n):
        out = []
        for i in range(n):
            size, addr, idx = struct.unpack_from("<IQI", raw, 16 * i)
            out.append(self._f.global_heap_object(addr, idx)[:size])
        a = np.empty(len(out), dtype=object)
        a[:] = out
        return a.reshape(self.shape) if self.shape else a[0]

    def _read_chunked(self, lay, nbytes):
        rank = lay[2]
        btree = struct.unpack_from("<Q", lay, 3)[0]
        cdims = struct.unpack_from("<%dI" % rank, lay, 11)
        item = cdims[-1]
        cshape = tuple(cdims[:-1])
        if len(cshape) != len(self.shape) or item != self.dtype.itemsize:
            raise Hdf5Error("chunk rank / element size does not match the dataset")
        out = np.zeros(self.shape, dtype=self.dtype)
        if btree == UNDEF:
            return out.tobytes()
        for offs, size, mask, addr in self._f.chunk_leaves(btree, rank):
            raw = self._f.buf[addr:addr + size]
            for k, (fid, cd) in reversed(list(enumerate(self._filters))):
                if mask & (1 << k):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    w = cd[0] if cd else item
                    raw = np.frombuffer(raw, dtype=np.uint8).reshape(w, -1).T.tobytes()
                else:
                    raise Hdf5Error("filter %d" % fid)
            chunk = np.frombuffer(raw, dtype=self.dtype, count=int(np.prod(cshape))).reshape(cshape)
            sel_o = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cshape, self.shape))
            sel_c = tuple(slice(0, s.stop - s.start) for s in sel_o)
            out[sel_o] = chunk[sel_c]
        return out.tobytes()


def _dataspace(data):
    ver = data[0]
    rank = data[1]
    if ver == 1:
        off = 8
    elif ver == 2:
        if data[3] == 2:  # null dataspace
            return (0,)
        off = 4
    else:
        raise Hdf5Error("dataspace message version %d" % ver)
    return tuple(struct.unpack_from("<%dQ" % rank, data, off)) if rank else ()


def _datatype(data):
    """-> (numpy dtype, is variable-length string)."""
    cls, ver = data[0] & 0x0F, data[0] >> 4
    bits0 = data[1]
    size = struct.unpack_from("<I", data, 4)[0]
    if cls == 0:  # fixed point
        order = ">" if bits0 & 1 else "<"
        signed = bool(bits0 & 0x08)
        return np.dtype("%s%s%d" % (order, "i" if signed else "u", size)), False
    if cls == 1:  # IEEE float (the only float h5py writes from numpy arrays)
        order = ">" if bits0 & 1 else "<"
        if size not in (2, 4, 8):
            raise Hdf5Error("%d-byte floating point type" % size)
        return np.dtype("%sf%d" % (order, size)), False
    if cls == 3:  # fixed-length string
        return np.dtype("S%d" % size), False
    if cls == 8:  # enumeration (h5py stores numpy bool as an enum over int8): read as its base type
        base, _ = _datatype(data[8:])
        return base, False
    if cls == 9:  # variable length
        if (bits0 & 0x0F) == 1:
            return np.dtype("V16"), True
        raise Hdf5Error("variable-length sequence type")
    raise Hdf5Error("datatype class %d (version %d)" % (cls, ver))


def _filters(data):
    ver, n = data[0], data[1]
    p = 8 if ver == 1 else 2
    out = []
    for _ in range(n):
        if ver == 1:
            fid, nlen, _flags, ncd = struct.unpack_from("<HHHH", data, p)
            p += 8 + ((nlen + 7) // 8) * 8
        else:
            fid = struct.unpack_from("<H", data, p)[0]
            p += 2
            nlen = 0
            if fid >= 256:
                nlen = struct.unpack_from("<H", data, p)[0]
                p += 2
            _flags, ncd = struct.unpack_from("<HH", data, p)
            p += 4 + nlen
        cd = struct.unpack_from("<%dI" % ncd, data, p)
        p += 4 * ncd
        if ver == 1 and ncd % 2:
            p += 4
        out.append((fid, cd))
    return out


class Hdf5File:
    """`Hdf5File(path)[name]` gives a numpy array (datasets) or a dict-like group; `keys()` lists the
    members of the root group.  Names may be paths ("ID/row_ids")."""

    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        b = self.buf
        if b[:8] != b"\x89HDF\r\n\x1a\n":
            raise Hdf5Error("%s is not an HDF5 file" % path)
        ver = b[8]
        if ver in (0, 1):
            if b[13] != 8 or b[14] != 8:
                raise Hdf5Error("offsets / lengths that are not 8 bytes wide")
            p = 24 + (4 if ver == 1 else 0)
            base = struct.unpack_from("<Q", b, p)[0]
            root_entry = p + 32
            self._root = struct.unpack_from("<Q", b, root_entry + 8)[0]
        elif ver in (2, 3):
            if b[9] != 8 or b[10] != 8:
                raise Hdf5Error("offsets / lengths that are not 8 bytes wide")
            base = struct.unpack_from("<Q", b, 12)[0]
            self._root = struct.unpack_from("<Q", b, 36)[0]
        else:
            raise Hdf5Error("superblock version %d" % ver)
        if base != 0:
            raise Hdf5Error("a non-zero base address")

    # ---- object headers ------------------------------------------------------------------
    def _messages(self, addr):
        b = self.buf
        msgs = []
        if b[addr:addr + 4] == b"OHDR":  # version 2
            flags = b[addr + 5]
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            width = 1 << (flags & 3)
            size = int.from_bytes(b[p:p + width], "little")
            p += width
            blocks = [(p, p + size)]
            order = bool(flags & 0x04)
            while blocks:
                p, end = blocks.pop(0)
                while p + 4 <= end - 4:
                    t = b[p]
                    s = struct.unpack_from("<H", b, p + 1)[0]
                    p += 4 + (2 if order else 0)
                    data = b[p:p + s]
                    p += s
                    if t == 0x10:
                        o, ln = struct.unpack_from("<QQ", data, 0)
                        blocks.append((o + 4, o + ln))  # skip the "OCHK" signature
                    elif t != 0:
                        msgs.append((t, data))
            return msgs
        ver, _, nmsg, _ref, size = struct.unpack_from("<BBHII", b, addr)
        if ver != 1:
            raise Hdf5Error("object header version %d" % ver)
        blocks = [(addr + 16, addr + 16 + size)]
        while blocks and nmsg > 0:
            p, end = blocks.pop(0)
            while p + 8 <= end and nmsg > 0:
                t, s = struct.unpack_from("<HH", b, p)
                data = b[p + 8:p + 8 + s]
                p += 8 + s
                nmsg -= 1
                if t == 0x10:
                    o, ln = struct.unpack_from("<QQ", data, 0)
                    blocks.append((o, o + ln))
                elif t != 0:
                    msgs.append((t, data))
        return msgs

    # ---- groups ----------------------------------------------------------------------------
    def _members(self, addr):
        """name -> object header address of a group's links."""
        out = {}
        for t, data in self._messages(addr):
            if t == 0x11:  # symbol table: B-tree + local heap
                btree, heap = struct.unpack_from("<QQ", data, 0)
                if self.buf[heap:heap + 4] != b"HEAP":
                    raise Hdf5Error("local heap signature")
                heap_data = struct.unpack_from("<Q", self.buf, heap + 24)[0]
                self._walk_group_tree(btree, heap_data, out)
            elif t == 0x06:  # link message (compact new-style group)
                name, target = self._link(data)
                if target is not None:
                    out[name] = target
            elif t == 0x02:  # link info: dense storage
                fractal = struct.unpack_from("<Q", data, 2 + (8 if data[1] & 1 else 0))[0]
                if fractal != UNDEF:
                    raise Hdf5Error("groups with dense link storage")
        return out

    @staticmethod
    def _link(data):
        flags = data[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = data[p]
            p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        width = 1 << (flags & 3)
        nlen = int.from_bytes(data[p:p + width], "little")
        p += width
        name = data[p:p + nlen].decode("utf-8")
        p += nlen
        if ltype != 0:
            return name, None  # soft / external links are not followed
        return name, struct.unpack_from("<Q", data, p)[0]

    def _walk_group_tree(self, addr, heap_data, out):
        b = self.buf
        if b[addr:addr + 4] != b"TREE" or b[addr + 4] != 0:
            raise Hdf5Error("group B-tree signature")
        level, n = b[addr + 5], struct.unpack_from("<H", b, addr + 6)[0]
        p = addr + 24
        for i in range(n):
            child = struct.unpack_from("<Q", b, p + 8)[0]
            p += 16
            if level > 0:
                self._walk_group_tree(child, heap_data, out)
                continue
            if b[child:child + 4] != b"SNOD":
                raise Hdf5Error("symbol table node signature")
            nsym = struct.unpack_from("<H", b, child + 6)[0]
            for s in range(nsym):
                e = child + 8 + 40 * s
                name_off, ohdr = struct.unpack_from("<QQ", b, e)
                q = heap_data + name_off
                name = b[q:b.index(b"\0", q)].decode("utf-8")
                out[name] = ohdr

    # ---- chunk index -----------------------------------------------------------------------
    def chunk_leaves(self, addr, rank):
        b = self.buf
        if b[addr:addr + 4] != b"TREE" or b[addr + 4] != 1:
            raise Hdf5Error("chunk B-tree signature")
        level, n = b[addr + 5], struct.unpack_from("<H", b, addr + 6)[0]
        key = 8 + 8 * rank
        p = addr + 24
        for i in range(n):
            size, mask = struct.unpack_from("<II", b, p)
            offs = struct.unpack_from("<%dQ" % (rank - 1), b, p + 8)
            child = struct.unpack_from("<Q", b, p + key)[0]
            p += key + 8
            if level > 0:
                yield from self.chunk_leaves(child, rank)
            else:
                yield offs, size, mask, child

    def global_heap_object(self, addr, idx):
        b = self.buf
        if b[addr:addr + 4] != b"GCOL":
            raise Hdf5Error("global heap signature")
        size = struct.unpack_from("<Q", b, addr + 8)[0]
        p, end = addr + 16, addr + size
        while p + 16 <= end:
            i, _ref, _, osz = struct.unpack_from("<HHIQ", b, p)
            if i == idx:
                return b[p + 16:p + 16 + osz]
            if i == 0:
                break
            p += 16 + ((osz + 7) // 8) * 8
        raise Hdf5Error("global heap object %d not found" % idx)

    # ---- public ----------------------------------------------------------------------------
    def _resolve(self, name):
        addr = self._root
        for part in [p for p in name.split("/") if p]:
            members = self._members(addr)
            if part not in members:
                raise KeyError(name)
            addr = members[part]
        return addr

    def keys(self, group=""):
        return sorted(self._members(self._resolve(group)))

    def is_group(self, name):
        kinds = {t for t, _ in self._messages(self._resolve(name))}
        return bool(kinds & {0x11, 0x02, 0x06}) and 0x08 not in kinds

    def __contains__(self, name):
        try:
            self._resolve(name)
            return True
        except KeyError:
            return False

    def __getitem__(self, name):
        """The dataset `name` as a numpy array (strings as bytes, as h5py gives them)."""
        return _Dataset(self, self._messages(self._resolve(name))).read()

    def shape(self, name):
        return _Dataset(self, self._messages(self._resolve(name))).shape
