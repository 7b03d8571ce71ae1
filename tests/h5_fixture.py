"""TEST INFRASTRUCTURE: writes small HDF5 files with the file structures of the trial files the
reference's tests hold (superblock 0, symbol-table groups, version 1 object headers, contiguous
datasets), so that the model-loading code can be run end to end without h5py.  Not a general HDF5
writer and not part of the package: whether another HDF5 library accepts these files is not
claimed, only that they use the structures the pinned reader was checked on."""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class _Buf:
    def __init__(self):
        self.b = bytearray(96)  # superblock + root symbol table entry, filled in at the end

    def put(self, data):
        while len(self.b) % 8:
            self.b.append(0)
        addr = len(self.b)
        self.b += data
        return addr


def _msg(kind, data):
    data = bytes(data) + b"\0" * (-len(data) % 8)
    return struct.pack("<HHB3x", kind, len(data), 0) + data


def _header(buf, msgs):
    body = b"".join(msgs)
    return buf.put(struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body)


def _dataset(buf, a):
    a = np.asarray(a)
    if a.dtype.kind in "US":
        raw = a.astype("S").tobytes()
        size = a.astype("S").dtype.itemsize
        dtype = struct.pack("<B3BI", 0x13, 0, 0, 0, size)
    else:
        a = a.astype("<f4") if a.dtype == np.float32 else a.astype("<f8")
        raw = a.tobytes()
        if a.dtype.itemsize == 8:
            dtype = struct.pack("<B3BIHHBBBBI", 0x11, 0x20, 0x3F, 0, 8, 0, 64, 52, 11, 0, 52, 1023)
        else:
            dtype = struct.pack("<B3BIHHBBBBI", 0x11, 0x20, 0x1F, 0, 4, 0, 32, 23, 8, 0, 23, 127)
    addr = buf.put(raw) if raw else UNDEF
    space = struct.pack("<BBB5x", 1, a.ndim, 0) + b"".join(struct.pack("<Q", d) for d in a.shape)
    layout = struct.pack("<BBQQ", 3, 1, addr, len(raw))
    return _header(buf, [_msg(0x01, space), _msg(0x03, dtype), _msg(0x08, layout)])


def _group(buf, members):
    """members: {name: array or dict}; returns the group's object header address."""
    assert len(members) <= 8, "one symbol table node holds 8 entries"
    addrs = {n: (_group(buf, v) if isinstance(v, dict) else _dataset(buf, v)) for n, v in members.items()}
    names = sorted(addrs)
    heap = bytearray(8)
    offs = {}
    for n in names:
        offs[n] = len(heap)
        raw = n.encode() + b"\0"
        heap += raw + b"\0" * (-len(raw) % 8)
    heap_data = buf.put(bytes(heap))
    heap_addr = buf.put(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), UNDEF, heap_data))
    snod = b"SNOD" + struct.pack("<BBH", 1, 0, len(names))
    for n in names:
        snod += struct.pack("<QQII16x", offs[n], addrs[n], 0, 0)
    snod_addr = buf.put(snod + b"\0" * (328 - len(snod)))
    tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod_addr,
                                                                                offs[names[-1]] if names else 0)
    tree_addr = buf.put(tree + b"\0" * (544 - len(tree)))
    return _header(buf, [_msg(0x11, struct.pack("<QQ", tree_addr, heap_addr))])


def write_h5(path, members):
    """members: nested dict of arrays; names with "/" are not split (pass nested dicts)."""
    buf = _Buf()
    root = _group(buf, members)
    sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, len(buf.b), UNDEF)
    sb += struct.pack("<QQII16x", 0, root, 0, 0)
    buf.b[:len(sb)] = sb
    with open(path, "wb") as f:
        f.write(bytes(buf.b))


def nested(flat):
    """{"a/b": x} -> {"a": {"b": x}}."""
    out = {}
    for k, v in flat.items():
        d = out
        parts = k.split("/")
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = v
    return out
