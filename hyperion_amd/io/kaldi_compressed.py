"""Kaldi compressed matrices (hyperion/utils/kaldi_matrix.py:255-834): the four on-disk formats of binary
archives, decoded to and encoded from numpy arrays with the reference's arithmetic, value for value.

A record is the token `CM `, `CM2 `, `CM3 ` or `CM4 ` followed by a 16-byte header (float32 minimum,
float32 range, int32 rows, int32 columns) and
  CM  (format 1, "speech-feat"): per column four uint16 levels (the 0th, 25th, 75th and 100th percentile
       of the column on the 16-bit scale of the header), then one byte per value, column by column:
       0..64 runs from the first level to the second, 64..192 to the third, 192..255 to the fourth;
  CM4 (format 4, "speech-feat-t"): the same with the bytes stored row by row;
  CM2 (format 2): one uint16 per value on the scale of the header, row by row;
  CM3 (format 3): one uint8 per value.
Compressed matrices are a storage format of feature archives; x-vectors are stored as plain vectors.
Decoding and encoding are host work on the bytes of a record (float64 out, like the reference's reader)."""
import struct

import numpy as np

from ..hyp_defs import float_cpu

TOKENS = {b"CM": 1, b"CM2": 2, b"CM3": 3, b"CM4": 4}
FORMAT_TOKEN = {1: b"CM ", 2: b"CM2 ", 3: b"CM3 ", 4: b"CM4 "}
METHOD_FORMAT = {"speech-feat": 1, "2byte-auto": 2, "2byte-signed-integer": 2, "1byte-auto": 3,
                 "1byte-unsigned-integer": 3, "1byte-0-1": 3, "speech-feat-t": 4}
_INV_65535 = 1.52590218966964e-05  # the reference's literal for 1 / 65535


def read_token(f):
    """The token at the file position (up to and including the space), without the space."""
    token = b""
    while True:
        c = f.read(1)
        if c in (b" ", b""):
            return token
        token += c


def read_header(f, fmt):
    """(min_value, data_range, rows, cols) of a record whose token has just been read."""
    lo, rng, rows, cols = struct.unpack("<ffii", f.read(16))
    return lo, rng, rows, cols


def body_bytes(fmt, rows, cols):
    """Bytes of a record behind its 16-byte header (kaldi_matrix.py:484-495)."""
    if fmt in (1, 4):
        return cols * (8 + rows)
    return (2 if fmt == 2 else 1) * rows * cols


def _levels(lo, rng, codes16):
    """uint16 codes -> values on the header's scale (kaldi_matrix.py:540-543)."""
    return lo + rng * _INV_65535 * codes16.astype(float_cpu())


def _bytes_to_values(codes, p):
    """Byte codes [rows, cols] with the four levels p [4, cols] of every column -> values
    (kaldi_matrix.py:650-660): three linear pieces."""
    v = codes.astype(float_cpu())
    p0, p25, p75, p100 = p[0][None, :], p[1][None, :], p[2][None, :], p[3][None, :]
    out = np.empty(v.shape, dtype=float_cpu())
    low, high = v <= 64, v > 192
    mid = ~(low | high)
    out[low] = (p0 + (p25 - p0) * v / 64.0)[low]
    out[mid] = (p25 + (p75 - p25) * (v - 64) / 128.0)[mid]
    out[high] = (p75 + (p100 - p75) * (v - 192) / 63.0)[high]
    return out


def read(f, fmt, row_offset=0, num_rows=0):
    """The matrix (or `num_rows` rows from `row_offset`; 0 = to the end) of the compressed record whose
    token (format `fmt`) has just been read from `f`; float_cpu() values
    (kaldi_matrix.py:440-481, 662-687, 699-785).  The file position ends behind the record."""
    lo, rng, rows, cols = read_header(f, fmt)
    assert row_offset <= rows, "row_offset (%d) > num_rows (%d)" % (row_offset, rows)
    avail = rows - row_offset
    if num_rows == 0:
        num_rows = avail
    assert num_rows <= avail, "requested rows (%d) > available rows (%d)" % (num_rows, avail)
    left = avail - num_rows
    if fmt in (1, 4):
        levels = np.frombuffer(f.read(8 * cols), dtype="<u2").reshape(cols, 4)
        p = _levels(lo, rng, levels).T  # [4, cols]
        if fmt == 1:  # column-major bytes: the rows wanted are a slice of every column
            codes = np.frombuffer(f.read(rows * cols), dtype=np.uint8).reshape(cols, rows)
            codes = codes[:, row_offset:row_offset + num_rows].T
        else:
            f.seek(row_offset * cols, 1)
            codes = np.frombuffer(f.read(num_rows * cols), dtype=np.uint8).reshape(num_rows, cols)
            f.seek(left * cols, 1)
        return _bytes_to_values(codes, p)
    width = 2 if fmt == 2 else 1
    f.seek(row_offset * cols * width, 1)
    raw = f.read(num_rows * cols * width)
    f.seek(left * cols * width, 1)
    if fmt == 2:
        vals = _levels(lo, rng, np.frombuffer(raw, dtype="<u2"))
    else:
        vals = lo + rng / 255.0 * np.frombuffer(raw, dtype=np.uint8).astype(float_cpu())
    return vals.reshape(num_rows, cols).astype(float_cpu(), copy=False)


def read_shape(f, fmt, skip=True):
    """(rows, cols) of the record whose token has just been read; with skip the position moves behind
    the record (kaldi_matrix.py:836-872)."""
    _, _, rows, cols = read_header(f, fmt)
    if skip:
        f.seek(body_bytes(fmt, rows, cols), 1)
    return rows, cols


# ---- encoding ----------------------------------------------------------------------------------

def _to_u16(x, lo, rng):
    """Values -> uint16 codes on the scale (lo, rng), clipped (kaldi_matrix.py:528-532)."""
    f = (np.asarray(x).ravel() - lo) / rng
    f[f > 1.0] = 1
    f[f < 0.0] = 0
    return (f * 65535 + 0.499).astype(np.uint16)


def _to_u8(x, lo, rng):
    f = (np.asarray(x).ravel() - lo) / rng
    f[f > 1.0] = 1
    f[f < 0.0] = 0
    return (f * 255 + 0.499).astype(np.uint8)


def _column_levels(col, rows, lo, rng):
    """The four strictly increasing uint16 levels of a column (kaldi_matrix.py:550-595): the order
    statistics at 0, rows/4, 3 rows/4 and the maximum (the first four values for fewer than five
    rows), each at least one above the previous and leaving room for the following ones."""
    one = np.uint16(1)
    srt = np.sort(col)
    if rows >= 5:
        q = int(rows / 4)
        picks = [srt[0], srt[q], srt[3 * q], srt[-1]]
    else:
        picks = [srt[i] if i < rows else None for i in range(4)]
    lv = [min(_to_u16(picks[0], lo, rng)[0], np.uint16(65532))]
    for i, cap in ((1, 65533), (2, 65534), (3, None)):
        nxt = lv[-1] + one
        if picks[i] is not None:
            nxt = max(_to_u16(picks[i], lo, rng)[0], nxt)
        if cap is not None and (rows >= 5 or picks[i] is not None):
            nxt = min(nxt, np.uint16(cap))
        lv.append(nxt)
    return lv


def _values_to_bytes(v, p0, p25, p75, p100):
    """A column -> byte codes with its four levels as values (kaldi_matrix.py:626-647)."""
    out = np.zeros(v.shape, dtype=np.int32)
    low, high = v < p25, v >= p75
    mid = ~(low | high)
    for sel, a, b, base, steps, top in ((low, p0, p25, 0, 64, 64), (mid, p25, p75, 64, 128, 192),
                                        (high, p75, p100, 192, 63, 255)):
        c = base + (((v[sel] - a) / (b - a)) * steps + 0.5).astype(np.int32)
        c[c < base] = base
        c[c > top] = top
        out[sel] = c
    return out.astype(np.uint8)


def compress(mat, method="auto"):
    """mat [rows, cols] (the dtype it is stored from: the archive writer hands float32) -> the bytes of
    the record from its token on (kaldi_matrix.py:388-437, 497-526, 797-828)."""
    mat = np.asarray(mat)
    if mat.ndim != 2:
        raise ValueError("only matrices are stored compressed (got %d dimension(s))" % mat.ndim)
    if method == "auto":
        method = "speech-feat" if mat.shape[0] > 8 else "2byte-auto"
    if method not in METHOD_FORMAT:
        raise ValueError(method)
    fmt = METHOD_FORMAT[method]
    rows, cols = mat.shape
    if method in ("speech-feat", "speech-feat-t", "2byte-auto", "1byte-auto"):
        lo, hi = np.min(mat), np.max(mat)
        if hi == lo:
            hi = lo + 1 + np.abs(lo)
        assert lo - lo == 0 and hi - hi == 0, "cannot compress matrix with infs or nans"
        rng = hi - lo
        assert rng > 0
    elif method == "2byte-signed-integer":
        lo, rng = -32768.0, 65535.0
    elif method == "1byte-unsigned-integer":
        lo, rng = 0.0, 255.0
    else:
        lo, rng = 0.0, 1.0
    head = FORMAT_TOKEN[fmt] + struct.pack("<ffii", lo, rng, rows, cols)
    if fmt == 2:
        return head + _to_u16(mat, lo, rng).tobytes()
    if fmt == 3:
        return head + _to_u8(mat, lo, rng).tobytes()
    levels, codes = [], np.empty((cols, rows), dtype=np.uint8)
    for c in range(cols):
        col = mat[:, c]
        lv = _column_levels(col, rows, lo, rng)
        levels.append(struct.pack("<HHHH", *[int(x) for x in lv]))
        p = lo + rng * _INV_65535 * np.array(lv, dtype=np.uint16).astype(float_cpu())
        codes[c] = _values_to_bytes(col, p[0], p[1], p[2], p[3])
    body = codes.tobytes() if fmt == 1 else np.ascontiguousarray(codes.T).tobytes()
    return head + b"".join(levels) + body
