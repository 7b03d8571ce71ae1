"""CPU restatement of the reference's x-vector loading and trial bookkeeping, in the reference's own
record-at-a-time style.  TEST INFRASTRUCTURE: imported only by tests/ and by the CPU-baseline leg of
tools/loader_probe.py; the product package never imports it.  Pinned by tests/golden/trials/
(written by the reference itself, oracle/gen_golden_io.py)."""
import struct

import numpy as np


def load_scp(path, sep=" "):
    """hyperion/utils/scp_list.py:203-229 with parse_script :159-201 -> keys, files, offsets, ranges."""
    keys, files, offsets, ranges = [], [], [], []
    with open(path) as f:
        for line in f:
            key, script = line.rstrip().split(sep, 1)
            fr = script.split("[", 1)
            fo = fr[0].split(":", 1)
            keys.append(key)
            files.append(fo[0])
            offsets.append(int(fo[1]))
            if len(fr) == 2:
                a, b = fr[1].rstrip("]").split(":", 1)
                a = int(a) if a.isdecimal() else 0
                ranges.append([a, int(b) - a + 1] if b.isdecimal() else [a, 0])
            else:
                ranges.append([0, 0])
    return keys, files, offsets, ranges


def read_record(f, row_offset=0, num_rows=0):
    """hyperion/utils/kaldi_io_funcs.py:17-22 (binary marker) + kaldi_matrix.py:63-117."""
    assert f.read(2) == b"\0B"
    token = b""
    while True:  # kaldi_io_funcs.py:41-55: characters up to the space
        c = f.read(1)
        if c == b" ":
            break
        token += c
    dtype = "float32" if token[:1] == b"F" else "float64"
    size = np.dtype(dtype).itemsize
    if token[1:2] == b"M":
        total = struct.unpack("<bi", f.read(5))[1] - row_offset
        if num_rows == 0:
            num_rows = total
        cols = struct.unpack("<bi", f.read(5))[1]
        f.seek(row_offset * cols * size, 1)
        return np.frombuffer(f.read(num_rows * cols * size), dtype=dtype).reshape(num_rows, cols)
    cols = struct.unpack("<bi", f.read(5))[1]
    return np.frombuffer(f.read(cols * size), dtype=dtype)


def read_vectors(scp_path, wanted):
    """RandomAccessArkDataReader.read(keys, squeeze=True) (hyperion/io/ark_data_reader.py:554-622,
    data_reader.py:59-83): dictionary lookup, seek, parse, one array per key, then stacked."""
    keys, files, offsets, ranges = load_scp(scp_path)
    index = {k: i for i, k in enumerate(keys)}
    handles = {}
    data = []
    for k in wanted:
        i = index[k]
        if files[i] not in handles:
            handles[files[i]] = open(files[i], "rb")
        f = handles[files[i]]
        f.seek(offsets[i], 0)
        data.append(np.expand_dims(read_record(f, ranges[i][0], ranges[i][1]), 0))
    for f in handles.values():
        f.close()
    return np.concatenate(tuple(data), axis=0)


def load_ndx(path):
    """TrialNdx.load_txt (hyperion/utils/trial_ndx.py:126-149)."""
    with open(path) as f:
        fields = [line.split() for line in f]
    model_set, mi = np.unique([x[0] for x in fields], return_inverse=True)
    seg_set, si = np.unique([x[1] for x in fields], return_inverse=True)
    mask = np.zeros((len(model_set), len(seg_set)), dtype=bool)
    for a, b in zip(mi, si):
        mask[a, b] = True
    return model_set, seg_set, mask


def split_range(n, idx, num_parts):
    """hyperion/utils/list_utils.py:108-111."""
    return int(np.floor((idx - 1) * float(n) / num_parts)), int(np.floor(idx * float(n) / num_parts))


def trial_data(scp_path, ndx_path, enroll_path, model_part=(1, 1), seg_part=(1, 1)):
    """TrialDataReader for eval_set enroll-test (hyperion/helpers/trial_data_reader.py:39-72):
    enrolment rows whose model is in the (split) index, in list order; test segments of the split."""
    model_set, seg_set, mask = load_ndx(ndx_path)
    with open(enroll_path) as f:
        enroll = [line.split() for line in f]
    m0, m1 = split_range(len(model_set), *model_part)
    s0, s1 = split_range(len(seg_set), *seg_part)
    model_set, seg_set, mask = model_set[m0:m1], seg_set[s0:s1], mask[m0:m1, s0:s1]
    keep = set(model_set)
    enroll = [e for e in enroll if e[1] in keep]
    x_e = read_vectors(scp_path, [e[0] for e in enroll])
    x_t = read_vectors(scp_path, list(seg_set))
    return x_e, x_t, np.array([e[1] for e in enroll]), model_set, seg_set, mask
