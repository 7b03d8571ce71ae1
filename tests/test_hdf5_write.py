"""The package's HDF5 writer (hyperion_amd/io/hdf5_write.py): files are read back by the reader that
is pinned on library-written files, their structures are walked independently of that reader, and
the header messages of a re-saved trial key are compared byte for byte with those the HDF5 library
wrote into the reference's own test file.  No HDF5 library is installed: acceptance of these files
by the library itself is parity unpinned."""
import json
import os
import struct

import numpy as np
import pytest

from hyperion_amd.io.hdf5_lite import Hdf5File
from hyperion_amd.io.hdf5_write import File

HERE = os.path.dirname(os.path.abspath(__file__))
KEY = os.path.join(HERE, "golden", "trials", "core-core_det5_key.h5")
NDX = os.path.join(HERE, "golden", "trials", "core-core_det5_ndx.h5")
UNDEF = 0xFFFFFFFFFFFFFFFF


def _walk_tree(b, addr, heap_data, level_expected=None):
    """Names of a group B-tree in key order, checking node invariants on the way."""
    assert b[addr:addr + 4] == b"TREE" and b[addr + 4] == 0
    level, n = b[addr + 5], struct.unpack_from("<H", b, addr + 6)[0]
    assert level_expected is None or level == level_expected
    assert n <= 32
    keys = [struct.unpack_from("<Q", b, addr + 24 + 16 * i)[0] for i in range(n + 1)]
    names = []
    for i in range(n):
        child = struct.unpack_from("<Q", b, addr + 32 + 16 * i)[0]
        if level > 0:
            sub = _walk_tree(b, child, heap_data, level - 1)
        else:
            assert b[child:child + 4] == b"SNOD" and b[child + 4] == 1
            nsym = struct.unpack_from("<H", b, child + 6)[0]
            assert 1 <= nsym <= 8
            sub = []
            for s in range(nsym):
                off = struct.unpack_from("<Q", b, child + 8 + 40 * s)[0]
                q = heap_data + off
                sub.append((bytes(b[q:b.index(b"\0", q)]), off))
        # every name of child i lies in (key[i], key[i + 1]], keys being names in the heap
        lo = bytes(b[heap_data + keys[i]:b.index(b"\0", heap_data + keys[i])])
        hi = bytes(b[heap_data + keys[i + 1]:b.index(b"\0", heap_data + keys[i + 1])])
        assert all(lo < nm <= hi for nm, _ in sub), (lo, hi, sub[:2])
        names += sub
    return names


def _check_group(b, hdr):
    ver, nmsg, _, size = struct.unpack_from("<BxHII", b, hdr)
    assert ver == 1 and nmsg == 1 and size == 24
    kind, msize = struct.unpack_from("<HH", b, hdr + 16)
    assert kind == 0x11 and msize == 16
    tree, heap = struct.unpack_from("<QQ", b, hdr + 24)
    assert b[heap:heap + 4] == b"HEAP"
    hsize, free, hdata = struct.unpack_from("<QQQ", b, heap + 8)
    assert free == 1 and hdata % 8 == 0 and hsize % 8 == 0  # 1: no free block
    names = [n for n, _ in _walk_tree(b, tree, hdata)]
    assert names == sorted(names) and len(set(names)) == len(names)
    return names, tree, heap


def test_structures_of_a_large_file(tmp_path):
    """3000 datasets in three groups (a two-level B-tree each), every supported element type."""
    rng = np.random.RandomState(0)
    path = str(tmp_path / "big.h5")
    data = {"grp%d/utt-%05d" % (i % 3, i): rng.randn(7).astype(np.float32) for i in range(3000)}
    extra = {"m": np.arange(12, dtype=np.int8).reshape(3, 4) - 5, "u": np.arange(6, dtype=np.uint16).reshape(3, 2),
             "q": np.arange(4, dtype=np.int64) << 40, "d": rng.randn(4, 5), "e": np.zeros((0, 4)),
             "deep/er/flag": np.array([True, False]), "names": np.array(["a", "bcd", ""]),
             "strided": np.arange(12.0).reshape(3, 4)[:, ::2]}
    with File(path, "w") as f:
        f.create_dataset("config", data=np.array(json.dumps({"a": 1}), dtype="S"))
        for k, v in list(data.items()) + list(extra.items()):
            f.create_dataset(k, data=v)
        assert "grp1/utt-00001" in f and "grp1/none" not in f and "deep/er" in f
        with pytest.raises(ValueError):
            f.create_dataset("m", data=np.zeros(2))
        with pytest.raises(TypeError):
            f.create_dataset("c", data=np.zeros(2, dtype=complex))
    g = Hdf5File(path)
    assert json.loads(str(np.asarray(g["config"]).astype("U"))) == {"a": 1}
    assert len(g.keys("grp0")) == 1000 and g.keys() == sorted(["config", "grp0", "grp1", "grp2", "deep"] + [
        k for k in extra if "/" not in k])
    for k, v in data.items():
        assert np.array_equal(g[k], v)
    for k, v in extra.items():
        got = np.asarray(g[k])
        if v.dtype.kind == "U":
            assert got.astype("U").tolist() == v.tolist()
        elif v.dtype.kind == "b":
            assert got.dtype == np.uint8 and got.tolist() == v.astype(int).tolist()
        else:
            assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v)

    b = open(path, "rb").read()
    assert bytes(b[:8]) == b"\x89HDF\r\n\x1a\n" and b[8] == 0 and b[13] == 8 and b[14] == 8
    leaf_k, internal_k = struct.unpack_from("<HH", b, 16)
    assert (leaf_k, internal_k) == (4, 16)
    base, _, eof, _ = struct.unpack_from("<QQQQ", b, 24)
    assert base == 0 and eof == len(b)
    _, root, cache, _, tree, heap = struct.unpack_from("<QQIIQQ", b, 56)
    names, t2, h2 = _check_group(b, root)
    assert cache == 1 and (tree, heap) == (t2, h2)
    assert names == [n.encode() for n in g.keys()]
    sub, t3, _ = _check_group(b, g._resolve("grp2"))
    assert len(sub) == 1000 and b[t3 + 5] == 1  # 125 symbol-table nodes -> 4 leaf-level nodes under one root
    assert _check_group(b, g._resolve("deep/er"))[0] == [b"flag"]
    # sibling links of the leaf-level nodes of grp2, left to right
    first = struct.unpack_from("<Q", b, t3 + 32)[0]
    chain, a = [], first
    while a != UNDEF:
        chain.append(a)
        left, right = struct.unpack_from("<QQ", b, a + 8)
        assert left == (chain[-2] if len(chain) > 1 else UNDEF)
        a = right
    assert len(chain) == 4


def test_messages_match_the_library_written_file(tmp_path):
    """A key loaded from the reference's test file and saved again: the dataspace and datatype
    messages of the datasets are the bytes the HDF5 library wrote; superblock fields agree."""
    from hyperion_amd.utils.trial_ndx import TrialKey, TrialNdx
    from hyperion_amd.utils.trial_scores import TrialScores

    key = TrialKey.load(KEY)
    path = str(tmp_path / "key.h5")
    key.save(path)
    assert TrialKey.load(path) == key
    ref, got = Hdf5File(KEY), Hdf5File(path)

    def msgs(f, name):
        return {m[0]: bytes(m[1]) for m in f._messages(f._resolve(name))}

    for name in ("trial_mask", "ID/row_ids", "ID/column_ids"):
        r, g = msgs(ref, name), msgs(got, name)
        assert r[0x01] == g[0x01], name  # dataspace
        if name == "trial_mask":
            assert r[0x03] == g[0x03]  # int8
        else:  # fixed-length strings: the test file holds null-terminated ones of one more byte
            assert r[0x03][0] == g[0x03][0] == 0x13 and g[0x03][1] == 1
            assert struct.unpack_from("<I", r[0x03], 4)[0] == struct.unpack_from("<I", g[0x03], 4)[0] + 1
    assert msgs(ref, "ID/row_ids")[0x05] == msgs(got, "ID/row_ids")[0x05]  # fill value message
    assert msgs(ref, "ID/row_ids")[0x08][:2] == msgs(got, "ID/row_ids")[0x08][:2]  # contiguous layout, version 3
    assert bytes(ref.buf[:24]) == bytes(got.buf[:24])  # signature, versions, sizes, group K values

    ndx = TrialNdx.load(NDX)
    p2 = str(tmp_path / "ndx.h5")
    ndx.save(p2)
    assert TrialNdx.load(p2) == ndx
    rng = np.random.RandomState(3)
    scr = TrialScores(ndx.model_set, ndx.seg_set, rng.randn(*ndx.trial_mask.shape), ndx.trial_mask)
    p3 = str(tmp_path / "scores.h5")
    scr.save(p3)
    back = TrialScores.load(p3)
    assert back == scr and back.scores.dtype == np.float64


def test_models_and_vector_files_round_trip(tmp_path):
    """save -> load of a TransformList, the three PLDA models and a vector file with a script."""
    from hyperion_amd.helpers.plda_factory import PLDAFactory
    from hyperion_amd.io import DataWriterFactory, RandomAccessDataReaderFactory
    from hyperion_amd.pdfs.plda import FRPLDA, PLDA, SPLDA
    from hyperion_amd.transforms import LDA, PCA, CentWhiten, LNorm, TransformList

    rng = np.random.RandomState(0)
    f32 = lambda *s: rng.randn(*s).astype(np.float32).astype(np.float64)  # noqa: E731  (what float_save keeps)
    t = TransformList([CentWhiten(mu=f32(6), T=f32(6, 6), name="cw"), LNorm(name="ln"),
                       LDA(mu=None, T=f32(6, 3), lda_dim=3, name="lda"), PCA(mu=f32(3), T=f32(3, 2), pca_dim=2, name="pca")])
    path = str(tmp_path / "sub" / "preproc.h5")
    t.save(path)
    back = TransformList.load(path)
    assert [type(x) for x in back.transforms] == [CentWhiten, LNorm, LDA, PCA]
    assert TransformList.load_config(path) == json.loads(t.to_json())
    for a, c in zip(t.transforms, back.transforms):
        assert a.name == c.name
        for p in ("mu", "T"):
            assert (getattr(a, p) is None and getattr(c, p) is None) or np.array_equal(getattr(a, p), getattr(c, p))
    assert back.transforms[2].lda_dim == 3 and back.transforms[3].pca_dim == 2
    assert Hdf5File(path)["cw/T"].dtype == np.float32

    W = np.eye(6) + 0.125  # exact in float32, the type model files are stored in
    models = [("splda", SPLDA(y_dim=2, mu=f32(6), V=f32(2, 6), W=W, update_V=False, name="plda")),
              ("frplda", FRPLDA(mu=f32(6), B=2 * W, W=W, name="fr")),
              ("plda", PLDA(y_dim=2, z_dim=3, mu=f32(6), V=f32(2, 6), U=f32(3, 6), D=(np.abs(f32(6)) + 1).astype(np.float32).astype(np.float64), name="full"))]
    for kind, m in models:
        p = str(tmp_path / (kind + ".h5"))
        m.save(p)
        b = PLDAFactory.load_plda(kind, p)
        assert type(b) is type(m) and b.name == m.name and b.get_config() == m.get_config()
        for k in m._param_names:
            assert np.array_equal(getattr(b, k), getattr(m, k)), (kind, k)

    vecs = {"spk1/utt-a": f32(5), "spk1/utt-b": f32(5), "solo": f32(5)}
    arch, scp = str(tmp_path / "v" / "xvector.h5"), str(tmp_path / "v" / "xvector.scp")
    with DataWriterFactory.create("h5,scp:%s,%s" % (arch, scp)) as w:
        w.write(list(vecs), list(vecs.values()))
        w.write("mat", f32(4, 5))
        with pytest.raises(AssertionError):
            w.write("bad key", f32(5))
    assert open(scp).read().splitlines()[0] == "spk1/utt-a " + arch
    for spec in (arch, "scp:" + scp):
        r = RandomAccessDataReaderFactory.create(spec)
        x = r.read(["solo", "spk1/utt-b"], squeeze=True)
        assert np.array_equal(x, np.stack([vecs["solo"], vecs["spk1/utt-b"]]))
        assert r.read(["mat"])[0].shape == (4, 5)
    assert DataWriterFactory.parse("ark,scp,f:a.ark,a.scp") == ("ark", "a.ark", "a.scp", True)
    assert DataWriterFactory.parse("x.h5") == ("h5", "x.h5", None, False)
