"""The package's read-only HDF5 parser on the two HDF5 files the reference's own tests hold
(tests/data_in/core-core_det5_{ndx,key}.h5, copied as data to tests/golden/trials/): trial index and
key as the reference's loaders read them (hyperion/utils/trial_ndx.py:110-124,
trial_key.py:163-211), then the operations the reference's tests run on those objects
(tests/hyperion/utils/test_trial_ndx.py, test_trial_key.py).  Vector files written by the
reference's H5DataWriter have no fixture here (and h5py is absent): the vector readers are checked
on these files' datasets only."""
import os

import numpy as np
import pytest

from hyperion_amd.io.hdf5_lite import Hdf5Error, Hdf5File
from hyperion_amd.utils import TrialKey, TrialNdx

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trials")
NDX = os.path.join(HERE, "core-core_det5_ndx.h5")
KEY = os.path.join(HERE, "core-core_det5_key.h5")


def test_structure_of_the_reference_files():
    f = Hdf5File(NDX)
    assert f.keys() == ["ID", "trial_mask"] and f.keys("ID") == ["column_ids", "row_ids"]
    assert f.is_group("ID") and not f.is_group("trial_mask")
    assert "ID/row_ids" in f and "ID/nothing" not in f and "nothing" not in f
    rows, cols, mask = f["ID/row_ids"], f["ID/column_ids"], f["trial_mask"]
    assert rows.shape == (580,) and rows.dtype == np.dtype("S6") and rows[0] == b"32708"
    assert cols.shape == (712,) and cols[0] == b"sre10_phn/taczw-a"
    assert mask.shape == (580, 712) and mask.dtype == np.uint8 and int(mask.sum()) == 30373
    assert f.shape("trial_mask") == (580, 712)
    k = Hdf5File(KEY)
    assert k["file_format"].tolist() == [b"2.0"]
    km = k["trial_mask"]
    assert km.dtype == np.int8
    assert dict(zip(*np.unique(km, return_counts=True))) == {-1: 29665, 0: 382587, 1: 708}
    with pytest.raises(KeyError):
        f["ID/missing"]
    with pytest.raises(Hdf5Error):
        Hdf5File(os.path.join(HERE, "trials.ndx"))


def test_index_and_key_agree():
    ndx, key = TrialNdx.load(NDX), TrialKey.load(KEY)
    ndx.validate()
    key.validate()
    assert ndx.trial_mask.shape == (580, 712) and ndx.num_tests == 712 and int(ndx.trial_mask.sum()) == 30373
    assert int(key.tar.sum()) == 708 and int(key.non.sum()) == 29665
    assert key.to_ndx() == ndx
    assert not np.any(key.tar & key.non)


def test_reference_operations_on_the_loaded_index(tmp_path):
    """sort / merge / filter / split / text round trip, as tests/hyperion/utils/test_trial_ndx.py
    does them on this file."""
    ndx1 = TrialNdx.load(NDX)
    ndx1.sort()
    assert list(ndx1.model_set) == sorted(ndx1.model_set) and list(ndx1.seg_set) == sorted(ndx1.seg_set)
    ndx2 = TrialNdx(ndx1.model_set[:10], ndx1.seg_set, ndx1.trial_mask[:10, :])
    ndx3 = TrialNdx(ndx1.model_set[5:], ndx1.seg_set, ndx1.trial_mask[5:, :])
    assert TrialNdx.merge([ndx2, ndx3]) == ndx1
    ndx2 = TrialNdx(ndx1.model_set[:5], ndx1.seg_set[:10], ndx1.trial_mask[:5, :10])
    assert ndx1.filter(ndx2.model_set, ndx2.seg_set, keep=True) == ndx2
    parts = [[ndx1.split(i + 1, 3, j + 1, 2) for j in range(2)] for i in range(3)]
    rows = [TrialNdx.merge(p) for p in parts]
    assert TrialNdx.merge(rows) == ndx1
    path = str(tmp_path / "ndx.txt")
    ndx1.save(path)
    back = TrialNdx.load(path)
    # a text list only holds the rows / columns that have trials
    keep_m, keep_s = ndx1.trial_mask.any(axis=1), ndx1.trial_mask.any(axis=0)
    assert back == TrialNdx(ndx1.model_set[keep_m], ndx1.seg_set[keep_s], ndx1.trial_mask[keep_m][:, keep_s])


def test_reference_operations_on_the_loaded_key(tmp_path):
    key1 = TrialKey.load(KEY)
    key1.sort()
    key2 = TrialKey(key1.model_set[:10], key1.seg_set, key1.tar[:10, :], key1.non[:10, :])
    key3 = TrialKey(key1.model_set[5:], key1.seg_set, key1.tar[5:, :], key1.non[5:, :])
    assert TrialKey.merge([key2, key3]) == key1
    key2 = TrialKey(key1.model_set[:5], key1.seg_set[:10], key1.tar[:5, :10], key1.non[:5, :10])
    assert key1.filter(key2.model_set, key2.seg_set, keep=True) == key2
    path = str(tmp_path / "key.txt")
    key1.save(path)
    back = TrialKey.load(path)
    assert int(back.tar.sum()) == 708 and int(back.non.sum()) == 29665


def test_vector_reader_interfaces_on_a_real_file(tmp_path):
    from hyperion_amd.io import RandomAccessDataReaderFactory as F
    from hyperion_amd.io.h5 import RandomAccessH5FileDataReader, RandomAccessH5ScriptDataReader

    r = F.create(NDX)  # a bare path is an HDF5 archive
    assert isinstance(r, RandomAccessH5FileDataReader) and r.keys() == ["ID", "trial_mask"]
    ref = Hdf5File(NDX)["trial_mask"].astype(np.float64)
    (m,) = r.read(["trial_mask"])
    assert m.dtype == np.float64 and np.array_equal(m, ref)
    (part,) = r.read("trial_mask", row_offset=3, num_rows=5)
    assert np.array_equal(part, ref[3:8])
    both = r.read(["trial_mask", "trial_mask"], squeeze=True, row_offset=[0, 10], num_rows=[4, 4])
    assert both.shape == (2, 4, 712) and np.array_equal(both[1], ref[10:14])
    assert r.read_shapes(["trial_mask"]) == [(580, 712)]
    assert r.read_num_rows(["trial_mask"]).tolist() == [580] and r.read_dims(["trial_mask"]).tolist() == [712]
    with pytest.raises(Exception, match="not found"):
        r.read(["nothing"])
    p = RandomAccessH5FileDataReader(NDX, permissive=True)
    assert p.read(["nothing"])[0].size == 0
    scp = tmp_path / "m.scp"
    scp.write_text("trial_mask %s[2:5]\n" % NDX)
    s = F.create("scp:%s" % scp)
    assert isinstance(s, RandomAccessH5ScriptDataReader)
    assert np.array_equal(s.read(["trial_mask"])[0], ref[2:6])
    assert np.array_equal(s.read(["trial_mask"], row_offset=1, num_rows=2)[0], ref[3:5])


def test_model_file_interfaces(tmp_path):
    """Model files (a JSON "config" string and one float dataset per parameter,
    hyperion/hyp_model.py:52-135) use the same file structures as the trial files above; none ships
    with the reference, so only the pieces that can be exercised on those are: parameter look-up
    with and without a name prefix, missing parameters, and the JSON-text fall-back of load_config."""
    from hyperion_amd.helpers.plda_factory import PLDAFactory
    from hyperion_amd.hyp_model import HypModel
    from hyperion_amd.pdfs.plda import SPLDA
    from hyperion_amd.transforms import LDA, TransformList

    f = Hdf5File(NDX)
    p = HypModel._load_params_to_dict(f, None, ["trial_mask", "absent"])
    assert p["absent"] is None and p["trial_mask"].dtype == np.float64 and p["trial_mask"].shape == (580, 712)
    assert HypModel._load_params_to_dict(f, "ID", ["absent"]) == {"absent": None}
    text = tmp_path / "config.json"
    text.write_text('{"class_name": "LDA", "name": "lda", "lda_dim": 3}')
    assert LDA.load_config(str(text))["lda_dim"] == 3
    with pytest.raises(KeyError):
        SPLDA.load(NDX)  # an HDF5 file without a "config" dataset
    with pytest.raises(ValueError):
        PLDAFactory.load_plda("other", NDX)
    assert isinstance(PLDAFactory.create_plda("splda", y_dim=3), SPLDA)
    with pytest.raises(NotImplementedError):
        TransformList.load_params(f, {"name": None, "transforms": {"0": {"class_name": "NAP", "name": "nap"}}})


def test_models_load_from_files_with_the_reference_layout(tmp_path):
    """A TransformList and an SPLDA model laid out as the reference saves them (config JSON string +
    `name/parameter` float datasets, float32 as float_save() makes them), written by the test's own
    fixture writer: configuration handling, name prefixes, float32 -> float64, nested groups."""
    import json

    from h5_fixture import nested, write_h5
    from hyperion_amd.helpers.plda_factory import PLDAFactory
    from hyperion_amd.transforms import LDA, CentWhiten, LNorm, TransformList

    rng = np.random.RandomState(0)
    mu, T = rng.randn(6).astype(np.float32), rng.randn(6, 6).astype(np.float32)
    Tl = rng.randn(6, 3).astype(np.float32)
    config = {"class_name": "TransformList", "name": None, "transforms": {
        "0": {"class_name": "CentWhiten", "name": "cw"}, "1": {"class_name": "LNorm", "name": "ln"},
        "2": {"class_name": "LDA", "name": "lda", "lda_dim": 3, "update_mu": True, "update_t": True}}}
    path = str(tmp_path / "preproc.h5")
    write_h5(path, nested({"config": np.array(json.dumps(config)), "cw/mu": mu, "cw/T": T, "lda/T": Tl}))
    f = Hdf5File(path)
    assert f.keys() == ["config", "cw", "lda"] and np.array_equal(f["cw/T"], T) and f["cw/T"].dtype == np.float32
    t = TransformList.load(path)
    assert [type(x) for x in t.transforms] == [CentWhiten, LNorm, LDA]
    assert t.transforms[0].mu.dtype == np.float64 and np.array_equal(t.transforms[0].T, T.astype(np.float64))
    assert t.transforms[1].mu is None and t.transforms[1].T is None
    assert t.transforms[2].mu is None and t.transforms[2].lda_dim == 3
    assert TransformList.load_config(path)["transforms"]["2"]["lda_dim"] == 3

    V, W = rng.randn(2, 6), np.eye(6) + 0.1 * np.ones((6, 6))
    cfg = {"class_name": "SPLDA", "name": "plda", "y_dim": 2, "update_mu": True, "update_W": True,
           "update_V": False, "fullcov_W": True}
    path = str(tmp_path / "plda.h5")
    write_h5(path, nested({"config": np.array(json.dumps(cfg)), "plda/mu": mu, "plda/V": V, "plda/W": W}))
    m = PLDAFactory.load_plda("splda", path)
    assert m.name == "plda" and m.y_dim == 2 and m.update_V is False and np.array_equal(m.V, V)
    assert m.is_init and m.x_dim == 6


def test_vector_file_with_one_dataset_per_key(tmp_path):
    """The layout of the reference's H5DataWriter (hyperion/io/h5_data_writer.py): every key a
    dataset, keys with "/" nested groups; float32 on disk, float_cpu() out."""
    from h5_fixture import nested, write_h5
    from hyperion_amd.io import RandomAccessDataReaderFactory as F

    rng = np.random.RandomState(1)
    vecs = {"spk1/utt-a": rng.randn(5), "spk1/utt-b": rng.randn(5), "spk2/utt-c": rng.randn(5), "solo": rng.randn(5)}
    path = str(tmp_path / "xvector.h5")
    write_h5(path, nested({k: v.astype(np.float32) for k, v in vecs.items()}))
    r = F.create(path)
    keys = ["solo", "spk2/utt-c", "spk1/utt-a"]
    x = r.read(keys, squeeze=True)
    assert x.shape == (3, 5) and x.dtype == np.float64
    assert np.array_equal(x, np.stack([vecs[k].astype(np.float32) for k in keys]).astype(np.float64))
    assert r.read_dims(keys).tolist() == [5, 5, 5]
    p = F.create("h5,p:" + path)
    y = p.read(["solo", "spk9/none"], squeeze=True)
    assert y.shape == (2, 5) and not y[1].any()
