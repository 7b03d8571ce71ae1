"""x-vector loading and trial bookkeeping (the step in front of the scoring path) against fixtures
written and read by the reference itself (tests/golden/trials/, oracle/gen_golden_io.py): archives
from the reference's ArkDataWriter, lists from its save_txt, and expected.npz with what its readers
returned.  Host code only: runs without a GPU (the native loader is a host function of
libhypmi.so)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import io_ref

DIR = os.path.join(GOLDEN, "trials")


@pytest.fixture()
def exp(monkeypatch):
    monkeypatch.chdir(DIR)  # the scripts name their archives relative to the fixture directory
    return np.load(os.path.join(DIR, "expected.npz"), allow_pickle=False)


def test_oracle_restatement_matches_reference(exp):
    assert np.array_equal(io_ref.read_vectors("vectors.scp", list(exp["pick_keys"])), exp["pick_x"])
    assert np.array_equal(io_ref.read_vectors("vectors64.scp", list(exp["all_keys"][:9])), exp["x64"])
    for name, mp, sp in (("full", (1, 1), (1, 1)), ("part_2of3_1of2", (2, 3), (1, 2)), ("part_3of3_2of2", (3, 3), (2, 2))):
        x_e, x_t, info, ms, ss, mask = io_ref.trial_data("vectors.scp", "trials.ndx", "enroll.lst", mp, sp)
        assert np.array_equal(x_e, exp[name + "_x_e"]) and np.array_equal(x_t, exp[name + "_x_t"])
        assert np.array_equal(info, exp[name + "_info"]) and np.array_equal(ms, exp[name + "_model_set"])
        assert np.array_equal(ss, exp[name + "_seg_set"]) and np.array_equal(mask, exp[name + "_trial_mask"])


def test_native_vector_reader_matches_reference(exp):
    from hyperion_amd.io import RandomAccessArkDataReader, RandomAccessDataReaderFactory

    r = RandomAccessDataReaderFactory.create("scp:vectors.scp")
    assert isinstance(r, RandomAccessArkDataReader)
    x = r.read(list(exp["pick_keys"]), squeeze=True)
    assert x.dtype == np.float32 and np.array_equal(x, exp["pick_x"])
    for threads in (1, 3, 16):
        r.num_threads = threads
        assert np.array_equal(r.read(list(exp["all_keys"]), squeeze=True), exp["all_x"])
    x64 = RandomAccessArkDataReader("vectors64.scp").read(list(exp["all_keys"][:9]), squeeze=True)
    assert x64.dtype == np.float64 and np.array_equal(x64, exp["x64"])
    # conversion on the way in: float64 records into a float32 buffer and the reverse, pitched rows
    buf = np.zeros((9, 28), dtype=np.float32)
    RandomAccessArkDataReader("vectors64.scp").read_into(list(exp["all_keys"][:9]), buf[:, :24])
    assert np.array_equal(buf[:, :24], exp["x64"].astype(np.float32)) and not buf[:, 24:].any()
    buf64 = np.zeros((5, 24), dtype=np.float64)
    r.read_into(list(exp["pick_keys"]), buf64)
    assert np.array_equal(buf64, exp["pick_x"].astype(np.float64))
    # un-squeezed reads and single keys go through the host parser
    one = r.read("t03")
    assert isinstance(one, list) and np.array_equal(one[0], exp["pick_x"][0])
    assert r.read_shapes(["t03", "c05"]) == [(24,), (24,)]
    assert list(r.read_dims(["t03", "c05"])) == [24, 24]


def test_reader_errors_and_permissive(exp, tmp_path):
    from hyperion_amd._lib import HypmiError
    from hyperion_amd.io import RandomAccessArkDataReader, RandomAccessDataReaderFactory

    r = RandomAccessArkDataReader("vectors.scp")
    with pytest.raises(Exception, match="Key nope not found"):
        r.read(["t03", "nope"], squeeze=True)
    p = RandomAccessDataReaderFactory.create("scp,p:vectors.scp")
    got = p.read(["t03", "nope", "t29"], squeeze=True)  # unknown key: a row of zeros, as the reference
    assert got.shape == (3, 24) and not got[1].any() and np.array_equal(got[0], exp["pick_x"][0])
    with pytest.raises(ValueError, match="needs a script file"):
        RandomAccessDataReaderFactory.create("ark:vectors.ark")
    # a script that points into the middle of a record, past the end, or at another dimension
    lines = open("vectors.scp").read().split("\n")
    for bad, msg in ((lines[0].replace(":4", ":9"), "not a binary record"),
                     (lines[0].replace(":4", ":999999"), "beyond the end"),
                     ("x " + os.path.join(DIR, "mats.ark") + ":2", "more than one row|different dimension")):
        scp = tmp_path / "bad.scp"
        first = lines[1].replace("./", DIR + "/")
        scp.write_text(first + "\n" + bad.replace("./", DIR + "/") + "\n")
        rb = RandomAccessArkDataReader(str(scp))
        with pytest.raises(HypmiError, match=msg):
            rb.read(list(rb.keys), squeeze=True)


def test_matrix_records_and_ranges(exp):
    from hyperion_amd.io import RandomAccessArkDataReader

    r = RandomAccessArkDataReader("mats_ranges.scp")
    for k, m in zip("abc", r.read(["a", "b", "c"])):
        assert np.array_equal(m, exp["mat_" + k])
    assert np.array_equal(r.read(["b"], row_offset=1, num_rows=2)[0], exp["mat_b_rows_1_2"])
    assert np.array_equal(r.read(["a"], row_offset=1, num_rows=2)[0], exp["mat_a_rows_1_2"])
    assert np.array_equal(np.array([list(s) for s in r.read_shapes(["a", "b", "c"])]), exp["mat_shapes"])


def test_writer_round_trip_is_byte_identical(exp, tmp_path):
    """Our writer produces the reference writer's bytes and script for the same records."""
    from hyperion_amd.io import ArkDataWriter, RandomAccessArkDataReader

    keys = [line.split()[0] for line in open("vectors.scp")]
    x = RandomAccessArkDataReader("vectors.scp").read(keys, squeeze=True)
    ark, scp = str(tmp_path / "v.ark"), str(tmp_path / "v.scp")
    with ArkDataWriter(ark, scp) as w:
        w.write(keys[:10], x[:10])
        w.write(keys[10:], list(x[10:]))
    assert open(ark, "rb").read() == open("vectors.ark", "rb").read()
    assert open(scp).read().replace(ark, "./vectors.ark") == open("vectors.scp").read()
    mats = RandomAccessArkDataReader("mats.scp").read(["a", "b", "c"])
    with ArkDataWriter(str(tmp_path / "m.ark")) as w:
        w.write(["a", "b", "c"], mats)
    assert open(str(tmp_path / "m.ark"), "rb").read() == open("mats.ark", "rb").read()


@pytest.mark.parametrize("name,kw", [
    ("full", {}),
    ("part_2of3_1of2", dict(model_part_idx=2, num_model_parts=3, seg_part_idx=1, num_seg_parts=2)),
    ("part_3of3_2of2", dict(model_part_idx=3, num_model_parts=3, seg_part_idx=2, num_seg_parts=2)),
    ("from_key", {}),
    ("coh_test", dict(eval_set="coh-test")),
])
def test_trial_data_reader_matches_reference(exp, name, kw):
    from hyperion_amd.helpers import TrialDataReader

    ndx_file = "trials.key" if name == "from_key" else "trials.ndx"
    enroll_file = "cohort.lst" if name == "coh_test" else "enroll.lst"
    tdr = TrialDataReader("scp:vectors.scp", ndx_file, enroll_file, **kw)
    x_e, x_t, info, ndx = tdr.read()
    assert np.array_equal(x_e, exp[name + "_x_e"]) and np.array_equal(x_t, exp[name + "_x_t"])
    assert np.array_equal(np.asarray(info, dtype=str), exp[name + "_info"])
    assert np.array_equal(np.asarray(tdr.enroll.key, dtype=str), exp[name + "_enroll_key"])
    assert np.array_equal(ndx.model_set, exp[name + "_model_set"])
    assert np.array_equal(ndx.seg_set, exp[name + "_seg_set"])
    assert np.array_equal(ndx.trial_mask, exp[name + "_trial_mask"])


def test_vector_reader_matches_reference(exp):
    from hyperion_amd.helpers import VectorReader

    assert np.array_equal(VectorReader("scp:vectors.scp", "cohort.lst").read(), exp["vector_reader_x"])


def test_key_scores_and_ndx_bookkeeping(exp, tmp_path):
    from hyperion_amd.utils import TrialKey, TrialNdx, TrialScores

    key = TrialKey.load("trials.key")
    assert np.array_equal(key.model_set, exp["key_model_set"]) and np.array_equal(key.seg_set, exp["key_seg_set"])
    assert np.array_equal(key.tar, exp["key_tar"]) and np.array_equal(key.non, exp["key_non"])
    key.save(str(tmp_path / "k.txt"))
    assert open(str(tmp_path / "k.txt")).read() == open("trials.key").read()
    ndx = TrialNdx.load("trials.ndx")
    ndx.save(str(tmp_path / "n.txt"))
    assert open(str(tmp_path / "n.txt")).read() == open("trials.ndx").read()
    assert key.to_ndx() == ndx
    scr = TrialScores.load("trials.scores")
    assert np.array_equal(scr.scores, exp["scores_loaded"]) and np.array_equal(scr.score_mask, exp["scores_mask_loaded"])
    scr.save(str(tmp_path / "s.txt"))
    assert open(str(tmp_path / "s.txt")).read() == open("trials.scores").read()
    tar, non = scr.get_tar_non(key)
    # the fixture's scores went through "%f": compare with the rounded originals
    assert np.allclose(tar, exp["scores_tar"], atol=1e-6) and np.allclose(non, exp["scores_non"], atol=1e-6)
    assert tar.shape == exp["scores_tar"].shape and non.shape == exp["scores_non"].shape
    assert np.allclose(scr.split(2, 3, 1, 2).scores, np.where(scr.split(2, 3, 1, 2).score_mask, exp["scores_split"], 0),
                       atol=1e-6)
    assert np.array_equal(key.filter(key.model_set[3:7], key.seg_set[5:11]).tar, exp["key_filter_tar"])
    merged = TrialNdx.merge([ndx.split(i + 1, 2, j + 1, 3) for i in range(2) for j in range(3)])
    assert merged == ndx and np.array_equal(merged.trial_mask, exp["ndx_merged_mask"])
    m2 = TrialScores.merge([scr.split(i + 1, 2, j + 1, 3) for i in range(2) for j in range(3)])
    assert m2 == scr
    with pytest.raises(AssertionError):
        TrialScores.merge([scr, scr])


def test_class_keys_of_a_block_key():
    from hyperion_amd.utils import TrialKey

    cls_m, cls_s = np.array([0, 1, 1, 2, 5]), np.array([1, 1, 0, 3, 2, 0, 7])
    tar = cls_m[:, None] == cls_s[None, :]
    key = TrialKey(["m%d" % i for i in range(5)], ["s%d" % i for i in range(7)], tar, ~tar)
    km, ks = key.class_keys()
    assert np.array_equal(km[:, None] == ks[None, :], tar) and km.min() >= 0 and ks.min() >= 0
    key.non[0, 0] = False  # a hole in the trial matrix: no class description
    assert key.class_keys() is None


def test_keys_spread_over_several_archives(tmp_path):
    """One script over three archives, keys asked for in an order that hops between them."""
    from hyperion_amd.io import ArkDataWriter, RandomAccessArkDataReader

    rng = np.random.RandomState(8)
    x = rng.randn(90, 12).astype(np.float32)
    keys = ["k%03d" % i for i in range(90)]
    lines = []
    for a in range(3):
        ark, scp = str(tmp_path / ("p%d.ark" % a)), str(tmp_path / ("p%d.scp" % a))
        with ArkDataWriter(ark, scp) as w:
            w.write(keys[a::3], x[a::3])
        lines += open(scp).read().splitlines()
    rng.shuffle(lines)
    (tmp_path / "all.scp").write_text("\n".join(lines) + "\n")
    r = RandomAccessArkDataReader(str(tmp_path / "all.scp"), num_threads=4)
    order = rng.permutation(90)
    assert np.array_equal(r.read([keys[i] for i in order], squeeze=True), x[order])
    assert np.array_equal(r.read(keys, squeeze=True), x)


VCR_CASES = {
    "plain": dict(),
    "min3": dict(min_spc=3),
    "max4_random": dict(min_spc=2, max_spc=4, spc_pruning_mode="random", vcr_seed=7),
    "max4_last": dict(max_spc=4, spc_pruning_mode="last"),
    "max4_first": dict(max_spc=4, spc_pruning_mode="first"),
    "split3_seq": dict(csplit_max_spc=3, csplit_mode="sequential", csplit_min_spc=2),
    "split4_seq_ov1": dict(csplit_max_spc=4, csplit_mode="sequential", csplit_overlap=1),
    "split3_random": dict(csplit_max_spc=3, csplit_mode="random", vcr_seed=11),
    "split_1part": dict(csplit_max_spc=3, csplit_mode="random_1part", csplit_min_spc=2, vcr_seed=3),
    "class2int": dict(class2int_file="train.class2int", min_spc=2),
    "twice": dict(csplit_max_spc=3, csplit_mode="random", csplit_once=False, vcr_seed=5),
}


@pytest.mark.parametrize("name", sorted(VCR_CASES))
def test_vector_class_reader_matches_reference(exp, name):
    """The labelled-vector loader of the training recipes: vectors, class numbers, kept keys and
    class sizes under every pruning / class-splitting option, as the reference's VectorClassReader
    returns them for the same seed (expected_vcr.npz by oracle/gen_golden_io.py gen_vcr)."""
    from hyperion_amd.helpers import VectorClassReader

    g = np.load(os.path.join(DIR, "expected_vcr.npz"))
    v = VectorClassReader("scp:vectors.scp", "train.utt2spk", **VCR_CASES[name])
    x, ids = v.read()
    assert x.dtype == g[name + "_x"].dtype and np.array_equal(x, g[name + "_x"]) and np.array_equal(ids, g[name + "_ids"])
    assert v.u2c.key.tolist() == g[name + "_keys"].tolist()
    assert np.array_equal(v.samples_per_class, g[name + "_spc"])
    assert np.asarray(v.class_names, dtype=str).tolist() == g[name + "_names"].tolist()
    if name == "twice":
        x2, ids2 = v.read()
        assert np.array_equal(x2, g[name + "_x2"]) and np.array_equal(ids2, g[name + "_ids2"])
    if name == "min3":
        x3, w3 = v.read(return_3d=True)
        assert np.array_equal(x3, g[name + "_x3d"]) and np.array_equal(w3, g[name + "_w3d"])
        assert v.max_samples_per_class == g[name + "_spc"].max()
    assert VectorClassReader.filter_args(min_spc=2, other=1) == {"min_spc": 2}
