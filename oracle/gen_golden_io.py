"""Generates tests/golden/trials/: small vector archives, trial lists and the outputs of the
REFERENCE's readers on them.  In-container only (imports /root/reference through oracle/load_ref.py);
the fixtures it writes are data: archives and text lists produced by the reference's own writers,
and .npz files with what its readers return.

    python -B oracle/gen_golden_io.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "trials")


def load_reference_io():
    from oracle import load_ref

    load_ref.load()
    io = types.ModuleType("hyperion.io")
    io.__path__ = [load_ref.REF_ROOT + "/hyperion/io"]
    sys.modules["hyperion.io"] = io
    from hyperion.io.data_rw_factory import RandomAccessDataReaderFactory

    io.RandomAccessDataReaderFactory = RandomAccessDataReaderFactory
    hp = types.ModuleType("hyperion.helpers")
    hp.__path__ = [load_ref.REF_ROOT + "/hyperion/helpers"]
    sys.modules["hyperion.helpers"] = hp
    from hyperion import hyp_defs
    from hyperion.helpers.trial_data_reader import TrialDataReader
    from hyperion.helpers.vector_reader import VectorReader
    from hyperion.io.ark_data_reader import RandomAccessArkDataReader
    from hyperion.io.ark_data_writer import ArkDataWriter
    from hyperion.utils.scp_list import SCPList
    from hyperion.utils.trial_key import TrialKey
    from hyperion.utils.trial_ndx import TrialNdx
    from hyperion.utils.trial_scores import TrialScores
    from hyperion.utils.utt2info import Utt2Info

    return types.SimpleNamespace(**locals())


def gen_vcr():
    """Adds the labelled-vector loader's fixtures to tests/golden/trials/ without touching the others:
    a `key class` list over the vectors of vectors.scp with classes of 1 to 9 vectors, and what the
    reference's VectorClassReader returns for it under every pruning / splitting option."""
    R = load_reference_io()
    from hyperion.helpers.vector_class_reader import VectorClassReader

    os.chdir(OUT)
    with open("vectors.scp") as f:
        keys = sorted(line.split()[0] for line in f)
    rng = np.random.RandomState(5)
    sizes = [9, 8, 7, 6, 5, 4, 3, 2, 1]
    assert sum(sizes) <= len(keys)
    order = rng.permutation(len(keys))[:sum(sizes)]
    cls = np.repeat(["spk%d" % i for i in range(len(sizes))], sizes)
    cls = cls[rng.permutation(len(cls))]
    with open("train.utt2spk", "w") as f:
        for i, c in zip(order, cls):
            f.write("%s %s\n" % (keys[i], c))
    with open("train.class2int", "w") as f:
        for i in range(len(sizes)):
            f.write("spk%d %d\n" % (i, len(sizes) - 1 - i))
    cases = {
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
    out = {}
    for name, kw in cases.items():
        v = VectorClassReader("scp:vectors.scp", "train.utt2spk", **kw)
        x, ids = v.read()
        out[name + "_x"], out[name + "_ids"] = x, ids
        out[name + "_keys"] = np.asarray(v.u2c.key, dtype=str)
        out[name + "_spc"] = np.asarray(v.samples_per_class)
        out[name + "_names"] = np.asarray(v.class_names, dtype=str)
        if name == "twice":  # a new split on every read
            x2, ids2 = v.read()
            out[name + "_x2"], out[name + "_ids2"] = x2, ids2
        if name == "min3":
            x3, w3 = v.read(return_3d=True)
            out[name + "_x3d"], out[name + "_w3d"] = x3, w3
    np.savez_compressed("expected_vcr.npz", **out)
    print("vcr fixtures ok:", {k: out[k + "_x"].shape[0] for k in cases})


def main():
    R = load_reference_io()
    os.makedirs(OUT, exist_ok=True)
    os.chdir(OUT)  # archive paths inside the scripts stay relative to the fixture directory
    rng = np.random.RandomState(77)
    dim = 24
    models = ["m%02d" % i for i in range(10)]
    enrol_segs, enrol_model = [], []
    for i, m in enumerate(models):
        for j in range(1 + i % 3):
            enrol_segs.append("e%02d_%d" % (i, j))
            enrol_model.append(m)
    tests = ["t%02d" % i for i in range(30)]
    cohort = ["c%02d" % i for i in range(12)]
    keys = enrol_segs + tests + cohort
    order = rng.permutation(len(keys))  # archive order differs from every list order
    x = rng.randn(len(keys), dim)
    with R.ArkDataWriter("./vectors.ark", "./vectors.scp") as w:  # float32 vector records
        w.write([keys[i] for i in order], [x[i] for i in order])
    R.hyp_defs.set_float_save("float64")
    with R.ArkDataWriter("./vectors64.ark", "./vectors64.scp") as w:  # float64 vector records
        w.write(keys[:9], [x[i] for i in range(9)])
    R.hyp_defs.set_float_save("float32")
    mats = [rng.randn(n, 5) for n in (7, 3, 11)]
    with R.ArkDataWriter("./mats.ark", "./mats.scp") as w:
        w.write(["a", "b", "c"], mats)
    with open("mats.scp") as f:
        lines = f.read().split("\n")
    with open("mats_ranges.scp", "w") as f:  # row ranges on two records
        f.write(lines[0] + "[2:5]\n" + lines[1] + "\n" + lines[2] + "[4:]\n")

    with open("enroll.lst", "w") as f:
        for s, m in zip(enrol_segs, enrol_model):
            f.write("%s %s\n" % (s, m))
    with open("cohort.lst", "w") as f:
        for c in cohort:
            f.write("%s %s\n" % (c, c))
    mask = rng.rand(len(models), len(tests)) < 0.6
    mask[0, :] = True
    mask[:, 0] = True
    ndx = R.TrialNdx(models, tests, mask)
    ndx.save_txt("trials.ndx")
    tar = np.logical_and(mask, rng.rand(*mask.shape) < 0.2)
    key = R.TrialKey(models, tests, tar, np.logical_and(mask, ~tar))
    key.save_txt("trials.key")

    out = {}
    # readers
    r = R.RandomAccessArkDataReader("vectors.scp")
    pick = [tests[3], enrol_segs[0], cohort[5], tests[29], tests[3]]
    out["pick_keys"] = np.array(pick)
    out["pick_x"] = r.read(pick, squeeze=True)
    out["all_keys"] = np.array(keys)
    out["all_x"] = r.read(keys, squeeze=True)
    r64 = R.RandomAccessArkDataReader("vectors64.scp")
    out["x64"] = r64.read(keys[:9], squeeze=True)
    rm = R.RandomAccessArkDataReader("mats_ranges.scp")
    for k, m in zip("abc", rm.read(["a", "b", "c"])):
        out["mat_" + k] = m
    out["mat_b_rows_1_2"] = rm.read(["b"], row_offset=1, num_rows=2)[0]
    out["mat_a_rows_1_2"] = rm.read(["a"], row_offset=1, num_rows=2)[0]
    out["mat_shapes"] = np.array([list(s) for s in rm.read_shapes(["a", "b", "c"])])
    # trial data reader over the job grid and the evaluation subsets that run in the reference
    cases = {"full": dict(), "part_2of3_1of2": dict(model_part_idx=2, num_model_parts=3, seg_part_idx=1,
                                                   num_seg_parts=2),
             "part_3of3_2of2": dict(model_part_idx=3, num_model_parts=3, seg_part_idx=2, num_seg_parts=2),
             "from_key": dict(), "coh_test": dict(eval_set="coh-test")}
    for name, kw in cases.items():
        ndx_file = "trials.key" if name == "from_key" else "trials.ndx"
        enroll_file = "cohort.lst" if name == "coh_test" else "enroll.lst"
        tdr = R.TrialDataReader("scp:vectors.scp", ndx_file, enroll_file, **kw)
        x_e, x_t, info, nd = tdr.read()
        out[name + "_x_e"], out[name + "_x_t"] = x_e, x_t
        out[name + "_info"] = np.asarray(info, dtype=str)
        out[name + "_enroll_key"] = np.asarray(tdr.enroll.key, dtype=str)
        out[name + "_model_set"] = np.asarray(nd.model_set, dtype=str)
        out[name + "_seg_set"] = np.asarray(nd.seg_set, dtype=str)
        out[name + "_trial_mask"] = nd.trial_mask
    vr = R.VectorReader("scp:vectors.scp", "cohort.lst")
    out["vector_reader_x"] = vr.read()
    # key / ndx / scores bookkeeping
    k2 = R.TrialKey.load("trials.key")
    out["key_model_set"], out["key_seg_set"] = np.asarray(k2.model_set, dtype=str), np.asarray(k2.seg_set, dtype=str)
    out["key_tar"], out["key_non"] = k2.tar, k2.non
    scores = rng.randn(len(models), len(tests)) * 5
    scr = R.TrialScores(models, tests, scores, mask)
    scr.save_txt("trials.scores")
    s2 = R.TrialScores.load_txt("trials.scores")
    out["scores_loaded"], out["scores_mask_loaded"] = s2.scores, s2.score_mask
    t, n = scr.get_tar_non(k2)
    out["scores_tar"], out["scores_non"] = t, n
    sub = scr.split(2, 3, 1, 2)
    out["scores_split"] = sub.scores
    flt = k2.filter(models[3:7], tests[5:11])
    out["key_filter_tar"] = flt.tar
    merged = R.TrialNdx.merge([ndx.split(i + 1, 2, j + 1, 3) for i in range(2) for j in range(3)])
    out["ndx_merged_mask"] = merged.trial_mask
    np.savez(os.path.join(OUT, "expected.npz"), **out)
    print("wrote", OUT, sorted(os.listdir(OUT)))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "vcr":
    gen_vcr()
elif __name__ == "__main__":
    main()
