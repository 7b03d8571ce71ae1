"""The S-norm evaluation step of the recipes (egs/voices_challenge/v1/steps_be/eval-be-snorm-v1.py) end to
end through examples/eval_be_snorm.py: archive, trial index, lists and model files in, a score file out, in
both flows, against the oracle on the vectors the reference's own reader returned
(tests/golden/trials/expected.npz)."""
import os
import sys

import numpy as np
import pytest

import synth
from conftest import GOLDEN, ROOT
from oracle import plda_ref as O

pytestmark = pytest.mark.gpu
DIR = os.path.join(GOLDEN, "trials")


def _files(tmp_path):
    from hyperion_amd.pdfs.plda import SPLDA
    from hyperion_amd.transforms import LDA, CentWhiten, LNorm, TransformList

    rng = np.random.RandomState(4)
    chain = [("cw", 0.1 * rng.randn(24), rng.randn(24, 24) / 5), ("lnorm", None, None),
             ("lda", 0.1 * rng.randn(24), rng.randn(24, 16) / 5)]
    m = synth.make_splda(16, 6)
    # the files hold float32 parameters: the oracle gets the same rounded values
    chain = [(k, None if mu is None else mu.astype(np.float32).astype(float),
              None if T is None else T.astype(np.float32).astype(float)) for k, mu, T in chain]
    m = {k: (v.astype(np.float32).astype(float) if isinstance(v, np.ndarray) else v) for k, v in m.items()}
    TransformList([CentWhiten(mu=chain[0][1], T=chain[0][2], name="cw"), LNorm(name="ln"),
                   LDA(mu=chain[2][1], T=chain[2][2], name="lda")]).save(str(tmp_path / "pre.h5"))
    SPLDA(mu=m["mu"], V=m["V"], W=m["W"], name="plda").save(str(tmp_path / "plda.h5"))
    # one enrolment segment per model (the score table is indexed by model name)
    seen, lines = set(), []
    for line in open(os.path.join(DIR, "enroll.lst")):
        seg, model = line.split()
        if model not in seen:
            seen.add(model)
            lines.append(line)
    (tmp_path / "enroll1.lst").write_text("".join(lines))
    return m, chain, [ln.split()[0] for ln in lines]


@pytest.mark.parametrize("flow", ["device", "arrays"])
def test_eval_be_snorm_script(tmp_path, monkeypatch, flow):
    from hyperion_amd.utils.trial_scores import TrialScores

    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import eval_be_snorm

    m, chain, enrol_keys = _files(tmp_path)
    monkeypatch.chdir(DIR)
    out = str(tmp_path / ("scores_%s.txt" % flow))
    eval_be_snorm.main(["--iv-file", "scp:vectors.scp", "--ndx-file", "trials.ndx", "--enroll-file",
                        str(tmp_path / "enroll1.lst"), "--preproc-file", str(tmp_path / "pre.h5"), "--model-file",
                        str(tmp_path / "plda.h5"), "--plda-type", "splda", "--coh-iv-file", "scp:vectors.scp",
                        "--coh-list", "cohort.lst", "--coh-nbest", "5", "--coh-nbest-discard", "1",
                        "--score-file", out, "--flow", flow, "-v", "0"])
    g = np.load("expected.npz", allow_pickle=False)
    keys = list(g["all_keys"])
    x = {k: g["all_x"][i].astype(float) for i, k in enumerate(keys)}
    seg_set = list(g["full_seg_set"])
    pre = lambda ks: O.chain_predict(np.stack([x[k] for k in ks]), chain)  # noqa: E731
    ye, yt, yc = pre(enrol_keys), pre(seg_set), pre([ln.split()[0] for ln in open("cohort.lst")])
    ref = O.adapt_s_norm(O.llr_1vs1(m, ye, yt), O.llr_1vs1(m, yc, yt), O.llr_1vs1(m, ye, yc), 5, 1)
    got = TrialScores.load_txt(out)
    assert got.score_mask.all() and got.scores.shape == ref.shape
    models = [ln.split()[1] for ln in open(str(tmp_path / "enroll1.lst"))]
    order_m = [models.index(mm) for mm in got.model_set]
    order_s = [seg_set.index(ss) for ss in got.seg_set]
    want = ref[np.ix_(order_m, order_s)]
    # the text file keeps six decimals
    assert np.max(np.abs(got.scores - want) / np.maximum(np.abs(want), 1.0)) <= 1e-4 + 1e-6


@pytest.mark.parametrize("pool", ["book", "vavg", "vavg-lnorm", "savg"])
def test_eval_plda_script_n_vs_1(tmp_path, monkeypatch, pool):
    """examples/eval_plda.py with multi-session models (19 enrolment segments of 10 models) into an HDF5 score
    file, against the oracle's llr_Nvs1; and one block of the job grid."""
    from hyperion_amd.utils.trial_scores import TrialScores

    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import eval_plda

    m, chain, _ = _files(tmp_path)
    monkeypatch.chdir(DIR)
    common = ["--iv-file", "scp:vectors.scp", "--ndx-file", "trials.ndx", "--enroll-file", "enroll.lst",
              "--preproc-file", str(tmp_path / "pre.h5"), "--model-file", str(tmp_path / "plda.h5"),
              "--plda-type", "splda", "--pool-method", pool, "-v", "0"]
    out = str(tmp_path / "scores.h5")
    eval_plda.main(common + ["--score-file", out])
    g = np.load("expected.npz", allow_pickle=False)
    ye = O.chain_predict(g["full_x_e"].astype(float), chain)
    yt = O.chain_predict(g["full_x_t"].astype(float), chain)
    models, ids = np.unique(g["full_info"], return_inverse=True)
    ref = O.llr_Nvs1(m, ye, yt, ids1=ids, method=pool)
    got = TrialScores.load(out)
    assert list(got.model_set) == list(models) and list(got.seg_set) == list(g["full_seg_set"])
    assert np.max(np.abs(got.scores - ref) / np.maximum(np.abs(ref), 1.0)) <= 1e-4
    # block (2 of 3 models, 1 of 2 segments) of the job grid
    part = str(tmp_path / "part.h5")
    eval_plda.main(common + ["--score-file", part, "--model-part-idx", "2", "--num-model-parts", "3",
                             "--seg-part-idx", "1", "--num-seg-parts", "2"])
    blk = TrialScores.load(part)
    assert list(blk.model_set) == list(g["part_2of3_1of2_model_set"])
    assert list(blk.seg_set) == list(g["part_2of3_1of2_seg_set"])
    rows = [list(got.model_set).index(x) for x in blk.model_set]
    cols = [list(got.seg_set).index(x) for x in blk.seg_set]
    assert np.allclose(blk.scores, got.scores[np.ix_(rows, cols)], rtol=0, atol=2e-5 * np.abs(got.scores).max())


def test_train_be_script_then_eval(tmp_path, monkeypatch):
    """examples/train_be.py (egs/voxceleb/v1/steps_be/train-be-v1.py) writes the chain, the model and the ELBO
    table; the chain and model it saves equal the same steps made by hand, and examples/eval_plda.py scores
    with the files."""
    from hyperion_amd.helpers import PLDAFactory, VectorClassReader
    from hyperion_amd.transforms import LDA, LNorm, TransformList
    from hyperion_amd.utils.trial_scores import TrialScores

    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import eval_plda
    import train_be

    monkeypatch.chdir(DIR)
    out = str(tmp_path / "be")
    train_be.main(["--iv-file", "scp:vectors.scp", "--train-list", "train.utt2spk", "--lda-dim", "6",
                   "--plda-type", "splda", "--y-dim", "3", "--epochs", "5", "--min-spc", "3",
                   "--output-path", out, "-v", "0"])
    elbo = np.loadtxt(out + "/elbo.csv", delimiter=",")
    assert elbo.shape == (5, 3) and np.array_equal(elbo[:, 0], np.arange(5))
    assert np.all(np.diff(elbo[:, 1]) > -1e-6 * abs(elbo[0, 1]))
    # the same steps by hand
    x, ids = VectorClassReader("scp:vectors.scp", "train.utt2spk", min_spc=3).read()
    lda = LDA(lda_dim=6, name="lda")
    lda.fit(x, ids)
    lnorm = LNorm(name="lnorm")
    lnorm.fit(lda.predict(x))
    y = lnorm.predict(lda.predict(x))
    plda = PLDAFactory.create_plda("splda", y_dim=3, name="plda")
    ref_elbo, _ = plda.fit(y, ids, epochs=5)
    assert np.allclose(elbo[:, 1], ref_elbo, rtol=1e-12)
    pre = TransformList.load(out + "/lda_lnorm.h5")
    assert np.allclose(pre.predict(x), y, rtol=1e-3, atol=1e-4)  # the files hold float32 parameters
    scores = str(tmp_path / "scores.h5")
    eval_plda.main(["--iv-file", "scp:vectors.scp", "--ndx-file", "trials.ndx", "--enroll-file", "enroll.lst",
                    "--preproc-file", out + "/lda_lnorm.h5", "--model-file", out + "/plda.h5", "--plda-type",
                    "splda", "--pool-method", "vavg-lnorm", "--score-file", scores, "-v", "0"])
    got = TrialScores.load(scores)
    g = np.load("expected.npz", allow_pickle=False)
    models, mid = np.unique(g["full_info"], return_inverse=True)
    want = plda.llr_Nvs1(lnorm.predict(lda.predict(g["full_x_e"].astype(float))),
                         lnorm.predict(lda.predict(g["full_x_t"].astype(float))), ids1=mid, method="vavg-lnorm")
    assert got.scores.shape == want.shape
    assert np.allclose(got.scores, want, rtol=2e-3, atol=2e-3)
