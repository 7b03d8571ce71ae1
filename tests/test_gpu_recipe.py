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
