#!/usr/bin/env python
"""PLDA evaluation with adaptive S-norm on an MI355X, with the command line of the reference's recipe step
(egs/voices_challenge/v1/steps_be/eval-be-snorm-v1.py:28-132): x-vector archive, trial index, enrolment
list, pre-processing chain and PLDA model files in; a `model segment score` text file out.

    python examples/eval_be_snorm.py --iv-file scp:xvector.scp --ndx-file trials.ndx \\
        --enroll-file enroll.lst --preproc-file lda_lnorm.h5 --model-file plda.h5 --plda-type splda \\
        --coh-iv-file scp:xvector_coh.scp --coh-list cohort.lst --coh-nbest 300 --score-file scores.txt

Two flows give the same scores:

  --flow device (default)  raw vectors are read onto the GPU, the chain is folded into the embedding
                           prologue and PLDABackend keeps the three score matrices, the cohort selection and
                           the normalisation in HBM; only the normalised scores come back.
  --flow arrays            the statements of the reference script, unchanged, on the classes of this package:
                           numpy arrays between the steps (three score matrices cross the host link twice).

`--seg-part-idx i --num-seg-parts n` (the reference's job grid) evaluates one shard of the test segments;
one such job per GPU covers a node.
"""
import argparse
import logging
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from hyperion_amd import engine  # noqa: E402
from hyperion_amd.backend import PLDABackend  # noqa: E402
from hyperion_amd.helpers import PLDAFactory as F  # noqa: E402
from hyperion_amd.helpers import TrialDataReader as TDR  # noqa: E402
from hyperion_amd.helpers import VectorReader as VR  # noqa: E402
from hyperion_amd.hyp_defs import config_logger, float_cpu  # noqa: E402
from hyperion_amd.score_norm import AdaptSNorm  # noqa: E402
from hyperion_amd.transforms import TransformList  # noqa: E402
from hyperion_amd.utils.trial_scores import TrialScores  # noqa: E402


def scores_through_arrays(tdr_args, preproc, model, coh_iv_file, coh_list, nbest, nbest_discard):
    """The reference's sequence: every step takes and returns host arrays."""
    x_e, x_t, enroll, ndx = TDR(*tdr_args[0], preproc, **tdr_args[1]).read()
    x_coh = VR(coh_iv_file, coh_list, preproc).read()
    scores = model.llr_1vs1(x_e, x_t)
    scores_coh_test = model.llr_1vs1(x_coh, x_t)
    scores_enr_coh = model.llr_1vs1(x_e, x_coh)
    snorm = AdaptSNorm(nbest=nbest, nbest_discard=nbest_discard)
    return snorm.predict(scores, scores_coh_test, scores_enr_coh), enroll, ndx


def scores_on_device(tdr_args, preproc, model, coh_iv_file, coh_list, nbest, nbest_discard):
    """Raw vectors to the GPU once; chain, scores, cohort selection and normalisation stay there."""
    x_e, x_t, enroll, ndx = TDR(*tdr_args[0], None, **tdr_args[1]).read_device()
    x_coh = VR(coh_iv_file, coh_list).read_device()
    backend = PLDABackend(model, preproc, nbest=nbest, nbest_discard=nbest_discard)
    cohort = backend.prepare_cohort(x_e, x_coh)
    scores = backend.score_snorm(x_t, cohort)
    return engine.to_host(scores, float_cpu()), enroll, ndx


def eval_plda(iv_file, ndx_file, enroll_file, test_file, preproc_file, coh_iv_file, coh_list, coh_nbest,
              coh_nbest_discard, model_file, score_file, plda_type, flow="device", **kwargs):
    preproc = TransformList.load(preproc_file) if preproc_file is not None else None
    logging.info("loading plda model: %s", model_file)
    model = F.load_plda(plda_type, model_file)
    # (the reference's filter_args drops the part indices, trial_data_reader.py:75-84: the names are taken here)
    grid = ("tlist_sep", "model_part_idx", "num_model_parts", "seg_part_idx", "num_seg_parts", "eval_set")
    tdr_args = ((iv_file, ndx_file, enroll_file, test_file), {k: kwargs[k] for k in grid if k in kwargs})
    run = scores_on_device if flow == "device" else scores_through_arrays
    t0 = time.time()
    scores, enroll, ndx = run(tdr_args, preproc, model, coh_iv_file, coh_list, coh_nbest, coh_nbest_discard)
    dt = time.time() - t0
    logging.info("%d x %d trials read, scored and normalised in %.2f s (%s flow)", scores.shape[0],
                 scores.shape[1], dt, flow)
    logging.info("saving scores to %s", score_file)
    TrialScores(enroll, ndx.seg_set, scores).save_txt(score_file)


def main(argv=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                     fromfile_prefix_chars="@", description="Eval PLDA with S-Norm")
    parser.add_argument("--iv-file", dest="iv_file", required=True)
    parser.add_argument("--ndx-file", dest="ndx_file", default=None)
    parser.add_argument("--enroll-file", dest="enroll_file", required=True)
    parser.add_argument("--test-file", dest="test_file", default=None)
    parser.add_argument("--preproc-file", dest="preproc_file", default=None)
    parser.add_argument("--coh-iv-file", dest="coh_iv_file", required=True)
    parser.add_argument("--coh-list", dest="coh_list", required=True)
    parser.add_argument("--coh-nbest", dest="coh_nbest", type=int, default=100)
    parser.add_argument("--coh-nbest-discard", dest="coh_nbest_discard", type=int, default=0)
    TDR.add_argparse_args(parser)
    F.add_argparse_eval_args(parser)
    parser.add_argument("--score-file", dest="score_file", required=True)
    parser.add_argument("--flow", default="device", choices=["device", "arrays"])
    parser.add_argument("-v", "--verbose", dest="verbose", default=1, choices=[0, 1, 2, 3], type=int)
    args = parser.parse_args(argv)
    config_logger(args.verbose)
    del args.verbose
    assert args.test_file is not None or args.ndx_file is not None
    eval_plda(**vars(args))


if __name__ == "__main__":
    main()
