#!/usr/bin/env python
"""PLDA evaluation of a trial list on an MI355X with the command lines of the reference's two evaluation
programs (hyperion/bin/eval-plda-1vs1.py:28-108 and eval-plda-nvs1.py:25-95): x-vector archive, trial index,
enrolment list, pre-processing chain and model files in, a score file (text or HDF5, by extension) out.

    python examples/eval_plda.py --iv-file scp:xvector.scp --ndx-file trials.ndx --enroll-file enroll.lst \\
        --preproc-file lda_lnorm.h5 --model-file plda.h5 --plda-type splda --score-file scores.h5
    python examples/eval_plda.py ... --pool-method vavg-lnorm      # models enrolled with several segments

Without --pool-method every enrolment segment is scored on its own (llr_1vs1) and the rows of the score table
are the enrolment list's model names, one segment per model; with it the segments of a model are pooled
(llr_Nvs1: book, vavg, vavg-lnorm or savg).  The raw vectors are read onto the GPU and the chain is folded
into the embedding prologue.  The job-grid options (--model-part-idx / --num-model-parts, --seg-part-idx /
--num-seg-parts) select one block of the trial matrix, e.g. one shard of test segments per GPU.
"""
import argparse
import logging
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from hyperion_amd import engine  # noqa: E402
from hyperion_amd.helpers import PLDAFactory as F  # noqa: E402
from hyperion_amd.helpers import TrialDataReader as TDR  # noqa: E402
from hyperion_amd.hyp_defs import config_logger, float_cpu  # noqa: E402
from hyperion_amd.transforms import TransformList  # noqa: E402
from hyperion_amd.utils.trial_scores import TrialScores  # noqa: E402


def eval_plda(iv_file, ndx_file, enroll_file, test_file, preproc_file, model_file, score_file, plda_type,
              pool_method=None, **kwargs):
    preproc = TransformList.load(preproc_file) if preproc_file is not None else None
    # (the reference's filter_args drops the part indices, trial_data_reader.py:75-84: the names are taken here)
    grid = ("tlist_sep", "model_part_idx", "num_model_parts", "seg_part_idx", "num_seg_parts", "eval_set")
    tdr = TDR(iv_file, ndx_file, enroll_file, test_file, None, **{k: kwargs[k] for k in grid if k in kwargs})
    x_e, x_t, enroll, ndx = tdr.read_device()
    model = F.load_plda(plda_type, model_file)
    t0 = time.time()
    if preproc is not None:
        x_e, x_t = preproc.predict(x_e), preproc.predict(x_t)  # device tensors in, device tensors out
    if pool_method is None:
        scores = model.llr_1vs1(x_e, x_t)
    else:
        enroll, ids_e = np.unique(enroll, return_inverse=True)
        scores = model.llr_Nvs1(x_e, x_t, method=pool_method, ids1=ids_e)
    scores = engine.to_host(scores, float_cpu())
    dt = time.time() - t0
    logging.info("%d x %d trials in %.3f s", scores.shape[0], scores.shape[1], dt)
    TrialScores(enroll, ndx.seg_set, scores).save(score_file)


def main(argv=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                     fromfile_prefix_chars="@", description="Eval PLDA")
    parser.add_argument("--iv-file", dest="iv_file", required=True)
    parser.add_argument("--ndx-file", dest="ndx_file", default=None)
    parser.add_argument("--enroll-file", dest="enroll_file", required=True)
    parser.add_argument("--test-file", dest="test_file", default=None)
    parser.add_argument("--preproc-file", dest="preproc_file", default=None)
    TDR.add_argparse_args(parser)
    F.add_argparse_eval_args(parser)
    parser.add_argument("--pool-method", dest="pool_method", type=str.lower, default=None,
                        choices=["book", "vavg", "vavg-lnorm", "savg"],
                        help="pool the enrolment segments of a model (N vs 1); default: 1 vs 1")
    parser.add_argument("--score-file", dest="score_file", required=True)
    parser.add_argument("-v", "--verbose", dest="verbose", default=1, choices=[0, 1, 2, 3], type=int)
    args = parser.parse_args(argv)
    config_logger(args.verbose)
    del args.verbose
    assert args.test_file is not None or args.ndx_file is not None
    eval_plda(**vars(args))


if __name__ == "__main__":
    main()
