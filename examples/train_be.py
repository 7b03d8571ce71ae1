#!/usr/bin/env python
"""Training of a PLDA back-end on an MI355X with the command line of the reference's recipe step
(egs/voxceleb/v1/steps_be/train-be-v1.py:32-134): labelled x-vectors in; `lda_lnorm.h5` (PCA when the data are
rank deficient, LDA, centring + length normalisation), `plda.h5` and `elbo.csv` out, the files the evaluation
steps load.

    python examples/train_be.py --iv-file scp:xvector.scp --train-list utt2spk --lda-dim 150 \\
        --plda-type splda --y-dim 120 --epochs 20 --output-path exp/be

The statistics of every step (scatter matrices, class sums, the E-step's accumulators) are taken on the device
in float64; eigendecompositions of d x d matrices stay on the host as in the reference.
"""
import argparse
import logging
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from hyperion_amd.helpers import PLDAFactory as F  # noqa: E402
from hyperion_amd.helpers import VectorClassReader as VCR  # noqa: E402
from hyperion_amd.hyp_defs import config_logger  # noqa: E402
from hyperion_amd.transforms import LDA, PCA, LNorm, TransformList  # noqa: E402


def train_be(iv_file, train_list, lda_dim, plda_type, y_dim, z_dim, epochs, ml_md, md_epochs, output_path,
             **kwargs):
    x, class_ids = VCR(iv_file, train_list, None, **VCR.filter_args(**kwargs)).read()
    t0 = time.time()
    steps = []
    rank = PCA.get_pca_dim_for_var_ratio(x, var_r=1)
    if rank < x.shape[1]:  # rank-deficient data: project on their span first
        pca = PCA(pca_dim=rank, name="pca")
        pca.fit(x)
        x = pca.predict(x)
        lda_dim, y_dim = min(lda_dim, rank), min(y_dim, rank)
        steps.append(pca)
        logging.info("PCA rank=%d", rank)
    lda = LDA(lda_dim=lda_dim, name="lda")
    lda.fit(x, class_ids)
    x = lda.predict(x)
    lnorm = LNorm(name="lnorm")
    lnorm.fit(x)
    x = lnorm.predict(x)
    steps += [lda, lnorm]
    logging.info("pre-processing chain trained in %.2f s", time.time() - t0)
    t0 = time.time()
    plda = F.create_plda(plda_type, y_dim=y_dim, z_dim=z_dim, name="plda")
    elbo = plda.fit(x, class_ids, epochs=epochs, ml_md=ml_md, md_epochs=md_epochs)
    logging.info("PLDA trained in %.2f s", time.time() - t0)
    os.makedirs(output_path, exist_ok=True)
    TransformList(steps).save(output_path + "/lda_lnorm.h5")
    plda.save(output_path + "/plda.h5")
    np.savetxt(output_path + "/elbo.csv", np.vstack((np.arange(epochs), elbo)).T, delimiter=",")


def main(argv=None):
    parser = argparse.ArgumentParser(description="Train Back-end")
    parser.add_argument("--iv-file", dest="iv_file", required=True)
    parser.add_argument("--train-list", dest="train_list", required=True)
    VCR.add_argparse_args(parser)
    F.add_argparse_train_args(parser)
    parser.add_argument("--output-path", dest="output_path", required=True)
    parser.add_argument("--lda-dim", dest="lda_dim", type=int, default=None)
    parser.add_argument("-v", "--verbose", dest="verbose", default=1, choices=[0, 1, 2, 3], type=int)
    args = parser.parse_args(argv)
    config_logger(args.verbose)
    del args.verbose
    train_be(**vars(args))


if __name__ == "__main__":
    main()
