"""One rank of the sharded chain -> SPLDA -> adaptive S-norm pipeline, started by
tests/test_gpu_multirank.py through `python -m torch.distributed.run` (never imported by pytest).

Every rank shares GPU 0 and talks over gloo: the flow is the one an N-GPU node runs over RCCL
(PLDABackend.prepare_cohort(rank, world) with its one all-gather of top-k lists, score_snorm on the
rank's own slice of the tests, TrialHistogram.all_reduce, ExactPass.all_gather), with real peers.
The partition is the reference's job grid: contiguous floor split of the test list
(hyperion/utils/list_utils.py:95-112 through hyperion/utils/trial_ndx.py:252-268).

Usage: multirank_worker.py OUT_DIR  (RANK / WORLD_SIZE / MASTER_* from the launcher).  Writes
OUT_DIR/rank<r>.npz."""
import faulthandler
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def problem():
    """The small problem every rank (and the test itself) builds from the same seeds: 53 enrolment
    models (uneven slices for 2 and 3 ranks), 333 tests, 400 cohort vectors, 96-d chain to 40-d."""
    import synth

    chain = synth.chain_cfg2(d=96, lda_dim=40, seed=3)
    m = synth.make_splda(40, 12)
    xe, ke = synth.raw_xvectors(53, d=96, seed=21, num_centers=30)
    xt, kt = synth.raw_xvectors(333, d=96, seed=22, num_centers=30)
    xc, _ = synth.raw_xvectors(400, d=96, seed=23, num_centers=200)
    return chain, m, (xe, ke), (xt, kt), xc


def make_backend(chain, m, test_chunk=100):
    from hyperion_amd.backend import PLDABackend
    from hyperion_amd.pdfs.plda import SPLDA
    from hyperion_amd.transforms import LDA, CentWhiten, LNorm, TransformList

    tl = TransformList([CentWhiten(mu=chain[0][1], T=chain[0][2], name="cw"), LNorm(name="ln"),
                        LDA(mu=chain[2][1], T=chain[2][2], name="lda")])
    return PLDABackend(SPLDA(mu=m["mu"], V=m["V"], W=m["W"]), tl, nbest=50, nbest_discard=2, test_chunk=test_chunk)


HIST = (-12.0, 12.0, 256)


def main():
    faulthandler.dump_traceback_later(150, exit=True)  # a hung collective must not outlive the test
    out_dir = sys.argv[1]
    import torch
    import torch.distributed as dist

    from hyperion_amd.backend import shard_range
    from hyperion_amd.metrics import TrialHistogram

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    try:
        assert dist.get_world_size() == world
        chain, m, (xe, ke), (xt, kt), xc = problem()
        be = make_backend(chain, m)
        coh = be.prepare_cohort(xe, xc, rank=rank, world=world)  # top-k of this rank's enrolment slice + all-gather
        lo, hi = shard_range(len(xt), rank, world)
        out = be.score_snorm(xt[lo:hi], coh)
        h = TrialHistogram(*HIST)
        h.add(out, ke, kt[lo:hi])
        local_non, local_tar = h.counts()
        h.all_reduce()
        non, tar = h.counts()
        exact = h.exact_pass()  # the bins that can hold a hull vertex: their scores from every rank
        n_bins = 0
        if exact is not None:
            n_bins = len(exact.bins)
            exact.add(out, ke, kt[lo:hi])
            exact.all_gather()
            exact.finish()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), lo=lo, hi=hi, scores=out.cpu().numpy(),
                 e_idx=coh["e_idx"].cpu().numpy(), non=non, tar=tar, local_non=local_non, local_tar=local_tar,
                 eer=h.eer(), min_dcf=h.min_dcf(0.05)[0], is_exact=h.is_exact(), n_bins=n_bins,
                 moments=h.mom.cpu().numpy(), topk=np.array(be.topk_report()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
