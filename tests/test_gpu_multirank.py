"""The sharded pipeline with real peers: 2 and 3 fresh rank processes (python -m torch.distributed.run,
gloo, all ranks on GPU 0) run PLDABackend.prepare_cohort(rank, world) -- each rank selects the top-k
cohort lists of its own slice of the enrolment rows, one all-gather makes them global -- then
score_snorm on their own slice of the tests, TrialHistogram.all_reduce and ExactPass.all_gather.
The shards are put together here and compared with the unsharded device result and with the float64
oracle.  The partition is the reference's job grid (hyperion/utils/list_utils.py:95-112,
hyperion/utils/trial_ndx.py:252-268: contiguous floor split; egs/voxceleb/v1/steps_be/eval_be_v1.sh:33-60
runs the parts as independent processes and concatenates their score files)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import plda_ref as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import multirank_worker as W  # noqa: E402  (problem definition shared with the rank processes)


def _run_ranks(world, out_dir):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "multirank_worker.py"),
           str(out_dir)]
    # child processes of their own (this process has a GPU context and is never replaced)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, "ranks failed:\n%s\n%s" % (r.stdout[-3000:], r.stderr[-6000:])
    return [np.load(os.path.join(out_dir, "rank%d.npz" % k)) for k in range(world)]


@pytest.fixture(scope="module")
def unsharded():
    import torch

    from hyperion_amd.metrics import TrialHistogram

    chain, m, (xe, ke), (xt, kt), xc = W.problem()
    be = W.make_backend(chain, m)
    coh = be.prepare_cohort(xe, xc)
    got = be.score_snorm(xt, coh)
    ye, yt, yc = (O.chain_predict(a, chain) for a in (xe, xt, xc))
    ref = O.adapt_s_norm(O.llr_1vs1(m, ye, yt), O.llr_1vs1(m, yc, yt), O.llr_1vs1(m, ye, yc), 50, 2)
    e_idx = coh["e_idx"].cpu().numpy()
    return dict(got=got.cpu().numpy(), ref=ref, e_idx=e_idx, ke=ke, kt=kt, torch=torch, hist=TrialHistogram)


def _sets(e_idx, k):
    """[k/8, rows, 8] list layout -> sorted members per row (the order inside a list is a layout matter)."""
    g, rows, _ = e_idx.shape
    flat = e_idx.transpose(1, 0, 2).reshape(rows, g * 8)[:, :k]
    return np.sort(flat.astype(np.int64) & 0xffff, axis=1)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_pipeline_with_real_ranks(world, tmp_path, unsharded):
    from hyperion_amd.backend import shard_range

    u = unsharded
    parts = _run_ranks(world, tmp_path)
    nt = u["got"].shape[1]
    # every rank owns the reference's slice of the test list, and the slices tile it
    assert [(int(p["lo"]), int(p["hi"])) for p in parts] == [shard_range(nt, r, world) for r in range(world)]
    assert int(parts[0]["lo"]) == 0 and int(parts[-1]["hi"]) == nt
    whole = np.hstack([p["scores"] for p in parts])
    assert whole.shape == u["got"].shape
    # against the float64 oracle: the bar of the device path
    err = np.abs(whole.astype(float) - u["ref"]) / np.maximum(np.abs(u["ref"]), 1.0)
    assert err.max() <= 1e-4, err.max()
    # against the unsharded device result: fp32 rounding only (the members of a list are summed in an
    # order that depends on the row's position in its launch, not on anything else)
    d = np.abs(whole - u["got"]) / np.maximum(np.abs(u["got"]), 1.0)
    assert d.max() <= 5e-6, d.max()
    # the gathered enrolment lists: the same on every rank, and the same cohort members as one process
    # selects for the whole enrolment set (53 rows: slices of 26 + 27, or 17 + 18 + 18)
    ref_sets = _sets(u["e_idx"], 50)
    for p in parts:
        assert p["e_idx"].shape == u["e_idx"].shape
        assert np.array_equal(_sets(p["e_idx"], 50), ref_sets)
        assert np.array_equal(p["e_idx"], parts[0]["e_idx"])
    # the all-reduced histogram: identical on every rank, the sum of the ranks' own counts, and equal to
    # the histogram of the assembled matrix
    torch = u["torch"]
    h = u["hist"](*W.HIST).add(torch.from_numpy(whole).cuda(), u["ke"], u["kt"])
    non, tar = h.counts()
    assert int(non.sum() + tar.sum()) == whole.size
    for p in parts:
        assert np.array_equal(p["non"], non) and np.array_equal(p["tar"], tar)
    assert np.array_equal(sum(p["local_non"] for p in parts), non)
    assert np.array_equal(sum(p["local_tar"] for p in parts), tar)
    m = h.mom.cpu().numpy()
    for p in parts:  # moments: float64 sums added in a different order
        assert np.allclose(p["moments"], m, rtol=1e-12, atol=0)
    # the exact second pass (scores of the undecided bins gathered from every rank) gives the reference's
    # EER / minDCF of the assembled scores on every rank
    key = np.asarray(u["ke"])[:, None] == np.asarray(u["kt"])[None, :]
    eer = O.compute_eer(whole[key].astype(float), whole[~key].astype(float))
    dcf = O.compute_min_dcf(whole[key].astype(float), whole[~key].astype(float), 0.05)[0]
    for p in parts:
        assert bool(p["is_exact"])
        assert abs(float(p["eer"]) - eer) <= 1e-12 and abs(float(p["min_dcf"]) - dcf) <= 1e-12
