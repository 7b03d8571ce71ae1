"""Multi-GPU host logic on CPU: the test-row partition (same contiguous floor split as the
reference's split_list) and the one collective of the pipeline (all-gather of per-enrolment
top-k lists), exercised with world_size 2 and 3 over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hyperion_amd.backend import all_gather_lists, shard_range


def ref_split(n, idx, num_parts):
    """hyperion/utils/list_utils.py:108-111 with 1-based idx."""
    i1 = int(np.floor((idx - 1) * float(n) / num_parts))
    i2 = int(np.floor(idx * float(n) / num_parts))
    return i1, i2


@pytest.mark.parametrize("n", [0, 1, 7, 8, 1000, 1_000_000, 4874])
@pytest.mark.parametrize("parts", [1, 2, 3, 8])
def test_shard_range_matches_split_list(n, parts):
    covered = []
    for p in range(parts):
        lo, hi = shard_range(n, p, parts)
        assert (lo, hi) == ref_split(n, p + 1, parts)
        covered.extend(range(lo, hi) if n <= 1000 else [lo, hi])
    if n <= 1000:
        assert covered == list(range(n))  # disjoint, ordered, complete
    else:
        assert covered[0] == 0 and covered[-1] == n


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(k * n_rows, dtype=torch.int32).reshape(k, n_rows)
        counts = [shard_range(n_rows, r, world)[1] - shard_range(n_rows, r, world)[0] for r in range(world)]
        lo, hi = shard_range(n_rows, rank, world)
        got = all_gather_lists(full[:, lo:hi].contiguous(), counts)
        ok = bool(torch.equal(got, full))
        # the dtype of the real top-k lists (16-bit, which RCCL cannot name: they travel as bytes)
        full16 = (full % 30011).to(torch.int16)
        ok = ok and bool(torch.equal(all_gather_lists(full16[:, lo:hi].contiguous(), counts), full16))
        if rank == 0:  # a solo caller inside the job (one entry in counts) must not start a collective
            ok = ok and all_gather_lists(full16, [n_rows]) is full16
        q.put((rank, ok, tuple(got.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rows", [(2, 11), (3, 10), (2, 1)])
def test_all_gather_lists_gloo(world, n_rows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rows, 5, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok, (rank, shape)
        assert shape == (5, n_rows)


def test_all_gather_lists_single_process():
    x = torch.arange(12, dtype=torch.int32).reshape(3, 4)
    assert all_gather_lists(x, [4]) is x
