"""One-rank (or more) check that the RCCL process group the benchmark uses comes up on this machine and
that the collectives of the sharded path run on it: barrier, MAX all-reduce (step time), SUM all-reduce
(histogram sink), all_gather_into_tensor (top-k lists).  Launch with torch.distributed.run."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.distributed as dist

from hyperion_amd.backend import all_gather_lists  # noqa: E402


def main():
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(lr)
    dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
    w, r = dist.get_world_size(), dist.get_rank()
    dist.barrier()
    t = torch.tensor([1.0 + r], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == float(w)
    h = torch.full((2, 4096), r + 1, dtype=torch.int64, device="cuda")
    dist.all_reduce(h)
    assert int(h[0, 0].item()) == w * (w + 1) // 2
    counts = [8 * (3 + i) for i in range(w)]  # rank i contributes 3 + i rows of 8-entry groups
    local = torch.full((38, counts[r]), r, dtype=torch.int16, device="cuda")
    if w > 1:
        g = all_gather_lists(local, counts)
    else:  # one rank: all_gather_lists returns its input; run the collective it would issue
        g = torch.empty_like(local)
        dist.all_gather_into_tensor(g.view(torch.uint8), local.view(torch.uint8))
    assert g.shape == (38, sum(counts)) and int(g[-1, -1].item()) == w - 1
    torch.cuda.synchronize()
    dist.barrier()
    if r == 0:
        print("rccl ok: world %d" % w)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
