"""bench.py --gpus N must start N ranks by itself (the driver calls it that way).  The launch path is
run here without a GPU: the ranks form a gloo group, add up one number and stop."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra):
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def test_gpus_2_starts_two_ranks():
    p = run(["--gpus", "2", "--dry-run"], {"HYPMI_BENCH_REHEARSE": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["requested"] == 2 and out["backend"] == "gloo"


def test_fewer_devices_than_ranks_fails():
    import torch

    if torch.cuda.device_count() >= 64:
        return
    p = run(["--gpus", "64", "--dry-run"], {"HYPMI_BENCH_REHEARSE": "0"})
    assert p.returncode != 0
    assert "only" in p.stderr and "GPU" in p.stderr


def test_rank_count_mismatch_fails():
    """A launcher that starts fewer ranks than --gpus asks for is an error, not a smaller run."""
    env = {"HYPMI_BENCH_REHEARSE": "1"}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-run"]
    p = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "process group has 2 rank" in p.stderr
