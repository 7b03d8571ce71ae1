"""Throughput of the x-vector loader: n 512-d float32 vectors from a Kaldi archive into one buffer,
keys in archive order and in random order; native threaded reader (hyperion_amd.io) against the
record-at-a-time restatement of the reference's reader (oracle/io_ref.py, CPU baseline), and, on a
GPU box, the page-locked path up to the device tensor.

    python tools/loader_probe.py [n] [dir]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperion_amd.io import ArkDataWriter, RandomAccessArkDataReader  # noqa: E402
from oracle import io_ref  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/loader_probe"
os.makedirs(work, exist_ok=True)
ark, scp = os.path.join(work, "x.ark"), os.path.join(work, "x.scp")
d = 512
keys = ["utt%07d" % i for i in range(n)]
rng = np.random.RandomState(0)
with ArkDataWriter(ark, scp) as w:
    for b in range(0, n, 50_000):
        w.write(keys[b:b + 50_000], rng.randn(min(50_000, n - b), d).astype(np.float32))
mb = n * d * 4 / 1e6
print("%d vectors x %d float32 = %.0f MB in %s" % (n, d, mb, ark), flush=True)
shuffled = list(np.random.RandomState(1).permutation(keys))
for name, order in (("archive order", keys), ("random order", shuffled)):
    for threads in (1, 4, 16):
        r = RandomAccessArkDataReader(scp, num_threads=threads)
        r.read(order[:10], squeeze=True)
        t0 = time.perf_counter()
        x = r.read(order, squeeze=True)
        dt = time.perf_counter() - t0
        print("native  %-13s %2d threads  %7.3f s  %8.0f vectors/ms  %7.2f GB/s" % (name, threads, dt, n / dt / 1e3,
                                                                                     mb / dt / 1e3), flush=True)
    m = min(n, 100_000)
    t0 = time.perf_counter()
    y = io_ref.read_vectors(scp, order[:m])
    dt = time.perf_counter() - t0
    print("oracle  %-13s  1 thread   %7.3f s  %8.0f vectors/ms  %7.2f GB/s  (%d vectors, scp parse included)"
          % (name, dt, m / dt / 1e3, m * d * 4 / 1e9 / dt, m), flush=True)
    assert np.array_equal(x[:m], y)
try:
    import torch

    if torch.cuda.is_available():
        r = RandomAccessArkDataReader(scp, num_threads=16)
        r.read_device(keys[:1000])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        xd = r.read_device(keys)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("native  archive order -> page-locked -> device  %7.3f s  %7.2f GB/s" % (dt, mb / dt / 1e3), flush=True)
        assert bool((xd[:1000].cpu() == torch.from_numpy(r.read(keys[:1000], squeeze=True))).all())
except ImportError:
    pass
