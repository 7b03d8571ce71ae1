#!/usr/bin/env python
"""Time PLDA training (statistics + EM epochs) on the device beside the float64 oracle on the host.

    python tools/train_probe.py [n_vectors] [n_classes] [dim] [y_dim] [epochs] [splda|frplda|plda]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from hyperion_amd.pdfs.plda import FRPLDA, PLDA, SPLDA
    from oracle import plda_ref as O

    kind = "splda"
    if sys.argv[1:] and not sys.argv[-1].isdigit():
        kind = sys.argv.pop()
    a = [int(v) for v in sys.argv[1:]]
    n, ncls, d, K, epochs = (a + [200_000, 5000, 200, 150, 10][len(a):])[:5]
    rng = np.random.RandomState(3)
    Vt = rng.randn(K, d) * np.geomspace(1.0, 0.1, K)[:, None]
    y = rng.randint(0, ncls, n)
    y[:ncls] = np.arange(ncls)
    x = rng.randn(ncls, K)[y] @ Vt + rng.randn(n, d) + 0.3
    z_dim = max(1, K // 3)
    new = {"splda": lambda: SPLDA(y_dim=K), "frplda": lambda: FRPLDA(), "plda": lambda: PLDA(y_dim=K, z_dim=z_dim)}[kind]
    ref_fit = {"splda": lambda D: O.splda_fit(D, K, epochs=epochs), "frplda": lambda D: O.frplda_fit(D, epochs=epochs),
               "plda": lambda D: O.plda_fit(D, K, z_dim, epochs=epochs)}[kind]
    new().fit(x[:4000], class_ids=y[:4000] % 1000, epochs=1)  # warm-up: library, allocator
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    D = SPLDA._stats_hard_device(x, y)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    m = new()
    m.initialize(D)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    elbo = []
    for _ in range(epochs):
        st = m.Estep(D)
        elbo.append(m.elbo(st))
        m.MstepML(st)
        m.MstepMD(st)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    te = time.perf_counter()
    for _ in range(epochs):
        st = m.Estep(D)
    torch.cuda.synchronize()
    te = (time.perf_counter() - te) / epochs
    print(kind, "n %d classes %d dim %d y_dim %d" % (n, ncls, d, K))
    print("device: stats (upload + class sums + x^T x) %.3f s, first model %.3f s, %d epochs %.3f s "
          "(%.1f ms per epoch, E-step alone %.1f ms)" % (t1 - t0, t2 - t1, epochs, t3 - t2,
                                                         (t3 - t2) / epochs * 1e3, te * 1e3))
    c0 = time.perf_counter()
    Dh = O.compute_stats_hard(x, y)
    c1 = time.perf_counter()
    ref, ref_elbo = ref_fit(Dh)
    c2 = time.perf_counter()
    print("host oracle: stats %.3f s, first model + %d epochs %.3f s (%.1f ms per epoch)"
          % (c1 - c0, epochs, c2 - c1, (c2 - c1) / epochs * 1e3))
    print("elbo max rel diff %.2e" % np.max(np.abs(np.array(elbo) / ref_elbo - 1)))


if __name__ == "__main__":
    main()
