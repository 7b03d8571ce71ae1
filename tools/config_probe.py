#!/usr/bin/env python
"""Time the multi-session configuration at its full size on the device: FRPLDA and PLDA, 50 000
enrolment models of 3 segments against 200 000 single-segment tests (1e10 scores), every pooling
method, device tensors in and out; beside it the float64 oracle on a slice on the host.

    python tools/config_probe.py [out.json]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import synth
    from hyperion_amd.pdfs.plda import FRPLDA, PLDA
    from oracle import plda_ref as O

    n_models, segs, n2, d, K = 50_000, 3, 200_000, 200, 150
    g = torch.Generator(device="cuda").manual_seed(21)
    x1 = torch.randn((n_models * segs, d), device="cuda", generator=g)
    x1 = x1 * (d ** 0.5) / x1.norm(dim=1, keepdim=True)
    x2 = torch.randn((n2, d), device="cuda", generator=g)
    x2 = x2 * (d ** 0.5) / x2.norm(dim=1, keepdim=True)
    ids1 = np.repeat(np.arange(n_models), segs)
    res = {"config": "%d enrolment models x %d segments vs %d tests, %d-d" % (n_models, segs, n2, d), "runs": []}
    for kind in ("frplda", "plda"):
        m = (synth.make_frplda if kind == "frplda" else synth.make_plda)(d, K)
        model = (FRPLDA if kind == "frplda" else PLDA)(**{k: v for k, v in m.items() if k != "type"})
        for method in ("vavg-lnorm", "vavg", "savg", "book"):
            S = model.llr_Nvs1(x1, x2, ids1=ids1, method=method)  # warm-up: plans, allocator
            del S
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            S = model.llr_Nvs1(x1, x2, ids1=ids1, method=method)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            chk = float(S[:64, :64].sum().item())
            del S
            # the oracle on a slice: 1000 models (3000 segments) x 4000 tests
            a, b = 1000, 4000
            h1, h2 = x1[:a * segs].cpu().numpy().astype(np.float64), x2[:b].cpu().numpy().astype(np.float64)
            c0 = time.perf_counter()
            O.llr_Nvs1(m, h1, h2, ids1=ids1[:a * segs], method=method)
            cdt = time.perf_counter() - c0
            run = {"model": kind, "method": method, "K": int(model.score_plan().k), "ms": dt * 1e3,
                   "scores_per_s": n_models * n2 / dt, "cpu_oracle_scores_per_s": a * b / cdt,
                   "cpu_sample": "%d x %d" % (a, b), "checksum_64x64": chk}
            res["runs"].append(run)
            print(json.dumps(run), flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
