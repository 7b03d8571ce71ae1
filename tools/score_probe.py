#!/usr/bin/env python
"""Time hypmi_score alone (HIP events on torch's stream) for a given shape; HYPMI_SCORE_VARIANT picks
an experimental kernel variant when the library was built with them."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hyperion_amd import _lib, engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n1", type=int, default=100000)
ap.add_argument("--n2", type=int, default=100000)
ap.add_argument("--kp", type=int, default=152)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--variants", default="0")
a = ap.parse_args()
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
G1 = torch.randn((a.n1, a.kp), device="cuda", generator=g)
G2 = torch.randn((a.n2, a.kp), device="cuda", generator=g)
S = torch.empty((a.n1, engine.round_up(a.n2, 32)), device="cuda")[:, :a.n2]
ref = None
for v in a.variants.split(","):
    os.environ["HYPMI_SCORE_VARIANT"] = v
    engine.score(G1, G2, out=S)
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        engine.score(G1, G2, out=S)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    chk = float(S[::997, ::991].double().sum())
    if ref is None:
        ref = chk
    ms = min(ts)
    tf = 2.0 * (a.kp - 2) * a.n1 * a.n2 / ms / 1e9
    print("variant %s: min %.3f ms  median %.3f ms  -> %.1f TF (K=kp-2)  %.1f%% of 157.3  checksum %s"
          % (v, ms, sorted(ts)[len(ts) // 2], tf, tf / 1.573, "ok" if abs(chk - ref) <= 1e-6 * abs(ref) else "DIFF %r %r" % (chk, ref)))
