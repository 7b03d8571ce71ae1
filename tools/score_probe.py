#!/usr/bin/env python
"""Time the two score kernels alone (HIP events on torch's stream) for a given shape and compare
their rounding error against float64 on a slab of the matrix."""
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
ap.add_argument("--scale", type=float, default=1.0)
a = ap.parse_args()
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
G1 = torch.randn((a.n1, a.kp), device="cuda", generator=g) * a.scale
G2 = torch.randn((a.n2, a.kp), device="cuda", generator=g) * a.scale
# columns shaped like real packed rows: a large q against a one
G1[:, a.kp - 2] = torch.randn(a.n1, device="cuda", generator=g) * 300.0
G1[:, a.kp - 1] = 1.0
G2[:, a.kp - 2] = 1.0
G2[:, a.kp - 1] = torch.randn(a.n2, device="cuda", generator=g) * 300.0
S = torch.empty((a.n1, engine.round_up(a.n2, 32)), device="cuda")[:, :a.n2]
rows = slice(0, min(a.n1, 517))
cols = slice(max(a.n2 - 1031, 0), a.n2)
ref = G1[rows].double() @ G2[cols].double().T
bound = (G1[rows].double().abs() @ G2[cols].double().abs().T)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), sorted(ts)[len(ts) // 2]


def report(name, ms, med):
    err = (S[rows, cols].double() - ref).abs()
    tf = 2.0 * a.kp * a.n1 * a.n2 / ms / 1e9
    print("%-10s min %.3f ms  median %.3f ms  %.1f fp32-equivalent TF  %.2f TB/s stored | max err %.3e  "
          "max err/sum|ab| %.3e (fp32 unit 5.96e-8)  rms %.3e"
          % (name, ms, med, tf, 4.0 * a.n1 * a.n2 / ms / 1e9, float(err.max()),
             float((err / bound).max()), float(err.pow(2).mean().sqrt())))


ms, med = timed(lambda: engine.score(G1, G2, out=S, split=False))
report("fp32", ms, med)
if lib.hypmi_split_row_bytes(a.kp):
    for fmt, label in (("triple", "bf16x3"), ("pair", "fp16x2")):
        S.zero_()
        H1, H2 = engine.split_rows(G1, fmt), engine.split_rows(G2, fmt)
        ms, med = timed(lambda: engine.score(H1, H2, out=S))
        report(label, ms, med)
        ms, med = timed(lambda: (engine.split_rows(G1, fmt), engine.split_rows(G2, fmt)))
        print("%s conversion of both sides: %.3f ms" % (label, ms))
