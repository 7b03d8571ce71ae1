#!/usr/bin/env python
"""Stage timing of the AS-norm pipeline on one GPU (HIP events on torch's current stream)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import synth  # noqa: E402
from hyperion_amd import _lib, engine  # noqa: E402
from hyperion_amd.engine import SIDE_ENROL, SIDE_TEST  # noqa: E402
from hyperion_amd.pdfs.plda import SPLDA  # noqa: E402
from hyperion_amd.score_norm.adapt_s_norm import gather_stats, topk_rows  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ne", type=int, default=10000)
ap.add_argument("--nt", type=int, default=65536)
ap.add_argument("--nc", type=int, default=20000)
ap.add_argument("--k", type=int, default=300)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()

m = synth.make_splda(200, 150)
plda = SPLDA(mu=m["mu"], V=m["V"], W=m["W"])
plan = plda.score_plan()
g = torch.Generator(device="cuda").manual_seed(1)


def vecs(n):
    x = torch.randn((n, 200), device="cuda", generator=g)
    return x * (200 ** 0.5) / x.norm(dim=1, keepdim=True)


xe, xt, xc = vecs(a.ne), vecs(a.nt), vecs(a.nc)
lib = _lib.load()


class T:
    def __init__(self):
        self.acc = {}

    def run(self, name, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        torch.cuda.synchronize()
        self.acc.setdefault(name, []).append(e0.elapsed_time(e1))
        return r


t = T()
for rep in range(a.reps):
    Ge = t.run("prologue_enrol", lambda: plan.prologue(xe, SIDE_ENROL))
    Gc = t.run("prologue_cohort", lambda: plan.prologue(xc, SIDE_TEST))
    A_ec = t.run("score_ec", lambda: engine.score(Ge, Gc))
    e_idx = t.run("topk_ec", lambda: topk_rows(A_ec, a.k, 0))
    Gtr = t.run("prologue_test_rows", lambda: plan.prologue(xt, SIDE_ENROL))
    Gtc = t.run("prologue_test_cols", lambda: plan.prologue(xt, SIDE_TEST))
    A_tc = t.run("score_tc", lambda: engine.score(Gtr, Gc))
    t_idx = t.run("topk_tc", lambda: topk_rows(A_tc, a.k, 0))
    mz, iz, ldz = t.run("gather_z", lambda: gather_stats(A_ec, None, t_idx, a.nt, 1e-5, a.k))
    mt, it, ldt = t.run("gather_t", lambda: gather_stats(A_tc, None, e_idx, a.ne, 1e-5, a.k))
    S = t.run("score_et", lambda: engine.score(Ge, Gtc))
    out = torch.empty_like(S)

    def app():
        _lib.check(lib.hypmi_asnorm_apply(engine._ptr(S), S.stride(0), a.ne, a.nt, engine._ptr(mz),
                                          engine._ptr(iz), ldz, engine._ptr(mt), engine._ptr(it), ldt,
                                          engine._ptr(out), out.stride(0), engine.stream_ptr()))
    t.run("apply", app)
import ctypes  # noqa: E402

redo = ctypes.c_ulonglong(0)
_lib.check(lib.hypmi_gather_redo_count(ctypes.byref(redo), 1))
print("gather pairs redone around a list pivot: %d of %d (all repetitions)" % (redo.value, 2 * a.reps * a.ne * a.nt))
tot = 0.0
for k, v in t.acc.items():
    print("%-22s %10.3f ms" % (k, v[-1]))
    tot += v[-1]
per_test = ["prologue_test_rows", "prologue_test_cols", "score_tc", "topk_tc", "gather_z", "gather_t", "score_et", "apply"]
pt = sum(t.acc[k][-1] for k in per_test)
print("total %.1f ms; per-test-chunk stages %.1f ms -> %.3f G normalised scores/s" % (tot, pt, a.ne * a.nt / pt / 1e6))
