#!/usr/bin/env python
"""Where one benchmark step spends its time: HIP events around every call of the step, in the same
order as bench.py issues them, next to the wall time of the whole step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import synth  # noqa: E402
from hyperion_amd import engine  # noqa: E402
from hyperion_amd.engine import SIDE_ENROL, SIDE_TEST  # noqa: E402
from hyperion_amd.pdfs.plda import SPLDA  # noqa: E402
from hyperion_amd.transforms import LDA, CentWhiten, LNorm, TransformList  # noqa: E402

n = 100_000
chain = synth.chain_cfg2(d=512, lda_dim=200)
model = synth.make_splda(200, 150)
plda = SPLDA(mu=model["mu"], V=model["V"], W=model["W"])
pre = TransformList([CentWhiten(mu=chain[0][1], T=chain[0][2], name="cw"), LNorm(name="ln"),
                     LDA(mu=chain[2][1], T=chain[2][2], name="lda")])
plan = plda.score_plan(pre)
g = torch.Generator(device="cuda").manual_seed(0)
x1 = torch.randn((n, 512), device="cuda", generator=g)
x2 = torch.randn((n, 512), device="cuda", generator=g)
S = torch.empty((n, n), device="cuda")
names = ["prologue_enrol", "prologue_test", "split_enrol", "split_test", "score"]


def step(evs):
    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs.append(e)
    ev()
    G1 = plan.prologue(x1, SIDE_ENROL); ev()
    G2 = plan.prologue(x2, SIDE_TEST); ev()
    H1 = engine.split_rows(G1); ev()
    H2 = engine.split_rows(G2); ev()
    engine.score(H1, H2, out=S); ev()


for _ in range(2):
    step([])
torch.cuda.synchronize()
all_evs = []
t0 = time.perf_counter()
for _ in range(5):
    evs = []
    step(evs)
    all_evs.append(evs)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 5 * 1e3
for i, nm in enumerate(names):
    print("%-16s %8.3f ms" % (nm, sum(e[i].elapsed_time(e[i + 1]) for e in all_evs) / 5))
print("%-16s %8.3f ms  (sum of the above %.3f)" % ("wall per step", wall,
      sum(e[0].elapsed_time(e[-1]) for e in all_evs) / 5))
