#!/usr/bin/env python
"""Stage times of the AS-norm pipeline with a given build of libhypmi.so, for A/B comparisons.

    python tools/ab_lib.py <path to a libhypmi.so> [n_test [nbest]]

MI355X boxes differ by up to 12 % in every stage, so two builds are only comparable inside one job:
build both libraries here, keep them side by side, and run this script once per library, alternating,
in one GPU call (each run is its own process: a process binds to the first library it loads).
Workload: the benchmark's chain + SPLDA + adaptive S-norm, 10 000 enrolment models, 20 000 cohort
members, top 300, n_test tests (default 196 608 = three chunks), four timed steps."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyperion_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import bench  # noqa: E402
from hyperion_amd.pdfs.plda import SPLDA  # noqa: E402
from hyperion_amd.transforms import LDA, CentWhiten, LNorm, TransformList  # noqa: E402

cfg = dict(bench.WORKLOADS["asnorm_10k_128k"])
cfg["n_test"] = int(sys.argv[2]) if len(sys.argv) > 2 else 196608
if len(sys.argv) > 3:
    cfg["nbest"] = int(sys.argv[3])
if os.environ.get("HYPMI_TEST_CHUNK"):
    cfg["test_chunk"] = int(os.environ["HYPMI_TEST_CHUNK"])
chain, model = bench.build_workload(cfg)
plda = SPLDA(mu=model["mu"], V=model["V"], W=model["W"])
preproc = TransformList([CentWhiten(mu=chain[0][1], T=chain[0][2], name="cw"), LNorm(name="ln"),
                         LDA(mu=chain[2][1], T=chain[2][2], name="lda")])
if os.environ.get("HYPMI_PAIR_STATS") == "0":  # the T-part statistics in two arrays, as before the paired layout
    from hyperion_amd.backend import PLDABackend

    _init = PLDABackend.__init__

    def _init_unpaired(self, *a, **kw):
        _init(self, *a, **kw)
        self.pair_stats = False

    PLDABackend.__init__ = _init_unpaired
dt, st, be, bufs = bench.asnorm_run(plda, preproc, cfg, 4, 2, 0, 1, None)
print(os.path.basename(sys.argv[1]), "ms/step %.1f" % (dt * 1e3),
      {k: round(x, 1) for k, x in st["per_step_ms"].items()}, flush=True)
