#!/usr/bin/env python
"""Stage times of the AS-norm pipeline with the trial scores launched before or after the selection of
the tests' cohort lists (PLDABackend.trials_before_topk), alternating inside one process.

    python tools/order_probe.py [n_test]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hyperion_amd.backend import PLDABackend  # noqa: E402
from hyperion_amd.pdfs.plda import SPLDA  # noqa: E402
from hyperion_amd.transforms import LDA, CentWhiten, LNorm, TransformList  # noqa: E402

cfg = dict(bench.WORKLOADS["asnorm_10k_1m"])
cfg["n_test"] = int(sys.argv[1]) if len(sys.argv) > 1 else 393216
chain, model = bench.build_workload(cfg)
plda = SPLDA(mu=model["mu"], V=model["V"], W=model["W"])
preproc = TransformList([CentWhiten(mu=chain[0][1], T=chain[0][2], name="cw"), LNorm(name="ln"),
                         LDA(mu=chain[2][1], T=chain[2][2], name="lda")])
_init = PLDABackend.__init__
for first in (True, False, True, False):
    def init(self, *a, _first=first, **kw):
        _init(self, *a, **kw)
        self.trials_before_topk = _first
    PLDABackend.__init__ = init
    dt, st, be, bufs = bench.asnorm_run(plda, preproc, cfg, 4, 2, 0, 1, None)
    print("trials before top-k: %-5s ms/step %.1f" % (first, dt * 1e3),
          {k: round(x, 1) for k, x in st["per_step_ms"].items()}, flush=True)
    del be, bufs
