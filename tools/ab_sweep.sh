#!/bin/bash
# gather stage times of the AS-norm pipeline for several builds and cohort sizes in one GPU call:
# tools/ab_sweep.sh "<lib> <lib> ..." "<cohort> <cohort> ..."
for NC in $2; do
  for L in $1; do
    HYPMI_LIB=$L timeout -k 10 200 python - $L $NC <<'PY' 2>/dev/null | tail -n 1
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from hyperion_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
from hyperion_amd.pdfs.plda import SPLDA
from hyperion_amd.transforms import LDA, CentWhiten, LNorm, TransformList
cfg = dict(bench.WORKLOADS["asnorm_10k_128k"]); cfg["n_coh"] = int(sys.argv[2]); cfg["test_chunk"] = 131072
chain, model = bench.build_workload(cfg)
plda = SPLDA(mu=model["mu"], V=model["V"], W=model["W"])
pre = TransformList([CentWhiten(mu=chain[0][1], T=chain[0][2], name="cw"), LNorm(name="ln"), LDA(mu=chain[2][1], T=chain[2][2], name="lda")])
dt, st, be, bufs = bench.asnorm_run(plda, pre, cfg, 3, 1, 0, 1, None)
print(os.path.basename(sys.argv[1]), "cohort", sys.argv[2], "ms/step %.1f" % (dt * 1e3), {k: round(v, 1) for k, v in st["per_step_ms"].items() if "gather" in k or "topk" in k})
PY
  done
done
