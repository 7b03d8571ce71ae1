#!/usr/bin/env python
"""Benchmark of the PLDA back-end scoring hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload asnorm_10k_1m|...]

Default workload: the configuration the headline metric is quoted on -- SPLDA scores with adaptive
S-norm, 10 000 enrolment models x 1 000 000 test x-vectors per GPU against a 20 000-member cohort,
top-300 (egs/voices_challenge/v1/steps_be/eval-be-snorm-v1.py:56-83 of the reference: three llr_1vs1
calls and AdaptSNorm.predict).  One "step" = one pass of that path over synthetic 512-d x-vectors
already resident in HBM: pre-processing chain + PLDA projection of enrolment, cohort and test
vectors, the three score matrices, both top-k selections, both cohort-statistics gathers and the
normalisation, the [n_enrol, n_test] result left in HBM.

--gpus N > 1 without a torch.distributed environment starts N ranks itself (one process per GPU,
RCCL); every rank scores the same enrolment set against its own shard of tests (per-GPU work is
fixed: weak scaling) and the reported value is what all ranks produced divided by the slowest
rank's time.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "PLDA trial scores/sec (512-d, SPLDA+AS-norm) at 1/8 GPU; % fp32 MFMA peak"
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X dense fp32 matrix peak (256 CU x 256 flop/clk x 2.4 GHz)
F16_MFMA_PEAK_TFLOPS = 2516.6  # dense fp16 / bf16 matrix peak (256 CU x 4096 flop/clk x 2.4 GHz)
HBM_PEAK_GBS = 8000.0
LDS_PEAK_GBS = 256 * 256 * 2.4  # GB/s: 256 CUs x 256 bytes per clock (ds_read_b64 / b128) x 2.4 GHz

WORKLOADS = {
    # n_enrol, n_test per GPU, input dim, LDA dim, y_dim; asnorm: cohort size and nbest
    # (131 072 tests per chunk: 1 % faster than the backend's default of 65 536, tools/chunk_sweep.sh; the chunk's
    # intermediates take 26 GB instead of 13)
    "asnorm_10k_1m": dict(n_enrol=10_000, n_test=1_000_000, d=512, lda=200, K=150, n_coh=20_000, nbest=300,
                          test_chunk=131072),
    "asnorm_10k_128k": dict(n_enrol=10_000, n_test=131_072, d=512, lda=200, K=150, n_coh=20_000, nbest=300),
    # one GPU's share of the 1M x 1M matrix on 8 GPUs (125 000 tests per GPU: `--gpus 8` is the whole
    # job); the normalised scores are never kept, every block goes into target / non-target histograms
    "asnorm_1m_125k": dict(n_enrol=1_000_000, n_test=125_000, d=512, lda=200, K=150, n_coh=20_000, nbest=300,
                           sink=True, test_chunk=32768),  # (16 384: 6.74 s per warm step, 32 768: 6.63, 65 536: 6.60
    # -- but a cold first step, which is what the default line's stage times, pays 1.4 s for its larger buffers)
    "asnorm_100k_16k": dict(n_enrol=100_000, n_test=16_384, d=512, lda=200, K=150, n_coh=20_000, nbest=300,
                            sink=True, test_chunk=32768),
    "asnorm_tiny": dict(n_enrol=300, n_test=2_000, d=512, lda=200, K=150, n_coh=1_000, nbest=100),
    # raw scores only (no normalisation): chain + SPLDA, full score matrix
    "chain_splda_100k": dict(n_enrol=100_000, n_test=100_000, d=512, lda=200, K=150),
    "chain_splda_20k": dict(n_enrol=20_000, n_test=20_000, d=512, lda=200, K=150),
    "splda_vox1": dict(n_enrol=4874, n_test=4874, d=512, lda=None, K=150),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="asnorm_10k_1m", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-stages", action="store_true",
                    help="skip the extra stage measurements (100k x 100k score matrix) of the default workload")
    ap.add_argument("--cpu-sample", type=int, default=20000,
                    help="rows/cols of the square slice the raw-score CPU baseline is timed on")
    ap.add_argument("--split-format", default="pair", choices=["pair", "triple", "none"],
                    help="operands of the score kernel: scaled fp16 pairs, exact bf16 triples, or fp32")
    ap.add_argument("--test-chunk", type=int, default=None,
                    help="tests per chunk of the AS-norm pipeline (default: the workload's own, else 65536)")
    ap.add_argument("--n-coh", type=int, default=None, help="cohort size of an AS-norm workload (sweeps)")
    ap.add_argument("--n-test", type=int, default=None, help="tests per GPU (sweeps)")
    ap.add_argument("--dry-run", action="store_true",
                    help="start the ranks, form the process group, exchange one number, print the rank "
                         "count and stop (no GPU work): a check of the launch path")
    return ap.parse_args(argv)


# ---- launching N ranks -------------------------------------------------------------------------
def rehearsing():
    """HYPMI_BENCH_REHEARSE=1: every rank on GPU 0 with the gloo backend -- a rehearsal of the multi-rank
    flow on a one-GPU (or, with --dry-run, no-GPU) machine; its throughput means nothing."""
    return os.environ.get("HYPMI_BENCH_REHEARSE", "0") == "1"


def launch_ranks(args):
    """Parent of a `--gpus N` run that was not started by a launcher: starts N fresh rank processes
    (python -m torch.distributed.run, one per GPU) before anything here has touched a GPU, and
    returns their exit code.  Rank 0 of the children prints the JSON line."""
    import socket
    import subprocess

    if not rehearsing():
        import torch

        have = torch.cuda.device_count()  # counting devices does not initialise the GPU
        if have < args.gpus:
            print("bench.py: --gpus %d but only %d GPU(s) are visible" % (args.gpus, have), file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def init_ranks(args):
    """(world, rank, dist) of this process; forms the process group when there is more than one rank and
    fails loudly when the group is not the size --gpus asked for."""
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        if rehearsing():
            import faulthandler

            faulthandler.dump_traceback_later(int(os.environ.get("HYPMI_BENCH_WATCHDOG_S", "100")), exit=True)
            dist.init_process_group("gloo")
        else:
            import torch

            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        seen = dist.get_world_size()
        if seen != args.gpus:
            print("bench.py: --gpus %d but the process group has %d rank(s)" % (args.gpus, seen), file=sys.stderr)
            dist.destroy_process_group()
            sys.exit(3)
    elif args.gpus != 1:
        print("bench.py: --gpus %d but WORLD_SIZE is 1" % args.gpus, file=sys.stderr)
        sys.exit(3)
    return world, rank, local_rank, dist


def dry_run(args, world, rank, dist):
    import torch

    t = torch.ones(1, dtype=torch.int64)
    if world > 1:
        if not rehearsing():
            t = t.cuda()
        dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": int(t.item()), "requested": args.gpus,
                          "backend": dist.get_backend() if world > 1 else None}))
    if world > 1:
        dist.destroy_process_group()


# ---- workload ----------------------------------------------------------------------------------
def build_workload(cfg):
    """Host-side synthetic model + chain (float64 parameters)."""
    import synth

    if cfg["lda"] is None:
        return None, synth.make_splda(cfg["d"], cfg["K"])
    chain = synth.chain_cfg2(d=cfg["d"], lda_dim=cfg["lda"])
    return chain, synth.make_splda(cfg["lda"], cfg["K"])


def make_inputs(cfg, n, seed):
    import synth

    if cfg["lda"] is None:
        m = synth.make_splda(cfg["d"], cfg["K"])
        per = 122
        x = synth.length_norm(synth.sample_splda(m, (n + per - 1) // per, per, seed))[:n]
        return x.astype(np.float32)
    x, _ = synth.raw_xvectors(n, d=cfg["d"], seed=seed)
    return x.astype(np.float32)


def device_xvectors(n, d, seed, num_centers=10_000):
    """Raw x-vectors drawn on the device (0.5 x class centre + unit noise, class = row % num_centers):
    the shape of synth.raw_xvectors without a host round trip."""
    import torch

    g = torch.Generator(device="cuda").manual_seed(20)
    C = torch.randn((num_centers, d), device="cuda", generator=g)
    g = torch.Generator(device="cuda").manual_seed(seed)
    cls = torch.arange(n, device="cuda") % num_centers
    x = torch.empty((n, d), device="cuda")
    for r0 in range(0, n, 1 << 18):  # in slices: no second full-size temporary
        r1 = min(n, r0 + (1 << 18))
        x[r0:r1] = torch.randn((r1 - r0, d), device="cuda", generator=g)
        x[r0:r1] += 0.5 * C[cls[r0:r1]]
    return x, cls.to(torch.int32)


def host_cpu():
    """(model name, threads numpy's BLAS uses) of the host the CPU baseline runs on."""
    model = "unknown"
    try:
        fields = {}
        with open("/proc/cpuinfo") as f:
            for line in f:
                if ":" in line:
                    key, val = (t.strip() for t in line.split(":", 1))
                    fields.setdefault(key, val)
                if not line.strip():
                    break  # first processor only
        model = fields.get("model name", model)
        if "cpu family" in fields:  # a virtual machine may hide the part number: family / model identify it
            model += " (family %s model %s)" % (fields["cpu family"], fields.get("model", "?"))
    except OSError:
        pass
    try:
        from threadpoolctl import threadpool_info

        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:  # noqa: BLE001
        cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count()))
    return model, int(cores)


def cpu_baseline(cfg, chain, model, sample):
    """The float64 numpy oracle (same operation sequence as the reference) timed on a bounded
    square slice of a raw-score workload."""
    from oracle import plda_ref as O

    cpu, cores = host_cpu()
    n = min(sample, cfg["n_enrol"], cfg["n_test"])
    x1 = make_inputs(cfg, n, 31).astype(np.float64)
    x2 = make_inputs(cfg, n, 32).astype(np.float64)
    t0 = time.perf_counter()
    if chain is not None:
        y1, y2 = O.chain_predict(x1, chain), O.chain_predict(x2, chain)
    else:
        y1, y2 = x1, x2
    s = O.llr_1vs1(model, y1, y2)
    dt = time.perf_counter() - t0
    del s
    return {"value": n * n / dt, "unit": "scores/s", "cores": cores, "kind": "port", "cpu": cpu,
            "sample": "%d x %d trials (chain on 2 x %d vectors + SPLDA llr_1vs1), float64 numpy, %.2f s"
                      % (n, n, n, dt)}


def cpu_baseline_asnorm(cfg, chain, model, n_enrol=1000, n_test=4000):
    """The float64 oracle on the host for the AS-norm workloads: chain + SPLDA scores of a slice of
    the trial matrix against the full cohort, then the reference's adaptive S-norm loops."""
    from oracle import plda_ref as O

    cpu, cores = host_cpu()
    nc, nbest = cfg["n_coh"], cfg["nbest"]
    ne, nt = min(n_enrol, cfg["n_enrol"]), min(n_test, cfg["n_test"])
    xe, xt, xc = (make_inputs(cfg, n, seed).astype(np.float64) for n, seed in ((ne, 41), (nt, 43), (nc, 42)))
    t0 = time.perf_counter()
    ye, yt, yc = (O.chain_predict(x, chain) for x in (xe, xt, xc))
    O.adapt_s_norm(O.llr_1vs1(model, ye, yt), O.llr_1vs1(model, yc, yt), O.llr_1vs1(model, ye, yc), nbest, 0)
    dt = time.perf_counter() - t0
    return {"value": ne * nt / dt, "unit": "scores/s", "cores": cores, "kind": "port", "cpu": cpu,
            "sample": "%d enrol x %d test against the whole %d-member cohort (chain + 3 x SPLDA llr_1vs1 + "
                      "AdaptSNorm nbest %d, float64 numpy), %.2f s" % (ne, nt, nc, nbest, dt)}


def profiled_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 counter passes
    (FETCH_SIZE / WRITE_SIZE, corrected; see tools/summarize_prof.py), or None when no profile of
    this workload has been recorded.  Counters cannot be read from inside the timed run."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


# ---- AS-norm workloads -------------------------------------------------------------------------
def asnorm_run(plda, preproc, cfg, steps, warmup, rank, world, dist, split_format="pair"):
    """Times `steps` passes of chain + SPLDA + adaptive S-norm; returns (seconds per step of this rank
    between barriers, stage dict, backend, buffers).  split_format: operands of the three score matrices
    ("pair" / "triple": 16-bit matrix cores at fp32 accuracy; "none": fp32 rows on the fp32 matrix cores)."""
    import torch

    from hyperion_amd import engine
    from hyperion_amd.backend import PLDABackend

    ne, nt = cfg["n_enrol"], cfg["n_test"]
    nc, nbest = cfg["n_coh"], cfg["nbest"]
    sink = bool(cfg.get("sink"))
    be = PLDABackend(plda, preproc, nbest=nbest, test_chunk=cfg.get("test_chunk", 65536))
    be.split_format = None if split_format == "none" else split_format
    xe, ke = device_xvectors(ne, cfg["d"], 41)
    xc, _ = device_xvectors(nc, cfg["d"], 42)
    xt, kt = device_xvectors(nt, cfg["d"], 43 + 1000 * rank)  # every rank owns a different test shard
    out = None
    if not sink:
        out = torch.empty((ne, engine.round_up(nt, 32)), dtype=torch.float32, device="cuda")[:, :nt]
    hists = []

    def step():
        coh = be.prepare_cohort(xe, xc, rank=rank, world=world)  # one all-gather of top-k lists
        if sink:
            from hyperion_amd.metrics import TrialHistogram

            h = TrialHistogram(-40.0, 40.0, 4096)
            be.score_snorm(xt, coh, reduce=h.sink(ke, kt))
            h.all_reduce()  # 2 x 4096 counts + 8 moments
            hists[:] = [h]
        else:
            be.score_snorm(xt, coh, out=out)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    be.trace = []  # HIP events around the launches of the timed steps (on the stream they run on)
    gather_clock(reset=True)  # (before the timed region: it synchronises)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    trace, be.trace = be.trace, None
    ghz = gather_clock()
    per = {}
    for name, e0, e1, shape in trace:
        per.setdefault(name, []).append((e0.elapsed_time(e1), shape))
    st = {"per_step_ms": {k: float(sum(ms for ms, _ in v)) / steps for k, v in per.items()}}
    # the dominant kernel: the cohort-statistics gather (two launches per test chunk)
    g = per.get("gather_z", []) + per.get("gather_t", [])
    if g:
        avg_ms = float(np.mean([ms for ms, _ in g]))
        avg_bytes = float(np.mean([4.0 * r * q * k for _, (r, q, k) in g]))
        st["gather"] = {"launches_per_step": len(g) // steps, "avg_launch_ms": avg_ms,
                        "algorithmic_bytes_per_launch": avg_bytes, "shader_clock_ghz": ghz}
    if sink:
        h = hists[0]
        m = h.moments()
        st["sink"] = {"trials": m["tar"]["count"] + m["non"]["count"], "tar": m["tar"], "non": m["non"],
                      "eer": h.eer(), "min_dcf_0.05": h.min_dcf(0.05)[0]}
    else:
        st["checksum"] = float(out.sum(dtype=torch.float64).item())
    return dt, st, be, (xe, xc, xt, out)


def gather_clock(reset=False):
    """Shader clock (GHz) the LDS-resident gather kernels have run at since the last reset (None if none ran)."""
    import ctypes

    from hyperion_amd import _lib

    ghz = ctypes.c_double(0.0)
    _lib.check(_lib.load().hypmi_gather_clock(ctypes.byref(ghz), 1 if reset else 0))
    return float(ghz.value) or None


def asnorm_check(be, bufs, chain, model, cfg, n_rows=64, n_cols=2048):
    """The AS-norm output against the float64 oracle on a slab of n_rows enrolment models x n_cols tests
    against the whole cohort, recomputed on the CPU from the same raw inputs (outside the timed region).
    Rows: first, middle and last third of the slab from the first, middle and last models; tests: the
    first ones, a run that straddles an interior chunk boundary, and the last ones.  Reports the mixed
    error |d| / max(|ref|, 1), the strict relative error |d| / |ref| and the share of entries whose
    strict relative error exceeds 1e-4."""
    import torch

    from oracle import plda_ref as O

    xe, xc, xt, out = bufs
    ne, nt = xe.shape[0], xt.shape[0]

    def spread(n, want, mid):
        want = min(want, n)
        a = want // 3
        b = want - 2 * a
        m0 = max(a, min(n - a - b, mid - b // 2))
        return np.unique(np.concatenate([np.arange(a), np.arange(m0, m0 + b), np.arange(n - a, n)]))

    chunk = be.last_memory_plan["test_chunk"] if be.last_memory_plan else cfg.get("test_chunk", 65536)
    rows = spread(ne, n_rows, ne // 2)
    cols = spread(nt, n_cols, (nt // chunk // 2) * chunk if nt > chunk else nt // 2)
    r_dev, c_dev = torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()
    if out is None:  # streamed workload: normalise the slab's tests again, materialised
        got = be.score_snorm(xt[c_dev], be.prepare_cohort(xe[r_dev], xc))
    else:
        got = out[r_dev][:, c_dev]
    got = got.cpu().numpy().astype(np.float64)
    f = lambda t: O.chain_predict(t.cpu().numpy().astype(np.float64), chain)  # noqa: E731
    ye, yt, yc = f(xe[r_dev]), f(xt[c_dev]), f(xc)
    ref = O.adapt_s_norm(O.llr_1vs1(model, ye, yt), O.llr_1vs1(model, yc, yt), O.llr_1vs1(model, ye, yc),
                         cfg.get("nbest", 300), 0)
    d = np.abs(got - ref)
    rel = d / np.maximum(np.abs(ref), 1e-300)
    return {"max_err_over_max_abs_1": float(np.max(d / np.maximum(np.abs(ref), 1.0))),
            "max_err_relative": float(rel.max()),
            "frac_relative_err_gt_1e-4": float(np.mean(rel > 1e-4)),
            "entries": int(ref.size), "min_abs_ref": float(np.abs(ref).min()),
            "slab": "%d enrol x %d test (first / middle / last models; first tests, a run across the chunk boundary "
                    "at test %d, last tests), whole cohort, float64 oracle" % (len(rows), len(cols), cols[len(cols) // 2])}


def score_matrix_stage(plda, preproc, n=100_000, reps=3):
    """The score-matrix stage on its own at the size of the fused-chain configuration (n x n trials,
    chain + SPLDA): the fp32-MFMA kernel against the fp32 matrix peak, and the default fp16-pair
    kernel against HBM (its products run on the 16-bit pipe, its stores bind it)."""
    import torch

    from hyperion_amd import _lib, engine
    from hyperion_amd.engine import SIDE_ENROL, SIDE_TEST

    plan = plda.score_plan(preproc)
    x1, _ = device_xvectors(n, 512, 21)
    x2, _ = device_xvectors(n, 512, 22)
    S = torch.empty((n, engine.round_up(n, 32)), dtype=torch.float32, device="cuda")[:, :n]
    K = plda.y_dim
    res = {"trials": "%d x %d" % (n, n)}
    for name, fmt in (("fp32_mfma", None), ("fp16_pair", "pair")):
        ms_score, ms_step = [], []
        for it in range(reps + 1):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            if fmt:  # one kernel per side: chain, projection, packing and the fp16-pair conversion
                G1, G2 = plan.prologue_split(x1, SIDE_ENROL, fmt), plan.prologue_split(x2, SIDE_TEST, fmt)
            else:
                G1, G2 = plan.prologue(x1, SIDE_ENROL), plan.prologue(x2, SIDE_TEST)
            e[1].record()
            engine.score(G1, G2, out=S, split=False)
            e[2].record()
            torch.cuda.synchronize()
            if it:  # the first pass warms up
                ms_score.append(e[1].elapsed_time(e[2]))
                ms_step.append(e[0].elapsed_time(e[2]))
        ms, step = float(np.mean(ms_score)), float(np.mean(ms_step))
        tf = 2.0 * K * n * n / (ms * 1e-3) / 1e12
        row_bytes = 4.0 * plan.kp if not fmt else float(getattr(_lib.load(), engine.SPLIT_FORMATS[fmt][0])(plan.kp))
        gbs = (4.0 * n * n + 2 * n * row_bytes) / (ms * 1e-3) / 1e9
        res[name] = {"score_kernel_ms": ms, "step_ms": step, "scores_per_s": n * float(n) / (step * 1e-3),
                     "hbm_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS}
        if fmt:  # fp16 products: the fp32 matrix peak is not this kernel's roof, so no fraction of it
            res[name]["fp32_equivalent_TFLOPs"] = tf
        else:
            res[name].update({"algorithmic_fp32_TFLOPs": tf, "pct_fp32_mfma_peak": 100.0 * tf / FP32_MFMA_PEAK_TFLOPS})
    res["fp16_pair"]["note"] = ("products on the fp16 matrix cores (3 partial products per fp32 product), so the "
                                "fp32 matrix peak is not this kernel's roof: the HBM store of the score matrix is (hbm_frac)")
    return res


def million_share_stage(plda, preproc):
    """The `asnorm_1m_125k` workload for one timed step: all 1 000 000 enrolment models against the
    125 000 tests one GPU of an 8-GPU node owns, every block of normalised scores streamed into the
    target / non-target histograms (1.25e11 trials, none kept)."""
    import torch

    cfg = WORKLOADS["asnorm_1m_125k"]
    dt, st, be, bufs = asnorm_run(plda, preproc, cfg, 1, 0, 0, 1, None)
    res = {"workload": "asnorm_1m_125k", "steps": 1, "warmup": 0, "s_per_step": dt,
           "scores_per_s": cfg["n_enrol"] * float(cfg["n_test"]) / dt,
           "stages_ms_per_step": st["per_step_ms"], "trials_counted": st["sink"]["trials"],
           "eer": st["sink"]["eer"]}
    del be, bufs
    torch.cuda.empty_cache()
    return res


def fp32_only_stage(plda, preproc, cfg):
    """The same workload with every score matrix on the fp32 matrix cores (fp32 packed rows,
    v_mfma_f32_32x32x2_f32; `--split-format none`): one warm-up and one timed step."""
    import torch

    dt, st, be, bufs = asnorm_run(plda, preproc, cfg, 1, 1, 0, 1, None, split_format="none")
    res = {"split_format": "none", "steps": 1, "warmup": 1, "ms_per_step": dt * 1e3,
           "scores_per_s": cfg["n_enrol"] * float(cfg["n_test"]) / dt, "stages_ms_per_step": st["per_step_ms"],
           "checksum": st.get("checksum")}
    del be, bufs
    torch.cuda.empty_cache()
    return res


# ms per step of the default workload on the boxes it has been measured on with one tree (same command):
# boxes differ by this much in every stage, so differences between runs below it say nothing
BOX_SPREAD_NOTE = ("same command on different MI355X boxes: 567-625 ms per step (9 %); the gather follows the shader "
                   "clock a box sustains under its power cap (roofline.shader_clock_ghz: 2.15-2.23 GHz seen)")


def main_asnorm(args, cfg, chain, model, plda, preproc, world, rank, dist):
    """AS-norm workloads: value = normalised scores/s over all ranks (test rows sharded)."""
    import torch

    dt, st, be, bufs = asnorm_run(plda, preproc, cfg, args.steps, args.warmup, rank, world, dist,
                                  split_format=args.split_format)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ne, nt, nc, nbest = cfg["n_enrol"], cfg["n_test"], cfg["n_coh"], cfg["nbest"]
    sink = bool(cfg.get("sink"))
    out = {
        "metric": METRIC, "value": ne * float(nt) * world / dt, "unit": "scores/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        # the arithmetic types that execute: score matrices = scaled fp16 pairs on the matrix cores with
        # fp32 accumulation (fp32-equivalent accuracy); selection, cohort statistics, normalisation = fp32
        "dtype": {"pair": "f32 (score products: fp16-pair split, f32 accumulate)",
                  "triple": "f32 (score products: bf16-triple split, f32 accumulate)",
                  "none": "f32"}[args.split_format], "data": "synthetic",
        "config": {"workload": "%s: CentWhiten(512x512)+LNorm+LDA(200)+SPLDA(y_dim 150) scores with adaptive "
                               "S-norm, %d enrol x %d test per GPU, cohort %d, nbest %d%s"
                               % (args.workload, ne, nt, nc, nbest,
                                  ", streamed into tar/non histograms" if sink else ", result kept in HBM"),
                   "n_enrol": ne, "n_test_per_gpu": nt, "n_cohort": nc, "nbest": nbest, "x_dim": cfg["d"],
                   "per_gpu": "one whole MI355X per rank (SPX mode), not an XCD partition",
                   "parallelism": "test rows sharded over %d GPU(s), enrol and cohort replicated, one "
                                  "all-gather of top-k lists" % world},
        "roofline": None,
        "stages_ms_per_step": st["per_step_ms"],
    }
    if args.workload == "asnorm_10k_1m":
        out["box_to_box_spread"] = BOX_SPREAD_NOTE
    if "gather" in st:
        g = st["gather"]
        ach = g["algorithmic_bytes_per_launch"] / (g["avg_launch_ms"] * 1e-3) / 1e9
        out["roofline"] = {
            "kernel": "gather_pivot_kernel<2>", "bound": "lds", "achieved": ach, "peak": LDS_PEAK_GBS,
            "unit": "GB/s", "frac": ach / LDS_PEAK_GBS, "traffic": profiled_traffic(args.workload),
            "avg_launch_ms": g["avg_launch_ms"], "launches_per_step": g["launches_per_step"],
            "algorithmic_bytes_per_launch": g["algorithmic_bytes_per_launch"],
            "shader_clock_ghz": g.get("shader_clock_ghz"),
            "note": "cohort-statistics gather, %.0f %% of the step: one gathered fp32 (4 B) per (row, list, member) "
                    "from LDS-resident rows; peak = conflict-free LDS read rate (256 B/clk/CU) at the nominal "
                    "2.4 GHz.  The loop is bound by instruction issue, not LDS bytes: per 8 gathered bytes a SIMD "
                    "issues one address op, one ds_read_b64 and four plain f32 ops; the walk alone sustains 0.33 of "
                    "this peak (12.1 us per round of 1024 lists, tools/gather_overhead.py), a workgroup pays 13 us "
                    "on top for its row fill, the uneven finish of its waves and the turnaround.  The kernel runs at "
                    "the clock the power budget of the box leaves it (shader_clock_ghz, measured inside the timed "
                    "launches: shader cycles over the constant 100 MHz clock); boxes differ in it"
                    % (100.0 * g["avg_launch_ms"] * g["launches_per_step"] / (dt * 1e3))}
    if "sink" in st:
        out["sink"] = st["sink"]
    else:
        out["checksum"] = st["checksum"]
    if rank == 0 and not args.no_check and chain is not None:
        out["parity"] = asnorm_check(be, bufs, chain, model, cfg)
    try:
        import ctypes

        from hyperion_amd import _lib

        redo = ctypes.c_ulonglong(0)
        _lib.check(_lib.load().hypmi_gather_redo_count(ctypes.byref(redo), 0))
        out["gather_pairs_redone_with_list_pivot"] = int(redo.value)
    except Exception:  # noqa: BLE001
        pass
    del be, bufs
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_stages and args.workload == "asnorm_10k_1m":
        try:  # extra measurements: a failure here must not cost the line its headline
            out["score_matrix_stage"] = score_matrix_stage(plda, preproc)
        except Exception as e:  # noqa: BLE001
            out["score_matrix_stage"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:  # one GPU's share of the 1M x 1M configuration, one step (kernels are warm from the run above)
            out["share_of_1m_x_1m_stage"] = million_share_stage(plda, preproc)
        except Exception as e:  # noqa: BLE001
            out["share_of_1m_x_1m_stage"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if args.split_format != "none":
            try:  # the headline workload with every score matrix on the fp32 matrix cores
                out["fp32_only"] = fp32_only_stage(plda, preproc, cfg)
            except Exception as e:  # noqa: BLE001
                out["fp32_only"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and chain is not None:
        try:
            out["cpu_baseline"] = cpu_baseline_asnorm(cfg, chain, model)
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        print(json.dumps(out))


# ---- raw-score workloads -----------------------------------------------------------------------
def main_scores(args, cfg, chain, model, plda, preproc, world, rank, dist):
    import torch

    from hyperion_amd import _lib, engine
    from hyperion_amd.engine import SIDE_ENROL, SIDE_TEST

    plan = plda.score_plan(preproc)
    n1, n2 = cfg["n_enrol"], cfg["n_test"]
    x1_h = make_inputs(cfg, n1, 21)
    x2_h = make_inputs(cfg, n2, 22 + 1000 * rank)  # every rank owns a different test shard
    x1 = engine.upload(x1_h)
    x2 = engine.upload(x2_h)
    ldo = engine.round_up(n2, 32)
    S = torch.empty((n1, ldo), dtype=torch.float32, device="cuda")[:, :n2]
    torch.cuda.synchronize()

    ev_pairs, ev_pro = [], []
    fmt = args.split_format
    use_split = fmt != "none" and engine.split_rows(plan.prologue(x1[:1], SIDE_ENROL), fmt) is not None
    products = {"pair": 3, "triple": 6}.get(fmt, 1)

    def step(record):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if record else None
        if record:
            ev[0].record()
        if use_split:  # 16-bit matrix cores at fp32 accuracy (fp16 pairs: made by the prologue kernel itself)
            G1, G2 = plan.prologue_split(x1, SIDE_ENROL, fmt), plan.prologue_split(x2, SIDE_TEST, fmt)
        else:
            G1, G2 = plan.prologue(x1, SIDE_ENROL), plan.prologue(x2, SIDE_TEST)
        if record:
            ev[1].record()
        engine.score(G1, G2, out=S, split=False)
        if record:
            ev[2].record()
            ev_pro.append((ev[0], ev[1]))
            ev_pairs.append((ev[1], ev[2]))

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    score_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_pairs]))
    pro_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_pro]))
    # algorithmic prologue traffic: the 512-d input read once, K + 1 floats (g, q) written per vector
    pro_bytes = float(n1 + n2) * (4.0 * cfg["d"] + 4.0 * (cfg["K"] + 1))
    K = cfg["K"]
    achieved_tflops = 2.0 * K * n1 * n2 / (score_ms * 1e-3) / 1e12
    # the split kernels execute 3 (fp16 pairs) or 6 (bf16 triples) partial products per fp32 product
    # over the K + 2 packed columns
    executed_tflops = products * 2.0 * (K + 2) * n1 * n2 / (score_ms * 1e-3) / 1e12
    out = {
        "metric": "PLDA trial scores/sec (512-d, SPLDA, no score normalisation)",
        "value": float(n1) * float(n2) * world * args.steps / dt, "unit": "scores/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"pair": "fp16-pair split products, f32 accumulate", "triple": "bf16-triple split products, "
                  "f32 accumulate", "none": "f32"}[fmt if use_split else "none"],
        "data": "synthetic",
        "config": {
            "workload": "%s: CentWhiten(512x512)+LNorm+LDA(%s)+SPLDA(y_dim %d) prologue on %d enrol + %d "
                        "test 512-d x-vectors, full %d x %d score matrix per GPU, no score normalisation"
                        % (args.workload, cfg["lda"], K, n1, n2, n1, n2),
            "n_enrol": n1, "n_test_per_gpu": n2, "x_dim": cfg["d"], "lda_dim": cfg["lda"], "y_dim": K,
            "parallelism": "test rows sharded over %d GPU(s), enrol replicated" % world,
        },
        "roofline": None,
    }
    # Two roofs for the score launch.  With 16-bit operands the matrix pipes need less time than the
    # 4 bytes per score need HBM, so HBM is the binding roof there; the fp32 kernel is matrix-bound.
    row_bytes = 4.0 * plan.kp
    if use_split:
        row_bytes = float(getattr(_lib.load(), engine.SPLIT_FORMATS[fmt][0])(plan.kp))
    hbm_bytes = 4.0 * n1 * n2 + (n1 + n2) * row_bytes  # the matrix written once, both operands read once
    hbm_view = {"bound": "hbm", "achieved": hbm_bytes / (score_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": hbm_bytes / (score_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": hbm_bytes}
    mfma_view = {"bound": "mfma", "achieved": achieved_tflops, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": achieved_tflops / FP32_MFMA_PEAK_TFLOPS,
                 "algorithmic_flops_per_launch": 2.0 * K * n1 * n2}
    head = dict(hbm_view if use_split else mfma_view)
    head["kernel"] = (("score_split_kernel<%d,%d>" % (2 if fmt == "pair" else 3, (plan.kp + 31) // 32)) if use_split
                      else "score_rs_kernel<%d,2>" % plan.kp)
    head["traffic"] = profiled_traffic(args.workload)
    head["avg_launch_ms"] = score_ms
    head["hbm" if not use_split else "mfma_fp32_equivalent"] = hbm_view if not use_split else mfma_view
    if use_split:
        head["executed"] = {"pipe": "%s MFMA (%d partial products per fp32 product, K + 2 packed columns)"
                                    % ("fp16" if fmt == "pair" else "bf16", products),
                            "achieved": executed_tflops, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": executed_tflops / F16_MFMA_PEAK_TFLOPS}
    out["roofline"] = head
    out["prologue_stage"] = {
        "what": "pre-processing chain + PLDA projection + packing (+ 16-bit split) of %d + %d vectors" % (n1, n2),
        "ms": pro_ms, "bound": "hbm", "achieved": pro_bytes / (pro_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": pro_bytes / (pro_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": pro_bytes}
    if chain is not None:  # u = x A1 (d x d, for the length norm) and x (A1 A2) (d x 2K) on the fp32 matrix cores
        pro_flops = 2.0 * cfg["d"] * (cfg["d"] + 2 * K) * (n1 + n2)
        out["prologue_stage"]["mfma"] = {"achieved": pro_flops / (pro_ms * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS,
                                         "unit": "TFLOP/s",
                                         "frac": pro_flops / (pro_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
    if rank == 0 and not args.no_check:
        # parity spot check outside the timed region: a slab of the device matrix vs the oracle
        from oracle import plda_ref as O

        r, c = min(16, n1), min(4096, n2)
        y1 = x1_h[:r].astype(np.float64)
        y2 = x2_h[:c].astype(np.float64)
        if chain is not None:
            y1, y2 = O.chain_predict(y1, chain), O.chain_predict(y2, chain)
        ref = O.llr_1vs1(model, y1, y2)
        got = S[:r, :c].cpu().numpy().astype(np.float64)
        err = np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)
        out["parity"] = {"max_err_over_max_abs_1": float(err.max()), "slab": [r, c],
                         "checksum": float(S.sum(dtype=torch.float64).item())}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(cfg, chain, model, args.cpu_sample)
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        print(json.dumps(out))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))  # nothing in this process has touched a GPU
    world, rank, local_rank, dist = init_ranks(args)
    if args.dry_run:
        return dry_run(args, world, rank, dist)
    import torch

    torch.cuda.set_device(0 if rehearsing() else local_rank)

    from hyperion_amd.pdfs.plda import SPLDA
    from hyperion_amd.transforms import LDA, CentWhiten, LNorm, TransformList

    cfg = dict(WORKLOADS[args.workload])
    if args.test_chunk:
        cfg["test_chunk"] = args.test_chunk
    if args.n_coh and "n_coh" in cfg:
        cfg["n_coh"] = args.n_coh
    if args.n_test:
        cfg["n_test"] = args.n_test
    chain, model = build_workload(cfg)
    plda = SPLDA(mu=model["mu"], V=model["V"], W=model["W"])
    preproc = None
    if chain is not None:
        preproc = TransformList([CentWhiten(mu=chain[0][1], T=chain[0][2], name="cw"), LNorm(name="ln"),
                                 LDA(mu=chain[2][1], T=chain[2][2], name="lda")])
    run = main_asnorm if "n_coh" in cfg else main_scores
    run(args, cfg, chain, model, plda, preproc, world, rank, dist)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
