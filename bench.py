#!/usr/bin/env python
"""Benchmark of the PLDA back-end scoring hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload chain_splda_100k|...]

One "step" = one pass of the hot path over one batch of synthetic x-vectors that are already
resident in HBM: embedding prologue (pre-processing chain + PLDA projection) for the enrol and the
test side, conversion of the packed rows to scaled fp16 pairs, then the full trial score matrix
(kept on the device).  With N > 1 every rank scores the
same enrol set against its own shard of test vectors (row-wise test sharding, no data-path
collective): per-GPU work is fixed, so scaling is "weak" and the reported value is the number of
scores all ranks produced divided by the slowest rank's time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X dense fp32 matrix peak (256 CU x 256 flop/clk x 2.4 GHz)
BF16_MFMA_PEAK_TFLOPS = 2516.6  # dense bf16 matrix peak (256 CU x 4096 flop/clk x 2.4 GHz)
HBM_PEAK_GBS = 8000.0
LDS_PEAK_GBS = 256 * 128 * 2.4  # GB/s: 128 bytes per clock and CU

WORKLOADS = {
    # n_enrol, n_test per GPU, input dim, LDA dim, y_dim; asnorm: cohort size and nbest
    "chain_splda_100k": dict(n_enrol=100_000, n_test=100_000, d=512, lda=200, K=150),
    "chain_splda_20k": dict(n_enrol=20_000, n_test=20_000, d=512, lda=200, K=150),
    "splda_vox1": dict(n_enrol=4874, n_test=4874, d=512, lda=None, K=150),
    "asnorm_10k_1m": dict(n_enrol=10_000, n_test=1_000_000, d=512, lda=200, K=150, n_coh=20_000, nbest=300),
    "asnorm_10k_128k": dict(n_enrol=10_000, n_test=131_072, d=512, lda=200, K=150, n_coh=20_000, nbest=300),
    # one GPU's share of the 1M x 1M matrix on 8 GPUs (125 000 tests per GPU: `--gpus 8` is the whole
    # job); the normalised scores are never kept, every block goes into target / non-target histograms
    "asnorm_1m_125k": dict(n_enrol=1_000_000, n_test=125_000, d=512, lda=200, K=150, n_coh=20_000, nbest=300,
                           sink=True, test_chunk=4096),
    "asnorm_100k_16k": dict(n_enrol=100_000, n_test=16_384, d=512, lda=200, K=150, n_coh=20_000, nbest=300,
                            sink=True, test_chunk=4096),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="chain_splda_100k", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=20000,
                    help="rows/cols of the square slice the CPU baseline is timed on")
    ap.add_argument("--split-format", default="pair", choices=["pair", "triple", "none"],
                    help="operands of the score kernel: scaled fp16 pairs, exact bf16 triples, or fp32")
    ap.add_argument("--no-asnorm-stage", action="store_true",
                    help="skip the extra AS-norm stage measurement of the default workload")
    return ap.parse_args()


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


def cpu_baseline(cfg, chain, model, sample):
    """The float64 numpy oracle (same operation sequence as the reference) timed on a bounded
    square slice of the same workload."""
    from oracle import plda_ref as O

    try:
        from threadpoolctl import threadpool_info

        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count()
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
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
    return {"value": n * n / dt, "unit": "scores/s", "cores": int(cores), "kind": "port", "cpu": cpu_model,
            "sample": "%d x %d trials (chain on 2 x %d vectors + SPLDA llr_1vs1), float64 numpy, %.2f s"
                      % (n, n, n, dt)}


def cpu_baseline_asnorm(cfg, chain, model, n_enrol=1000, n_test=4000):
    """The float64 oracle on the host for the AS-norm workloads: chain + SPLDA scores of a slice
    against the full cohort, then the reference's adaptive S-norm loops."""
    from oracle import plda_ref as O

    nc, nbest = cfg["n_coh"], cfg["nbest"]
    ne, nt = min(n_enrol, cfg["n_enrol"]), min(n_test, cfg["n_test"])
    xe, xt, xc = (make_inputs(cfg, n, seed).astype(np.float64) for n, seed in ((ne, 41), (nt, 43), (nc, 42)))
    t0 = time.perf_counter()
    ye, yt, yc = (O.chain_predict(x, chain) for x in (xe, xt, xc))
    O.adapt_s_norm(O.llr_1vs1(model, ye, yt), O.llr_1vs1(model, yc, yt), O.llr_1vs1(model, ye, yc), nbest, 0)
    dt = time.perf_counter() - t0
    return {"value": ne * nt / dt, "unit": "scores/s", "cores": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count())),
            "kind": "port", "sample": "%d enrol x %d test against the %d cohort (chain + SPLDA + AdaptSNorm, "
            "float64 numpy), %.2f s" % (ne, nt, nc, dt)}


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


def asnorm_stage(plda, preproc, cfg, n_test, steps, warmup, rank, world=1):
    """Chain + SPLDA + adaptive S-norm (20k cohort, top-300) for n_test test vectors against 10k
    enrolment models, everything device resident; returns normalised scores/s of one GPU."""
    import torch

    from hyperion_amd import engine
    from hyperion_amd.backend import PLDABackend

    ne = cfg["n_enrol"] if "n_coh" in cfg else 10_000  # the default workload appends a 10k-enrol stage
    nc, nbest = cfg.get("n_coh", 20_000), cfg.get("nbest", 300)
    sink = bool(cfg.get("sink"))
    be = PLDABackend(plda, preproc, nbest=nbest, test_chunk=cfg.get("test_chunk", 65536))
    if sink:  # too large for host-side generation and for a materialised result
        xe, ke = device_xvectors(ne, cfg["d"], 41)
        xc, _ = device_xvectors(nc, cfg["d"], 42)
        xt, kt = device_xvectors(n_test, cfg["d"], 43 + 1000 * rank)
        out = None
    else:
        xe = engine.upload(make_inputs(cfg, ne, 41))
        xc = engine.upload(make_inputs(cfg, nc, 42))
        xt = engine.upload(make_inputs(cfg, n_test, 43 + 1000 * rank))
        out = torch.empty((ne, engine.round_up(n_test, 32)), dtype=torch.float32, device="cuda")[:, :n_test]
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
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = {"workload": "chain+SPLDA+AdaptSNorm: %d enrol x %d test per GPU, cohort %d, nbest %d%s"
                      % (ne, n_test, nc, nbest, ", streamed into tar/non histograms" if sink else ""),
          "value": ne * n_test / dt, "unit": "normalised scores/s", "ms_per_step": dt * 1e3,
          "gather_GBs": 2.0 * nbest * 4 * ne * n_test / dt / 1e9}
    if sink:
        h = hists[0]
        m = h.moments()
        st["sink"] = {"trials": m["tar"]["count"] + m["non"]["count"], "tar": m["tar"], "non": m["non"],
                      "eer": h.eer(), "min_dcf_0.05": h.min_dcf(0.05)[0]}
        second = h.exact_pass()
        # outside the timed region: the second pass that makes the hull exact, when what it has to
        # keep fits comfortably (it re-scores everything)
        if second is not None and world == 1 and sum(second.expect) <= 2_000_000_000:
            t0 = time.perf_counter()
            be.score_snorm(xt, be.prepare_cohort(xe, xc), reduce=second.sink(ke, kt))
            second.finish()
            st["sink"].update({"exact_eer": h.eer(), "exact_min_dcf_0.05": h.min_dcf(0.05)[0],
                               "second_pass_s": time.perf_counter() - t0,
                               "second_pass_scores_kept": int(sum(second.expect))})
    else:
        st["checksum"] = float(out.sum(dtype=torch.float64).item())
    return st, be, (xe, xc, xt, out)


def device_xvectors(n, d, seed, num_centers=10_000):
    """Raw x-vectors drawn on the device (0.5 x class centre + unit noise, class = row % num_centers):
    the shape of synth.raw_xvectors without a host round trip, for the million-vector workloads."""
    import torch

    g = torch.Generator(device="cuda").manual_seed(20)
    C = torch.randn((num_centers, d), device="cuda", generator=g)
    g = torch.Generator(device="cuda").manual_seed(seed)
    cls = torch.arange(n, device="cuda") % num_centers
    x = torch.randn((n, d), device="cuda", generator=g)
    x += 0.5 * C[cls]
    return x, cls.to(torch.int32)


def asnorm_check(be, bufs, chain, model, cfg):
    """Spot check of the AS-norm output against the oracle on a slab (all 20k cohort, 8 enrol rows,
    64 tests): the slab's rows/columns are recomputed on the CPU from the same raw inputs."""
    from oracle import plda_ref as O

    xe, xc, xt, out = bufs
    r, c = 8, 64
    if out is None:  # streamed workload: normalise the slab's tests again, materialised
        out = be.score_snorm(xt[:c], be.prepare_cohort(xe[:r], xc))
    f = lambda t, n: O.chain_predict(t[:n].cpu().numpy().astype(np.float64), chain)  # noqa: E731
    ye, yt, yc = f(xe, r), f(xt, c), f(xc, xc.shape[0])
    ref = O.adapt_s_norm(O.llr_1vs1(model, ye, yt), O.llr_1vs1(model, yc, yt), O.llr_1vs1(model, ye, yc),
                         cfg.get("nbest", 300), 0)
    got = out[:r, :c].cpu().numpy().astype(np.float64)
    return float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)))


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # HYPMI_BENCH_REHEARSE=1: every rank on GPU 0 with the gloo backend -- a rehearsal of the multi-rank
    # flow on a one-GPU machine (its throughput means nothing); the real launch is one rank per GPU over RCCL.
    rehearse = os.environ.get("HYPMI_BENCH_REHEARSE", "0") == "1"
    if world > 1 and rehearse:
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ.get("HYPMI_BENCH_WATCHDOG_S", "100")), exit=True)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
    elif world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    from hyperion_amd import engine
    from hyperion_amd.engine import SIDE_ENROL, SIDE_TEST
    from hyperion_amd.pdfs.plda import SPLDA
    from hyperion_amd.transforms import LDA, CentWhiten, LNorm, TransformList

    cfg = WORKLOADS[args.workload]
    chain, model = build_workload(cfg)
    plda = SPLDA(mu=model["mu"], V=model["V"], W=model["W"])
    preproc = None
    if chain is not None:
        preproc = TransformList([CentWhiten(mu=chain[0][1], T=chain[0][2], name="cw"), LNorm(name="ln"),
                                 LDA(mu=chain[2][1], T=chain[2][2], name="lda")])
    plan = plda.score_plan(preproc)

    if "n_coh" in cfg:
        return main_asnorm(args, cfg, chain, model, plda, preproc, plan, world, rank, dist)

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
        if record:
            p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            p0.record()
        G1 = plan.prologue(x1, SIDE_ENROL)
        G2 = plan.prologue(x2, SIDE_TEST)
        if use_split:  # 16-bit matrix cores at fp32 accuracy
            G1, G2 = engine.split_rows(G1, fmt), engine.split_rows(G2, fmt)
        if record:
            p1.record()
            ev_pro.append((p0, p1))
        if record:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            engine.score(G1, G2, out=S, split=False)
            e1.record()
            ev_pairs.append((e0, e1))
        else:
            engine.score(G1, G2, out=S, split=False)

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
    scores_per_step = float(n1) * float(n2) * world
    value = scores_per_step * args.steps / dt
    K = cfg["K"]
    achieved_tflops = 2.0 * K * n1 * n2 / (score_ms * 1e-3) / 1e12
    # the split kernels execute 3 (fp16 pairs) or 6 (bf16 triples) partial products per fp32 product
    # over the K + 2 packed columns
    executed_tflops = products * 2.0 * (K + 2) * n1 * n2 / (score_ms * 1e-3) / 1e12

    out = {
        "metric": "PLDA trial scores/sec (512-d, SPLDA)",
        "value": value,
        "unit": "scores/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
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
    from hyperion_amd import _lib

    row_bytes = 4.0 * plan.kp
    if use_split:
        row_bytes = float(getattr(_lib.load(), engine.SPLIT_FORMATS[fmt][0])(plan.kp))
    # algorithmic bytes: the score matrix written once plus both operands read once
    hbm_bytes = 4.0 * n1 * n2 + (n1 + n2) * row_bytes
    hbm_view = {"bound": "hbm", "achieved": hbm_bytes / (score_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": hbm_bytes / (score_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": hbm_bytes,
                "hbm_store_GBs": 4.0 * n1 * n2 / (score_ms * 1e-3) / 1e9}
    mfma_view = {"bound": "mfma", "achieved": achieved_tflops, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": achieved_tflops / FP32_MFMA_PEAK_TFLOPS,
                 "algorithmic_flops_per_launch": 2.0 * K * n1 * n2,
                 "note": ("algorithmic fp32 flops (2 K per score) against the fp32 matrix peak; above 1 because "
                          "the products run on the 16-bit matrix cores (%s)" % (
                              "scaled fp16 pairs, 3 partial products" if fmt == "pair" else
                              "exact bf16 triples, 6 partial products")) if use_split else "fp32 matrix cores"}
    head = dict(hbm_view if use_split else mfma_view)
    head["kernel"] = (("score_split_kernel<%d,%d>" % (2 if fmt == "pair" else 3, (plan.kp + 31) // 32)) if use_split
                      else "score_rs_kernel<%d,2>" % plan.kp)
    head["traffic"] = profiled_traffic(args.workload)
    head["avg_launch_ms"] = score_ms
    head["hbm" if not use_split else "mfma_fp32_equivalent"] = hbm_view if not use_split else mfma_view
    out["roofline"] = head
    out["prologue_stage"] = {
        "what": "pre-processing chain + PLDA projection + packing (+ 16-bit split) of %d + %d vectors" % (n1, n2),
        "ms": pro_ms, "bound": "hbm", "achieved": pro_bytes / (pro_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": pro_bytes / (pro_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "algorithmic_bytes": pro_bytes,
        "note": "matrix-core bound in practice: see mfma"}
    if chain is not None:  # u = x A1 (d x d, for the length norm) and x (A1 A2) (d x 2K) on the fp32 matrix cores
        pro_flops = 2.0 * cfg["d"] * (cfg["d"] + 2 * K) * (n1 + n2)
        out["prologue_stage"]["mfma"] = {"achieved": pro_flops / (pro_ms * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS,
                                         "unit": "TFLOP/s", "frac": pro_flops / (pro_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                         "algorithmic_flops": pro_flops}
    if use_split:
        out["roofline"]["executed"] = {
            "pipe": "%s MFMA (%d partial products per fp32 product, K + 2 packed columns)"
                    % ("fp16" if fmt == "pair" else "bf16", products),
            "achieved": executed_tflops, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": executed_tflops / BF16_MFMA_PEAK_TFLOPS}

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
    if rank == 0 and world == 1 and not args.no_asnorm_stage and chain is not None:
        del S
        torch.cuda.empty_cache()
        try:  # an extra measurement: a failure here must not cost the line its headline
            st, be, bufs = asnorm_stage(plda, preproc, cfg, 131_072, 2, 1, rank)
            if not args.no_check:
                st["parity_max_err_over_max_abs_1"] = asnorm_check(be, bufs, chain, model, cfg)
            out["asnorm_stage"] = st
        except Exception as e:  # noqa: BLE001
            out["asnorm_stage"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(cfg, chain, model, args.cpu_sample)
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main_asnorm(args, cfg, chain, model, plda, preproc, plan, world, rank, dist):
    """AS-norm workloads: value = normalised scores/s over all ranks (test rows sharded)."""
    import torch

    if world > 1:
        dist.barrier()
    st, be, bufs = asnorm_stage(plda, preproc, cfg, cfg["n_test"], args.steps, args.warmup, rank, world)
    dt = st["ms_per_step"] * 1e-3
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ne, nt = cfg["n_enrol"], cfg["n_test"]
    out = {
        "metric": "PLDA trial scores/sec (512-d, SPLDA+AS-norm)", "value": ne * nt * world / dt,
        "unit": "scores/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.workload + ": " + st["workload"], "parallelism":
                   "test rows sharded over %d GPU(s), enrol and cohort replicated" % world},
        # the dominant kernel gathers from LDS, not from HBM: 2 x nbest fp32 values per normalised score
        # against the LDS bandwidth of the chip (256 CUs x 128 B/clk x 2.4 GHz)
        "roofline": {"kernel": "gather_stats_kernel<2,false>", "bound": "lds",
                     "note": "random LDS gathers; achieved = gathered bytes (2 x nbest x 4 B per normalised "
                             "score) / step time, the step including scoring, top-k and the sink",
                     "achieved": st["gather_GBs"], "peak": LDS_PEAK_GBS, "unit": "GB/s",
                     "frac": st["gather_GBs"] / LDS_PEAK_GBS, "traffic": None},
    }
    if "sink" in st:
        out["sink"] = st["sink"]
    if rank == 0 and not args.no_check and chain is not None:
        out["parity"] = {"max_err_over_max_abs_1": asnorm_check(be, bufs, chain, model, cfg)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and chain is not None:
        try:
            out["cpu_baseline"] = cpu_baseline_asnorm(cfg, chain, model)
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
