"""Device-resident back-end: pre-processing chain -> PLDA scores -> adaptive S-norm for large trial
matrices, with test vectors sharded over the GPUs of a node.

This is the flow of the reference's evaluation scripts (egs/voices_challenge/v1/steps_be/
eval-be-snorm-v1.py:28-91: llr_1vs1 on trials, cohort-vs-test and enrol-vs-cohort, then
AdaptSNorm.predict) kept on the device end to end: cohort score panels never leave HBM, test
vectors are processed in chunks so that the [test, cohort] panel stays bounded, and the enrol-side
top-k lists are the only thing exchanged between GPUs.

Sharding follows the reference's job grid (hyperion/utils/list_utils.py:95-112 `split_list`,
used by TrialNdx.split, hyperion/utils/trial_ndx.py:252-268): part i of n owns the contiguous
range [floor((i-1) N / n), floor(i N / n)).
"""
import numpy as np

from . import engine
from .engine import SIDE_ENROL, SIDE_TEST
from .score_norm.adapt_s_norm import (gather_apply, gather_apply_pairs, gather_apply_supported, gather_pairs_supported,
                                      gather_stats, gather_stats_pairs, seg_plan, topk_rows)


def shard_range(n, part, num_parts):
    """Rows owned by `part` (0-based) of `num_parts`: the contiguous floor partition of
    split_list (1-based idx there)."""
    lo = int(np.floor(part * float(n) / num_parts))
    hi = int(np.floor((part + 1) * float(n) / num_parts))
    return lo, hi


def settle_for_gloo(t, group=None):
    """The gloo backend (CPU rehearsals of the sharded path with device tensors) waits for the
    tensor's stream inside its worker thread; on ROCm that wait does not return while the stream is
    still busy, so the stream is drained first.  Nothing happens under RCCL."""
    import torch
    import torch.distributed as dist

    if t.is_cuda and dist.get_backend(group) == "gloo":
        torch.cuda.synchronize()


def all_gather_lists(local, counts, group=None):
    """All-gather of per-rank top-k lists [G, rows_of_rank * w] (w entries per row and group, rows
    contiguous) into [G, sum(rows) * w]; `counts[r]` is the width rank r contributes.  One collective;
    with a single rank, no process group, or a single entry in `counts` (a caller that scores on its
    own inside a larger job, e.g. rank 0's spot check) the input is returned unchanged."""
    import torch
    import torch.distributed as dist

    if len(counts) == 1 or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    assert len(counts) == world
    k = local.shape[0]
    width = max(counts)
    send = torch.zeros((k, width), dtype=local.dtype, device=local.device)
    send[:, :local.shape[1]] = local
    recv = torch.empty((world * k, width), dtype=local.dtype, device=local.device)
    settle_for_gloo(send, group)
    # gathered as raw bytes: RCCL process groups have no 16-bit integer type
    dist.all_gather_into_tensor(recv.view(torch.uint8), send.contiguous().view(torch.uint8), group=group)
    recv = recv.view(world, k, width)
    return torch.cat([recv[r, :, :counts[r]] for r in range(world)], dim=1).contiguous()


LAZY_MAX_COHORT = 65_536  # largest cohort the register-resident top-k kernel takes


# enrolment rows one workgroup of the T-part statistics kernel walks at most (128 rounds of 1024 lists): blocks
# of a multiple of it leave no short last workgroup per test
WALK_ROWS = 128 * 1024


class SnormMemoryPlan(dict):
    """HBM budget of one rank of the chain -> PLDA -> adaptive S-norm pipeline, in bytes, and the tile
    shape chosen from it: `test_chunk` tests per chunk and `enrol_block` enrolment rows per tile.
    Keys: resident (packed rows, enrol-vs-cohort panel, lists), per_chunk (test-vs-cohort panel, lists
    and packed rows of one chunk of tests), per_tile (the [enrol_block, test_chunk] intermediates),
    result (the materialised matrix, 0 when it is streamed), total, budget."""


def plan_snorm_memory(n_enrol, n_cohort, n_test, nbest, row_bytes, budget_bytes, test_chunk=65536,
                      materialise=True, tile_arrays=4, exact=True, x_dim=0, min_chunk=4096, min_block=1024,
                      tile_bytes=None, resident_allocated=False):
    """Sizes the adaptive S-norm pipeline of one rank before anything is allocated.

    The job grid of the reference gives every process one block of the trial matrix and lets the
    operating system find out whether it fits (egs/voxceleb/v1/steps_be/eval_be_v1.sh:33-60); here the
    rank computes what it will hold -- nothing of it grows with the number of tests except the
    materialised result -- picks the largest chunk of tests (<= test_chunk) and block of enrolment rows
    that fit `budget_bytes`, and raises MemoryError with the numbers when even the smallest tile does not.

    row_bytes: bytes of one packed row in the score kernel's operand format.  tile_arrays: fp32
    [enrol_block, test_chunk] arrays alive per tile (4 on the fused path: raw scores, the two T-part
    statistics, the normalised block; 8 unfused).  tile_bytes: optional cap on the per-tile part alone.
    resident_allocated: the enrolment / cohort side is in memory already and `budget_bytes` is what is
    left beside it.  Pure arithmetic (no device needed)."""
    ne, nc, nt, k = int(n_enrol), int(n_cohort), int(n_test), int(nbest)
    ld_c = (nc + 31) // 32 * 32
    k8 = (k + 7) // 8 * 8
    f64_row = 8 * 152 if exact else 0  # upper bound of a float64 packed row (K + 2 doubles, K <= 150 typical)
    resident = (ne * (row_bytes + 8)                 # enrolment packed rows + (|g|, |q|) bounds
                + nc * (row_bytes + f64_row + 8)     # cohort rows, fp32 operand and float64
                + ne * ld_c * 4                      # enrol-vs-cohort panel
                + 2 * ne * k8 * 2)                   # enrolment top-k lists: local + gathered
    result = ne * ((nt + 31) // 32 * 32) * 4 if materialise else 0

    def per_chunk(c):
        return c * (ld_c * 4 + 2 * row_bytes + 8 + k8 * 2 + 4 * x_dim)  # panel, both packings, bounds, lists, chain scratch

    def per_tile(b, c):
        return tile_arrays * 4 * (b + 32) * (c + 32)

    held = 0 if resident_allocated else resident
    avail = int(budget_bytes) - held - result
    chunk = max(1, min(int(test_chunk), nt))
    while True:
        room = avail - per_chunk(chunk)
        if tile_bytes is not None:
            room = min(room, int(tile_bytes))
        block = ne
        if per_tile(ne, chunk) > room:
            block = room // (tile_arrays * 4 * (chunk + 32)) - 32
            block = block // min_block * min_block
            if block >= WALK_ROWS:
                block = block // WALK_ROWS * WALK_ROWS
        if block >= min(min_block, ne):
            break
        if chunk <= min_chunk:
            need = held + result + per_chunk(chunk) + per_tile(min(min_block, ne), chunk)
            raise MemoryError(
                "adaptive S-norm of %d enrolment models x %d tests against a %d-member cohort does not fit: "
                "%.1f GB resident%s (enrol-vs-cohort panel %.1f GB) + %.1f GB result + %.1f GB for the smallest "
                "tile (%d tests x %d models) = %.1f GB to allocate, %.1f GB available -- shard the enrolment set "
                "or stream the result (reduce=)"
                % (ne, nt, nc, resident / 1e9, " (already held)" if resident_allocated else "", ne * ld_c * 4 / 1e9,
                   result / 1e9, (need - held - result) / 1e9, chunk, min(min_block, ne), need / 1e9,
                   budget_bytes / 1e9))
        chunk = max(min_chunk, chunk // 2)
    plan = SnormMemoryPlan(resident=resident, result=result, per_chunk=per_chunk(chunk),
                           per_tile=per_tile(block, chunk), test_chunk=chunk, enrol_block=block,
                           budget=int(budget_bytes))
    plan["total"] = resident + result + plan["per_chunk"] + plan["per_tile"]  # what the rank holds at its peak
    return plan


def device_budget(fraction=0.9):
    """Bytes this process may still use on the current device: what the driver reports free plus what
    torch's allocator holds without using, times `fraction`."""
    torch = engine.require_gpu()
    free, _ = torch.cuda.mem_get_info()
    cached = torch.cuda.memory_reserved() - torch.cuda.memory_allocated()
    return int(fraction * (free + cached))


def _take64(x, lo, hi, rows):
    """Rows `rows` (device int32, relative to lo) of x[lo:hi] as device float64, from the caller's own
    array (a float64 host array keeps the digits its fp32 device copy lost)."""
    torch = engine.require_gpu()
    if isinstance(x, torch.Tensor):
        return engine.upload64(x[lo:hi][rows.long()])
    return engine.upload64(np.asarray(x)[lo:hi][rows.cpu().numpy()])


def _rows(G, r0, r1):
    """Rows r0..r1-1 of packed rows in either form (fp32 matrix or SplitRows)."""
    return G.rows(r0, r1) if isinstance(G, engine.SplitRows) else G[r0:r1]


class PLDABackend:
    """plda: a hyperion_amd SPLDA / FRPLDA / PLDA; preproc: optional TransformList folded into the
    embedding prologue; nbest / nbest_discard: adaptive S-norm cohort selection."""

    def __init__(self, plda, preproc=None, nbest=300, nbest_discard=0, test_chunk=65536, exact_topk=True):
        self.plda = plda
        self.plan = plda.score_plan(preproc)
        self.nbest = int(nbest)
        self.nbest_discard = int(nbest_discard)
        self.test_chunk = int(test_chunk)
        # cap (bytes, or None) on the [enrol, chunk] fp32 intermediates of one tile; less when the device
        # has less free when score_snorm starts (see plan_snorm_memory).  With more enrolment models than
        # fit per chunk of tests the enrolment rows are walked in blocks, so that the chunk of tests can
        # stay long (every workgroup of the statistics kernels amortises its rows over the lists of one tile)
        self.tile_bytes = 96 << 30
        self.trials_before_topk = True  # order of a chunk's launches, see score_snorm
        # operands of the score kernel: "pair" (scaled fp16 pairs, the default), "triple" (exact bf16 triples)
        # or None: fp32 packed rows on the fp32 matrix cores
        self.split_format = engine.DEFAULT_SPLIT
        self.memory_fraction = 0.85  # share of the free device memory the pipeline may plan with
        self.last_memory_plan = None
        # finish the normalisation inside the Z-part statistics kernel (no separate apply pass)
        self.fuse_apply = True
        self.pair_stats = True  # fused path: T-part mean and 1 / std side by side (one line per list in the Z part)
        # float64 rows for the exact top-k only where a boundary is in doubt (False: for every vector)
        self.lazy_exact = True
        # re-decide top-k members that fp32 cannot separate from the selection boundary with fp64
        # scores (costs one fp64 prologue per vector; without it ~0.05 % of test columns pick one
        # different cohort member than float64 arithmetic would)
        self.exact_topk = bool(exact_topk)
        # set to a list to record (stage, start event, end event, shape) of every launch group on the
        # stream it runs on (bench.py: live kernel durations inside the timed step)
        self.trace = None
        self.topk_counters = None
        self._skipped_reported = 0

    def topk_report(self):
        """(rows whose top-k boundary was re-decided from float64 scores, rows where that was given
        up) since this object was made.  A skipped row has more than 64 cohort scores that fp32 cannot
        tell from the boundary score (dense or quantised cohort scores): its selection follows the
        fp32 tie order and may differ from a float64 ranking.  Synchronises with the device."""
        if self.topk_counters is None:
            return 0, 0
        a, b = (int(v) for v in self.topk_counters.cpu())
        return a, b

    def _warn_skipped(self):
        skipped = self.topk_report()[1]
        if skipped > self._skipped_reported:
            import warnings

            warnings.warn("adaptive S-norm: %d cohort selection(s) have more than 64 scores inside the fp32 "
                          "uncertainty band of the top-k boundary and were not re-decided in float64; those "
                          "rows follow the fp32 tie order" % (skipped - self._skipped_reported), RuntimeWarning)
            self._skipped_reported = skipped

    def _timed(self, name, fn, shape=None):
        if self.trace is None:
            return fn()
        torch = engine.require_gpu()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self.trace.append((name, e0, e1, shape))
        return r

    def memory_plan(self, n_enrol, n_cohort, n_test, materialise=True, resident_allocated=False, fused=True,
                    budget_bytes=None):
        """plan_snorm_memory for this backend's plan and settings against the memory the device has
        free now (or `budget_bytes`); with resident_allocated the enrolment / cohort side is taken to be
        in memory already (score_snorm after prepare_cohort).  Raises MemoryError before anything is
        allocated when the job cannot fit; the plan is kept in `last_memory_plan`."""
        from . import _lib

        fmt = self.split_format
        row_bytes = int(getattr(_lib.load(), engine.SPLIT_FORMATS[fmt][0])(self.plan.kp)) if fmt else 0
        row_bytes = row_bytes or 4 * self.plan.kp
        if budget_bytes is None:
            budget_bytes = device_budget(self.memory_fraction)
        k = min(self.nbest, max(1, n_cohort - self.nbest_discard))
        mp = plan_snorm_memory(n_enrol, n_cohort, n_test, k, row_bytes, budget_bytes, self.test_chunk,
                               materialise=materialise, tile_arrays=4 if fused else 8, exact=self.exact_topk,
                               x_dim=self.plan.d_in, tile_bytes=self.tile_bytes,
                               resident_allocated=resident_allocated)
        self.last_memory_plan = mp
        return mp

    # ---- embeddings ---------------------------------------------------------------------------
    def _bounds_of(self, G, side):
        """(|g|, |q|) of fp32 packed rows [g, q, 1] / [g, 1, q + c]: what the pair-writing prologue kernel
        returns beside its rows, for the fp32 operand format."""
        torch = engine.require_gpu()
        K = self.plan.k
        return torch.stack([G[:, :K].norm(dim=1), G[:, K if side == SIDE_ENROL else K + 1].abs()], dim=1).contiguous()

    def _embed(self, x, side, bounds=False):
        fmt = self.split_format
        if fmt is None:
            G = self.plan.prologue(x, side)
            return (G, self._bounds_of(G, side)) if bounds else G
        return self.plan.prologue_split(x, side, fmt, bounds=bounds)

    def _embed_both(self, x, bounds=False):
        fmt = self.split_format
        if fmt is None:
            Ge, Gt = self.plan.prologue(x, SIDE_ENROL), self.plan.prologue(x, SIDE_TEST)
            return (Ge, Gt, self._bounds_of(Ge, SIDE_ENROL)) if bounds else (Ge, Gt)
        return self.plan.prologue_split_both(x, fmt, bounds=bounds)

    def _lazy_ok(self):
        """Whether per-row bounds come with the fp32 prologue (then float64 rows are made only where needed)."""
        return self.split_format is None or self.plan.pairs_in_prologue(self.split_format)

    def embed(self, x, side):
        """Raw vectors (numpy or device tensor) -> packed device rows for one side of a trial."""
        return self._embed(engine.upload(x), side)

    def score(self, x_enrol, x_test, out=None):
        """Raw PLDA log-likelihood ratios, device [n_enrol, n_test]."""
        return engine.score(self.embed(x_enrol, SIDE_ENROL), self.embed(x_test, SIDE_TEST), out=out, split=False)

    # ---- scoring with adaptive S-norm -----------------------------------------------------------
    def prepare_cohort(self, x_enrol, x_cohort, rank=0, world=1, group=None):
        """Everything that depends only on enrolment and cohort: packed rows, the enrol-vs-cohort
        panel A_ec [n_enrol, n_cohort] and the per-enrolment top-k cohort lists.  With world > 1 each
        rank selects the lists of its own slice of enrolment rows and one all-gather makes them
        global (the only inter-GPU exchange of the whole pipeline)."""
        xc = engine.upload(x_cohort)
        nc_all = xc.shape[0]
        # what stays resident for the whole job, checked before the panel is allocated
        self.memory_plan(len(x_enrol), nc_all, 0, materialise=False)
        # fp64 rows only for the rows whose top-k boundary is in doubt (see topk_rows): needs the bounds
        # the pair-writing prologue kernel gives and the register-resident selection kernel
        lazy = self.exact_topk and self.lazy_exact and self._lazy_ok() and nc_all <= LAZY_MAX_COHORT
        xe = engine.upload(x_enrol)
        be_rows = None
        if lazy:
            Ge, be_rows = self._timed("prologue", lambda: self._embed(xe, SIDE_ENROL, bounds=True))
        else:
            Ge = self._timed("prologue", lambda: self._embed(xe, SIDE_ENROL))
        Gc_test = self._timed("prologue", lambda: self._embed(xc, SIDE_TEST))
        A_ec = self._timed("score_enrol_cohort", lambda: engine.score(Ge, Gc_test, split=False))
        ne, nc = A_ec.shape
        assert self.nbest_discard < nc
        k = min(self.nbest, nc - self.nbest_discard)
        counts = [8 * (shard_range(ne, r, world)[1] - shard_range(ne, r, world)[0]) for r in range(world)]
        lo, hi = shard_range(ne, rank, world)
        exact, coh64 = None, None
        if self.exact_topk and self.topk_counters is None:
            import torch

            # [0] rows whose boundary was re-decided in float64, [1] rows where that was given up (more
            # than 64 cohort scores inside the fp32 uncertainty band of the boundary: those rows keep
            # the fp32 tie order); totals over the life of this object, see topk_report()
            self.topk_counters = torch.zeros(2, dtype=torch.int64, device="cuda")
        if self.exact_topk:
            Gc64, bc = self.plan.prologue_f64(engine.upload64(x_cohort), SIDE_TEST)
            coh64 = {"G_coh": Gc64, "coh_gmax": float(bc[:, 0].max()), "coh_qmax": float(bc[:, 1].max()),
                     "counters_dev": self.topk_counters}
            if lazy:
                exact = dict(coh64, bnd_rows=be_rows[lo:hi].contiguous(),
                             rows64=lambda rows: self.plan.prologue_f64(_take64(x_enrol, lo, hi, rows), SIDE_ENROL))
            else:
                Ge64, be = self.plan.prologue_f64(engine.upload64(x_enrol[lo:hi]), SIDE_ENROL)
                exact = dict(coh64, G_rows=Ge64, bnd_rows=be)
        local = self._timed("topk_enrol", lambda: topk_rows(A_ec[lo:hi], k, self.nbest_discard, exact))
        local = local[:, :hi - lo]  # [k/8, rows, 8]
        e_idx = all_gather_lists(local.reshape(local.shape[0], -1), counts, group)
        e_idx = e_idx.reshape(e_idx.shape[0], -1, 8).contiguous()
        return {"Ge": Ge, "Gc_test": Gc_test, "A_ec": A_ec, "e_idx": e_idx, "k": k, "coh64": coh64, "lazy": lazy}

    def score_snorm(self, x_test, cohort, out=None, reduce=None):
        """AS-normalised scores of every enrolment model against `x_test` (this rank's shard).
        cohort: result of prepare_cohort.  Returns a device [n_enrol, n_test] tensor, or, when
        `reduce` is given, calls reduce(col0, block) on every [n_enrol, chunk] block instead of
        keeping the matrix (for trial sets whose score matrix does not fit); when the enrolment rows
        are walked in blocks (see tile_bytes) the call is reduce(col0, block, row0)."""
        from . import _lib

        torch = engine.require_gpu()
        lib = _lib.load()
        xt = engine.upload(x_test)
        nt = xt.shape[0]
        Ge, A_ec, e_idx, k = cohort["Ge"], cohort["A_ec"], cohort["e_idx"], cohort["k"]
        ne = Ge.shape[0]
        segmented = seg_plan(A_ec.shape[1], k) is not None  # the cohort is taken in segments: pairs only
        fused = self.fuse_apply and (gather_pairs_supported(A_ec.shape[1], ne, k) if segmented
                                     else gather_apply_supported(A_ec.shape[1], ne))
        # tile shape from the memory this rank still has (the cohort panel and lists are already resident),
        # before the result or anything else is allocated
        mp = self.memory_plan(ne, A_ec.shape[1], nt, materialise=out is None and reduce is None,
                              resident_allocated=True, fused=fused and (self.pair_stats or segmented))
        test_chunk, eb = mp["test_chunk"], mp["enrol_block"]
        if out is None and reduce is None:
            ld = engine.round_up(max(nt, 1), 32)
            out = torch.empty((ne, ld), dtype=torch.float32, device=xt.device)[:, :nt]
        for c0 in range(0, nt, test_chunk):
            c1 = min(nt, c0 + test_chunk)
            xc = xt[c0:c1]
            T = self._timed
            # tests as rows of the cohort panel and as columns of the trial matrix
            exact = None
            if cohort.get("coh64") is not None and cohort.get("lazy"):
                Gt_rows, Gt_cols, bt = T("prologue", lambda: self._embed_both(xc, bounds=True))
                exact = dict(cohort["coh64"], bnd_rows=bt, defer=True,
                             rows64=lambda rows: self.plan.prologue_f64(_take64(x_test, c0, c1, rows), SIDE_ENROL))
            else:
                Gt_rows, Gt_cols = T("prologue", lambda: self._embed_both(xc))
            A_tc = T("score_test_cohort", lambda: engine.score(Gt_rows, cohort["Gc_test"], split=False))  # [chunk, cohort]
            if cohort.get("coh64") is not None and exact is None:
                def f64_rows():
                    Gt64, bt = self.plan.prologue_f64(engine.upload64(x_test[c0:c1]), SIDE_ENROL)
                    return dict(cohort["coh64"], G_rows=Gt64, bnd_rows=bt)
                exact = T("prologue_f64", f64_rows)
            # The matrix-core kernels of a chunk stand together at its start: each of them pulls the clock down,
            # and whatever starts behind one runs its first half millisecond at that clock -- with the trial
            # scores (one enrolment block) in front of the selection only the selection, which mostly waits for
            # memory, starts at the low clock, and both gathers run at the full one.
            S_first = None
            if self.trials_before_topk and eb >= ne:
                S_first = T("score_trials", lambda: engine.score(Ge, Gt_cols, split=False))           # [enrol, chunk]
            t_idx = T("topk_test", lambda: topk_rows(A_tc, k, self.nbest_discard, exact))  # [k, chunk]
            nq = c1 - c0
            for e0 in range(0, ne, eb):
                e1 = min(ne, e0 + eb)
                # (several enrolment blocks: the trial scores of a block before its two gathers, so that no
                # matrix-core kernel runs between them)
                if S_first is not None:
                    S, S_first = S_first, None
                else:
                    S = T("score_trials", lambda: engine.score(_rows(Ge, e0, e1), Gt_cols, split=False))
                paired = fused and (self.pair_stats or segmented)
                if paired:
                    mt, ldt = T("gather_t", lambda: gather_stats_pairs(A_tc, e_idx[:, e0:e1], e1 - e0, 1e-5, k),
                                (nq, e1 - e0, k))                                         # [chunk, enrol, 2]
                    it = None
                else:
                    mt, it, ldt = T("gather_t", lambda: gather_stats(A_tc, None, e_idx[:, e0:e1], e1 - e0, 1e-5, k),
                                    (nq, e1 - e0, k))                                     # [chunk, enrol]
                if exact is not None and "finish" in exact:
                    # the tests' lists are first needed by the Z part: the host's wait for the number of rows
                    # in doubt is spent behind the T part's launch
                    T("topk_test", exact.pop("finish"))
                if reduce is None:
                    dst = out[e0:e1, c0:c1]
                else:
                    dst = torch.empty((e1 - e0, engine.round_up(nq, 32)), dtype=torch.float32,
                                      device=xt.device)[:, :nq]
                if paired:
                    T("gather_z", lambda: gather_apply_pairs(A_ec[e0:e1], t_idx, nq, 1e-5, k, S, mt, ldt, dst),
                      (e1 - e0, nq, k))
                elif fused:
                    # Z-part statistics and the normalisation in one launch: the statistics never leave
                    # the registers
                    T("gather_z", lambda: gather_apply(A_ec[e0:e1], t_idx, nq, 1e-5, k, S, mt, it, ldt, dst),
                      (e1 - e0, nq, k))
                else:
                    mz, iz, ldz = T("gather_z", lambda: gather_stats(A_ec[e0:e1], None, t_idx, nq, 1e-5, k),
                                    (e1 - e0, nq, k))                                     # [enrol, chunk]
                    T("apply", lambda: _lib.check(lib.hypmi_asnorm_apply(
                        engine._ptr(S), S.stride(0), e1 - e0, nq, engine._ptr(mz), engine._ptr(iz), ldz,
                        engine._ptr(mt), engine._ptr(it), ldt, engine._ptr(dst), dst.stride(0),
                        engine.stream_ptr())))
                    del mz, iz
                if reduce is not None:
                    if eb >= ne:
                        reduce(c0, dst)
                    else:
                        reduce(c0, dst, e0)
                del mt, it, S, dst
        if self.exact_topk:
            self._warn_skipped()  # one read of two counters per call
        return out
