import numpy as np

from .. import _lib, engine
from ..hyp_defs import float_cpu
from .score_norm import ScoreNorm, dev_mask, dev_matrix, transpose, transpose_mask, zero_masked


def topk_rows(A, k, discard, exact=None):
    """Device [rows, cohort] -> grouped uint16 index lists [ceil(k/8), rows, 8] (stored as int16):
    entry j of row q at [j >> 3, q, j & 7]; cohort indices ranked discard..discard+k-1 per row.

    exact: optional source of float64 values used to re-decide members that fp32 cannot separate
    from a selection boundary: {"A64": float64 device matrix A was rounded from} or
    {"G_rows", "G_coh", "bnd_rows", "coh_gmax", "coh_qmax"} (fp64 packed rows of Plan.prologue_f64).
    Instead of "G_rows" the dict may hold "rows64", a function (device int32 row numbers) ->
    (fp64 packed rows, bounds) of just those rows, and "bnd_rows" from the fp32 prologue
    (Plan.prologue_split(..., bounds=True)): the fp64 rows are then made only for the few per cent of
    rows whose boundary has a second score inside its fp32 uncertainty band (one host read of their
    count in between; the dict receives "redone" = that count).  With exact["defer"] = True that read and
    the second selection are left to exact["finish"](), which the caller runs before it uses the lists:
    work launched in between hides the host's wait.
    With exact["count"] = True the dict receives "counters" = (rows re-decided, rows skipped);
    exact["counters_dev"], a device int64[2], is added to instead and never read here (no host
    synchronisation: for callers that look at the totals once at the end)."""
    import ctypes

    torch = engine.require_gpu()
    lib = _lib.load()
    n, nc = A.shape
    ldi = max(n, 1)
    idx = torch.empty(((k + 7) // 8, ldi, 8), dtype=torch.int16, device=A.device)
    ws_bytes = lib.hypmi_topk_workspace_bytes(n, nc)
    ws = torch.empty(max(ws_bytes // 4, 4), dtype=torch.int32, device=A.device)
    ex_ptr, counters = None, None
    if exact is not None:
        ex = _lib.TopkExact()
        if exact.get("counters_dev") is not None:
            ex.counters = exact["counters_dev"].data_ptr()
        elif exact.get("count"):
            counters = torch.zeros(2, dtype=torch.int64, device=A.device)
            ex.counters = counters.data_ptr()
        if "A64" in exact:
            A64 = exact["A64"]
            assert A64.dtype == torch.float64 and tuple(A64.shape) == (n, nc) and A64.stride(1) == 1
            ex.mode, ex.A64, ex.lda64 = 1, A64.data_ptr(), A64.stride(0)
        else:
            Gc = exact["G_coh"]
            assert Gc.shape[0] == nc
            ex.G_coh, ex.ldg, ex.K = Gc.data_ptr(), Gc.stride(0), Gc.shape[1] - 2
            bnd = exact["bnd_rows"]
            assert bnd.dtype == torch.float32 and tuple(bnd.shape) == (n, 2) and bnd.is_contiguous()
            ex.bnd_rows = bnd.data_ptr()
            ex.coh_gmax, ex.coh_qmax = float(exact["coh_gmax"]), float(exact["coh_qmax"])
            ex.err_coeff = float(exact.get("err_coeff", engine.ERR_COEFF))
            if "rows64" in exact:
                todo = torch.empty(max(n, 1), dtype=torch.int32, device=A.device)
                todo_count = torch.zeros(1, dtype=torch.int32, device=A.device)
                ex.mode, ex.todo, ex.todo_count = 3, todo.data_ptr(), todo_count.data_ptr()
            else:
                Gr = exact["G_rows"]
                assert Gr.shape[0] == n and Gr.shape[1] == Gc.shape[1] and Gr.stride(0) == Gc.stride(0)
                ex.mode, ex.G_rows = 2, Gr.data_ptr()
        ex_ptr = ctypes.byref(ex)
    _lib.check(lib.hypmi_topk_rows(engine._ptr(A), n, nc, A.stride(0), k, discard, engine._ptr(idx),
                                   ldi, ex_ptr, engine._ptr(ws), ws_bytes, engine.stream_ptr()))
    if exact is not None and ex.mode == 3:
        # the number of rows that need exact values travels to a page-locked word behind the selection;
        # the host waits for that copy alone, not for what the caller may have launched after it
        count_host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        count_host.copy_(todo_count, non_blocking=True)
        copied = torch.cuda.Event()
        copied.record()

        def finish():
            copied.synchronize()
            m = int(count_host[0])
            exact["redone"] = m
            if m:
                rows = torch.sort(todo[:m]).values
                Gr, bnd_sub = exact["rows64"](rows)
                assert Gr.shape[0] == m and Gr.shape[1] == Gc.shape[1] and Gr.stride(0) == Gc.stride(0)
                assert bnd_sub.dtype == torch.float32 and tuple(bnd_sub.shape) == (m, 2) and bnd_sub.is_contiguous()
                ex.mode, ex.G_rows, ex.bnd_rows, ex.row_map = 2, Gr.data_ptr(), bnd_sub.data_ptr(), rows.data_ptr()
                ex.todo = ex.todo_count = None
                _lib.check(lib.hypmi_topk_rows(engine._ptr(A), m, nc, A.stride(0), k, discard, engine._ptr(idx),
                                               ldi, ctypes.byref(ex), engine._ptr(ws), ws_bytes, engine.stream_ptr()))

        if exact.get("defer"):
            exact["finish"] = finish  # the caller runs it before the lists are used
            return idx
        finish()
    if counters is not None:
        exact["counters"] = tuple(int(v) for v in counters.cpu())
    return idx


def lists_to_numpy(idx, k):
    """Grouped device lists -> host int array [rows, k] (for tests and debugging)."""
    a = idx.cpu().numpy().view("uint16").astype("int64")  # [k8, rows, 8]
    return a.transpose(1, 0, 2).reshape(a.shape[1], -1)[:, :k]


class SegLists:
    """Top-k lists partitioned by cohort segment for the statistics kernels that take the cohort in
    pieces (cohorts whose rows do not fit LDS two at a time, see hypmi_gather_seg_plan): the lists as
    selected (`idx`), the partitioned copy (`seg`, members as 1 + offset in their segment) and the first
    group of every segment of every list (`gstart`)."""

    def __init__(self, idx, seg, gstart, k, nc, n_lists):
        self.idx, self.seg, self.gstart, self.k, self.nc, self.n_lists = idx, seg, gstart, k, nc, n_lists


def seg_plan(nc, k):
    """(segments, members per segment, groups per list) when a cohort of nc members is taken in segments,
    else None (it fits LDS two rows at a time, or has more than 65 535 members)."""
    import ctypes

    a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    if _lib.load().hypmi_gather_seg_plan(int(nc), int(k), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)):
        return a.value, b.value, c.value
    return None


def segment_lists(idx, k, nc, n_lists):
    """SegLists of grouped lists `idx` (or of a range of them, `idx[:, q0:q1]`).  Kept on the tensor
    object: the enrolment side's lists are partitioned once and used for every chunk of tests."""
    torch = engine.require_gpu()
    cached = getattr(idx, "_hypmi_seg", None)
    if cached is not None and (cached.k, cached.nc, cached.n_lists) == (k, nc, n_lists):
        return cached
    S, _, groups = seg_plan(nc, k)
    assert idx.stride(2) == 1 and idx.stride(1) == 8 and idx.shape[1] >= n_lists
    ldi = idx.stride(0) // 8
    seg = torch.empty((groups, ldi, 8), dtype=torch.int16, device=idx.device)
    gstart = torch.empty((S + 1, ldi), dtype=torch.int16, device=idx.device)
    _lib.check(_lib.load().hypmi_segment_lists(engine._ptr(idx), ldi, int(k), n_lists, int(nc), engine._ptr(seg),
                                               engine._ptr(gstart), engine.stream_ptr()))
    sl = SegLists(idx, seg, gstart, k, nc, n_lists)
    try:
        idx._hypmi_seg = sl
    except AttributeError:
        pass
    return sl


def _stats_ld(n_lists, pairs):
    ld = engine.round_up(max(n_lists, 1), 4)
    if ld % (256 if pairs else 512) == 0:
        ld += 16 if pairs else 32  # read down its columns as well: no stride that maps a column to one memory channel
    return ld


def gather_stats(X, mask_u8, idx, n_lists, std_floor, k):
    """mean / inverse std of X[r, list_q] for every row r and list q: two device [rows, lists].
    idx: grouped lists as topk_rows makes them, or a range of their lists (`idx[:, q0:q1]`)."""
    torch = engine.require_gpu()
    lib = _lib.load()
    n, nc = X.shape
    assert idx.stride(2) == 1 and idx.stride(1) == 8 and idx.shape[1] >= n_lists
    ld = _stats_ld(n_lists, False)
    mean = torch.empty((n, ld), dtype=torch.float32, device=X.device)
    istd = torch.empty((n, ld), dtype=torch.float32, device=X.device)
    if mask_u8 is None and seg_plan(nc, k) is not None:
        sl = segment_lists(idx, k, nc, n_lists)
        _lib.check(lib.hypmi_gather_stats_seg(engine._ptr(X), n, nc, X.stride(0), engine._ptr(idx), engine._ptr(sl.seg),
                                              engine._ptr(sl.gstart), idx.stride(0) // 8, int(k), n_lists,
                                              float(std_floor), engine._ptr(mean), engine._ptr(istd), ld,
                                              engine.stream_ptr()))
        return mean, istd, ld
    _lib.check(lib.hypmi_gather_stats(engine._ptr(X), n, nc, X.stride(0),
                                      engine._ptr(mask_u8) if mask_u8 is not None else None,
                                      mask_u8.stride(0) if mask_u8 is not None else 0,
                                      engine._ptr(idx), idx.stride(0) // 8, int(k), n_lists,
                                      float(std_floor), engine._ptr(mean), engine._ptr(istd), ld,
                                      engine.stream_ptr()))
    return mean, istd, ld


def gather_stats_pairs(X, idx, n_lists, std_floor, k):
    """gather_stats (unmasked, rows that fit in LDS) with mean and inverse std of a (row, list) side by side:
    one device array [rows, ld, 2] and ld (in pairs)."""
    torch = engine.require_gpu()
    n, nc = X.shape
    assert idx.stride(2) == 1 and idx.stride(1) == 8 and idx.shape[1] >= n_lists
    ld = _stats_ld(n_lists, True)
    stats = torch.empty((n, ld, 2), dtype=torch.float32, device=X.device)
    if seg_plan(nc, k) is not None:
        sl = segment_lists(idx, k, nc, n_lists)
        _lib.check(_lib.load().hypmi_gather_stats_seg(engine._ptr(X), n, nc, X.stride(0), engine._ptr(idx),
                                                      engine._ptr(sl.seg), engine._ptr(sl.gstart), idx.stride(0) // 8,
                                                      int(k), n_lists, float(std_floor), engine._ptr(stats), None, ld,
                                                      engine.stream_ptr()))
        return stats, ld
    _lib.check(_lib.load().hypmi_gather_stats_pairs(engine._ptr(X), n, nc, X.stride(0), engine._ptr(idx),
                                                    idx.stride(0) // 8, int(k), n_lists, float(std_floor),
                                                    engine._ptr(stats), ld, engine.stream_ptr()))
    return stats, ld


def gather_apply_pairs(X, idx, n_lists, std_floor, k, S, stats_t, ldt, out):
    """gather_apply with the T-part statistics as the pairs gather_stats_pairs makes ([test, ldt, 2])."""
    n, nc = X.shape
    assert idx.stride(2) == 1 and idx.stride(1) == 8 and idx.shape[1] >= n_lists
    assert S.shape[0] == n and S.shape[1] >= n_lists and out.shape[0] == n and out.shape[1] >= n_lists
    assert stats_t.shape[0] >= n_lists and ldt >= n
    if seg_plan(nc, k) is not None:
        sl = segment_lists(idx, k, nc, n_lists)
        _lib.check(_lib.load().hypmi_gather_apply_seg(engine._ptr(X), n, nc, X.stride(0), engine._ptr(idx),
                                                      engine._ptr(sl.seg), engine._ptr(sl.gstart), idx.stride(0) // 8,
                                                      int(k), n_lists, float(std_floor), engine._ptr(S), S.stride(0),
                                                      engine._ptr(stats_t), ldt, engine._ptr(out), out.stride(0),
                                                      engine.stream_ptr()))
        return out
    _lib.check(_lib.load().hypmi_gather_apply_pairs(engine._ptr(X), n, nc, X.stride(0), engine._ptr(idx),
                                                    idx.stride(0) // 8, int(k), n_lists, float(std_floor),
                                                    engine._ptr(S), S.stride(0), engine._ptr(stats_t), ldt,
                                                    engine._ptr(out), out.stride(0), engine.stream_ptr()))
    return out


def gather_apply_supported(nc, n_rows):
    """Whether gather_apply (the other side's statistics in two arrays) can run for a cohort of nc members:
    its rows have to fit in LDS undivided."""
    return bool(_lib.load().hypmi_gather_apply_supported(int(nc), int(n_rows)))


def gather_pairs_supported(nc, n_rows, k):
    """Whether gather_stats_pairs / gather_apply_pairs can run: the rows fit in LDS, undivided or in segments."""
    return gather_apply_supported(nc, n_rows) or seg_plan(nc, k) is not None


def gather_apply(X, idx, n_lists, std_floor, k, S, mt, it, ldt, out):
    """Z-part statistics of X [enrol, cohort] over the tests' lists and the normalisation in one launch:
    out[e, t] from the raw scores S [enrol, test] and the T-part statistics mt / it [test, enrol]
    (leading dimension ldt).  Same values as gather_stats followed by hypmi_asnorm_apply."""
    lib = _lib.load()
    n, nc = X.shape
    assert idx.stride(2) == 1 and idx.stride(1) == 8 and idx.shape[1] >= n_lists
    assert S.shape[0] == n and S.shape[1] >= n_lists and out.shape[0] == n and out.shape[1] >= n_lists
    assert mt.shape[0] >= n_lists and it.shape[0] >= n_lists and ldt >= n
    _lib.check(lib.hypmi_gather_apply(engine._ptr(X), n, nc, X.stride(0), engine._ptr(idx), idx.stride(0) // 8,
                                      int(k), n_lists, float(std_floor), engine._ptr(S), S.stride(0),
                                      engine._ptr(mt), engine._ptr(it), ldt, engine._ptr(out), out.stride(0),
                                      engine.stream_ptr()))
    return out


# fp32 error estimate of a normalised score, in units of 2^-24 times the magnitudes that enter it (see
# hypmi_asnorm_refine): the largest error seen over the parity problems is a fifth of the estimate.
REFINE_ERR_COEFF = 2.0 ** -24


def adapt_s_norm_device(S, A_ec, A_tc, nbest, nbest_discard, mask_ec=None, mask_tc=None,
                        std_floor=1e-5, exact_ec=None, exact_tc=None, refine=None):
    """Adaptive S-norm on device matrices: S [enrol, test], A_ec [enrol, cohort], A_tc [test, cohort]
    (already zeroed where masked).  exact_ec / exact_tc: see topk_rows.  Returns a device
    [enrol, test] tensor.

    refine (unmasked only): {"S64", "E64", "T64"} = the float64 device matrices the three fp32 ones were
    rounded from, optional "rel_tol" (0.5e-4) and "err_coeff": entries whose fp32 error estimate exceeds
    rel_tol * |score| are recomputed from them over the same selections (hypmi_asnorm_refine); the dict
    receives "refined" = their number."""
    torch = engine.require_gpu()
    lib = _lib.load()
    ne, nt = S.shape
    nc = A_ec.shape[1]
    assert A_tc.shape[1] == nc and A_ec.shape[0] == ne and A_tc.shape[0] == nt
    assert nbest_discard < nc
    k = min(int(nbest), nc - int(nbest_discard))
    ld = engine.round_up(max(nt, 1), 32)
    out = torch.empty((ne, ld), dtype=torch.float32, device=S.device)
    if ne == 0 or nt == 0:
        return out[:, :nt]
    t_idx = topk_rows(A_tc, k, int(nbest_discard), exact_tc)  # per test: cohort ranked by coh-vs-test
    e_idx = topk_rows(A_ec, k, int(nbest_discard), exact_ec)  # per enrol: cohort ranked by enr-vs-coh
    if refine is not None:
        assert mask_ec is None and mask_tc is None, "refinement of masked scores is not available"
        S64, E64, T64 = refine["S64"], refine["E64"], refine["T64"]
        for a, b in ((S64, S), (E64, A_ec), (T64, A_tc)):
            assert a.dtype == torch.float64 and tuple(a.shape) == tuple(b.shape) and a.stride(1) == 1
        mt, it, ldt = gather_stats(A_tc, None, e_idx, ne, std_floor, k)  # [test, enrol]
        mz, iz, ldz = gather_stats(A_ec, None, t_idx, nt, std_floor, k)  # [enrol, test]
        _lib.check(lib.hypmi_asnorm_apply(engine._ptr(S), S.stride(0), ne, nt, engine._ptr(mz), engine._ptr(iz), ldz,
                                          engine._ptr(mt), engine._ptr(it), ldt, engine._ptr(out), ld,
                                          engine.stream_ptr()))
        counter = torch.zeros(1, dtype=torch.int64, device=S.device)
        _lib.check(lib.hypmi_asnorm_refine(
            engine._ptr(S64), S64.stride(0), engine._ptr(E64), E64.stride(0), engine._ptr(T64), T64.stride(0), ne, nt,
            nc, engine._ptr(t_idx), t_idx.stride(0) // 8, engine._ptr(e_idx), e_idx.stride(0) // 8, k, engine._ptr(mz),
            engine._ptr(iz), ldz, engine._ptr(mt), engine._ptr(it), ldt, float(refine.get("rel_tol", 0.5e-4)),
            float(refine.get("err_coeff", REFINE_ERR_COEFF)), float(std_floor), engine._ptr(out), ld,
            engine._ptr(counter), engine.stream_ptr()))
        refine["refined"] = int(counter.item())
        return out[:, :nt]
    if mask_ec is None and mask_tc is None and gather_pairs_supported(nc, min(ne, nt), k):
        # T-part mean and 1 / std side by side: one line per list in the Z part
        st, ldt = gather_stats_pairs(A_tc, e_idx, ne, std_floor, k)  # [test, enrol, 2]
        gather_apply_pairs(A_ec, t_idx, nt, std_floor, k, S, st, ldt, out)
        return out[:, :nt]
    fused = mask_ec is None and gather_apply_supported(nc, ne) and seg_plan(nc, k) is None
    mt, it, ldt = gather_stats(A_tc, mask_tc, e_idx, ne, std_floor, k)  # [test, enrol]
    if fused:
        gather_apply(A_ec, t_idx, nt, std_floor, k, S, mt, it, ldt, out)
        return out[:, :nt]
    mz, iz, ldz = gather_stats(A_ec, mask_ec, t_idx, nt, std_floor, k)  # [enrol, test]
    _lib.check(lib.hypmi_asnorm_apply(engine._ptr(S), S.stride(0), ne, nt, engine._ptr(mz),
                                      engine._ptr(iz), ldz, engine._ptr(mt), engine._ptr(it), ldt,
                                      engine._ptr(out), ld, engine.stream_ptr()))
    return out[:, :nt]


class AdaptSNorm(ScoreNorm):
    """Adaptive S-norm (hyperion/score_norm/adapt_s_norm.py:13-88): Z- and T-normalisation with
    statistics over the `nbest` cohort members closest to each test / enrolment side, after
    dropping the `nbest_discard` closest.  The std floor is the reference's literal 1e-5 (its
    `std_floor` attribute is not consulted there either).  Caller arrays are never modified;
    a tie exactly at the selection boundary goes to the lower cohort index (numpy's argsort leaves
    it unspecified).

    refine: the reference works in float64; the device normalises in fp32, which leaves a normalised
    score near zero (most non-target trials) with an absolute error of a few 1e-6 and no relative accuracy.
    With refine=True such entries are recomputed from the float64 inputs on the device, so that every score
    is within 1e-4 RELATIVE of the reference's; None (default) = whenever the three score matrices arrive as
    float64 host arrays without masks.  `last_refined` holds the number of entries recomputed."""

    def __init__(self, nbest=100, nbest_discard=0, refine=None, **kwargs):
        super().__init__(**kwargs)
        self.nbest = nbest
        self.nbest_discard = nbest_discard
        self.refine = refine
        self.last_refined = None

    def predict(self, scores, scores_coh_test, scores_enr_coh, mask_coh_test=None, mask_enr_coh=None):
        torch = engine.require_gpu()
        host_ec = not isinstance(scores_enr_coh, torch.Tensor)
        host_ct = not isinstance(scores_coh_test, torch.Tensor)
        wide = lambda a: np.asarray(a).dtype.itemsize > 4  # noqa: E731  (fp32 arrays lose nothing on the device)
        # Cohort scores that arrive as host arrays are ranked in fp32 on the device; members whose fp32
        # roundings collide at a selection boundary are re-decided from the float64 values.  Those go
        # to the device once, as float64, and are narrowed there.
        exact_ec = exact_tc = None
        m_ec = m_tc = None
        all64 = (mask_enr_coh is None and mask_coh_test is None and host_ec and host_ct
                 and not isinstance(scores, torch.Tensor) and wide(scores) and wide(scores_enr_coh)
                 and wide(scores_coh_test))
        refine = all64 if self.refine is None else bool(self.refine)
        if refine and not all64:
            raise ValueError("AdaptSNorm(refine=True) needs the three score matrices as float64 host arrays, unmasked")
        self.last_refined = None
        if refine:
            S64 = engine.upload64(scores)
            S = engine.upload(S64)
        else:
            S = dev_matrix(scores)
        if mask_enr_coh is None and host_ec and wide(scores_enr_coh):
            E64 = engine.upload64(scores_enr_coh)
            EC = engine.upload(E64)
            exact_ec = {"A64": E64, "count": True}
        else:
            EC = dev_matrix(scores_enr_coh)
        if mask_coh_test is None and host_ct and wide(scores_coh_test):
            T64 = engine.upload64(scores_coh_test).t().contiguous()  # [test, cohort]
            TC = engine.upload(T64)
            exact_tc = {"A64": T64, "count": True}
        else:
            CT = dev_matrix(scores_coh_test)
            assert CT.shape[1] == S.shape[1]
            if mask_coh_test is not None:
                m_ct = dev_mask(mask_coh_test, CT.shape)
                CT = CT.clone() if not host_ct else CT
                zero_masked(CT, m_ct)
                m_tc = transpose_mask(m_ct)
            TC = transpose(CT)
        assert EC.shape[1] == TC.shape[1]
        assert self.nbest_discard < EC.shape[1]
        assert EC.shape[0] == S.shape[0] and TC.shape[0] == S.shape[1]
        if mask_enr_coh is not None:
            m_ec = dev_mask(mask_enr_coh, EC.shape)
            EC = EC.clone() if not host_ec else EC
            zero_masked(EC, m_ec)
        ref = {"S64": S64, "E64": E64, "T64": T64} if refine else None
        out = adapt_s_norm_device(S, EC, TC, self.nbest, self.nbest_discard, m_ec, m_tc,
                                  exact_ec=exact_ec, exact_tc=exact_tc, refine=ref)
        if ref is not None:
            self.last_refined = ref["refined"]
        skipped = sum(e["counters"][1] for e in (exact_ec, exact_tc) if e is not None and "counters" in e)
        if skipped:
            import warnings

            # more than 64 cohort scores tie with a selection boundary in fp32 (dense or quantised
            # scores): those rows keep the fp32 tie order instead of the float64 ranking
            warnings.warn("AdaptSNorm: %d cohort selection(s) could not be re-decided from the float64 scores "
                          "(more than 64 scores inside the fp32 uncertainty band of the boundary)" % skipped,
                          RuntimeWarning)
        if isinstance(scores, torch.Tensor):
            return out
        return engine.to_host(out, float_cpu())
