"""ctypes binding of libhypmi.so (the C ABI declared in include/hypmi.h).

There is no CPU fallback: if the shared library has not been built, or no HIP device is present
when a compute entry point is called, the caller gets an exception."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhypmi.so")

HYPMI_MAX_SEG = 4
SIDE_ENROL = 0
SIDE_TEST = 1

c_double_p = ctypes.POINTER(ctypes.c_double)
c_void_p = ctypes.c_void_p
i64 = ctypes.c_int64
i32 = ctypes.c_int32


class HypmiError(RuntimeError):
    pass


class Segment(ctypes.Structure):
    _fields_ = [("A", c_double_p), ("b", c_double_p), ("d_out", i32), ("lnorm", i32)]


class TopkExact(ctypes.Structure):
    _fields_ = [
        ("mode", i32),
        ("A64", c_void_p),
        ("lda64", i64),
        ("G_rows", c_void_p),
        ("G_coh", c_void_p),
        ("ldg", i64),
        ("K", i32),
        ("bnd_rows", c_void_p),
        ("coh_gmax", ctypes.c_float),
        ("coh_qmax", ctypes.c_float),
        ("err_coeff", ctypes.c_float),
        ("counters", c_void_p),
        ("row_map", c_void_p),
        ("todo", c_void_p),
        ("todo_count", c_void_p),
    ]


class PlanDesc(ctypes.Structure):
    _fields_ = [
        ("d_in", i32),
        ("n_seg", i32),
        ("seg", Segment * HYPMI_MAX_SEG),
        ("At", c_double_p),
        ("bt", c_double_p),
        ("An", c_double_p),
        ("bn", c_double_p),
        ("K", i32),
        ("c", ctypes.c_double),
        ("shift_t", c_double_p),
        ("scale_t", c_double_p),
        ("shift_n", c_double_p),
        ("scale_n", c_double_p),
    ]


# name -> (restype, argtypes); every symbol include/hypmi.h declares
SIGNATURES = {
    "hypmi_last_error": (ctypes.c_char_p, []),
    "hypmi_version": (ctypes.c_int, []),
    "hypmi_plan_create": (ctypes.c_int, [ctypes.POINTER(PlanDesc), ctypes.POINTER(c_void_p)]),
    "hypmi_plan_destroy": (None, [c_void_p]),
    "hypmi_plan_kp": (ctypes.c_int, [c_void_p]),
    "hypmi_plan_k": (ctypes.c_int, [c_void_p]),
    "hypmi_plan_d_in": (ctypes.c_int, [c_void_p]),
    "hypmi_plan_d_out": (ctypes.c_int, [c_void_p]),
    "hypmi_prologue_workspace_bytes": (i64, [c_void_p, i64]),
    "hypmi_prologue": (ctypes.c_int, [c_void_p, c_void_p, i64, i64, ctypes.c_int, c_void_p, c_void_p,
                                      i64, c_void_p]),
    "hypmi_prologue_pairs_supported": (ctypes.c_int, [c_void_p]),
    "hypmi_prologue_pairs": (ctypes.c_int, [c_void_p, c_void_p, i64, i64, ctypes.c_int, c_void_p, c_void_p,
                                            c_void_p, c_void_p, i64, c_void_p]),
    "hypmi_prologue_pairs_bounds": (ctypes.c_int, [c_void_p, c_void_p, i64, i64, ctypes.c_int, c_void_p, c_void_p,
                                                   c_void_p, c_void_p, c_void_p, i64, c_void_p]),
    "hypmi_prologue_noalloc_max_rows": (i64, [c_void_p]),
    "hypmi_prologue_noalloc": (ctypes.c_int, [c_void_p, c_void_p, i64, i64, ctypes.c_int, c_void_p, c_void_p]),
    "hypmi_transform": (ctypes.c_int, [c_void_p, c_void_p, i64, i64, c_void_p, i64, c_void_p, i64,
                                       c_void_p]),
    "hypmi_score": (ctypes.c_int, [c_void_p, i64, c_void_p, i64, ctypes.c_int, c_void_p, i64,
                                   c_void_p]),
    "hypmi_split_row_bytes": (i64, [ctypes.c_int]),
    "hypmi_split_bytes": (i64, [ctypes.c_int, i64]),
    "hypmi_split_rows": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, c_void_p, c_void_p]),
    "hypmi_score_split": (ctypes.c_int, [c_void_p, i64, c_void_p, i64, ctypes.c_int, c_void_p, i64,
                                         c_void_p]),
    "hypmi_pair_row_bytes": (i64, [ctypes.c_int]),
    "hypmi_pair_bytes": (i64, [ctypes.c_int, i64]),
    "hypmi_pair_rows": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, c_void_p, c_void_p]),
    "hypmi_score_pair": (ctypes.c_int, [c_void_p, i64, c_void_p, i64, ctypes.c_int, c_void_p, i64,
                                         c_void_p]),
    "hypmi_affine": (ctypes.c_int, [c_void_p, i64, i64, ctypes.c_int, c_void_p, i64, c_void_p,
                                    ctypes.c_int, c_void_p, i64, c_void_p]),
    "hypmi_lnorm_rows": (ctypes.c_int, [c_void_p, i64, i64, ctypes.c_int, c_void_p]),
    "hypmi_prologue_f64_workspace_bytes": (i64, [c_void_p, i64]),
    "hypmi_prologue_f64": (ctypes.c_int, [c_void_p, c_void_p, i64, i64, ctypes.c_int, c_void_p, i64,
                                          c_void_p, c_void_p, i64, c_void_p]),
    "hypmi_refine_scores": (ctypes.c_int, [c_void_p, i64, i64, i64, c_void_p, c_void_p, i64,
                                           ctypes.c_int, c_void_p, c_void_p, ctypes.c_float,
                                           ctypes.c_float, c_void_p, c_void_p]),
    "hypmi_pool_rows": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, c_void_p, c_void_p, i64,
                                       ctypes.c_int, c_void_p, i64, c_void_p, c_void_p]),
    "hypmi_packed_width": (ctypes.c_int, [ctypes.c_int]),
    "hypmi_div_rows": (ctypes.c_int, [c_void_p, i64, i64, ctypes.c_int, c_void_p, c_void_p]),
    "hypmi_book_pack": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, c_void_p, i64, c_void_p, c_void_p,
                                       c_void_p, c_void_p, ctypes.c_int, ctypes.c_double, c_void_p,
                                       ctypes.c_int, c_void_p]),
    "hypmi_book_pack_f64": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, c_void_p, i64, c_void_p, c_void_p,
                                       c_void_p, c_void_p, ctypes.c_int, ctypes.c_double, c_void_p,
                                       ctypes.c_int, c_void_p]),
    "hypmi_book_pack_f64_rows": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, c_void_p, i64, c_void_p, c_void_p,
                                                c_void_p, c_void_p, ctypes.c_int, ctypes.c_double, c_void_p,
                                                ctypes.c_int, c_void_p, i64, c_void_p, c_void_p]),
    "hypmi_scatter_block": (ctypes.c_int, [c_void_p, i64, c_void_p, i64, c_void_p, i64, c_void_p, i64,
                                           c_void_p]),
    "hypmi_topk_workspace_bytes": (i64, [i64, ctypes.c_int]),
    "hypmi_topk_list_bytes": (i64, [ctypes.c_int, i64]),
    "hypmi_topk_rows": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, i64, ctypes.c_int, ctypes.c_int,
                                       c_void_p, i64, ctypes.POINTER(TopkExact), c_void_p, i64, c_void_p]),
    "hypmi_gather_stats": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, i64, c_void_p, i64, c_void_p,
                                          i64, ctypes.c_int, i64, ctypes.c_float, c_void_p, c_void_p,
                                          i64, c_void_p]),
    "hypmi_gather_apply_supported": (ctypes.c_int, [ctypes.c_int, i64]),
    "hypmi_gather_seg_plan": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                             ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "hypmi_segment_lists": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, i64, ctypes.c_int, c_void_p, c_void_p,
                                           c_void_p]),
    "hypmi_gather_stats_seg": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, i64, c_void_p, c_void_p, c_void_p, i64,
                                              ctypes.c_int, i64, ctypes.c_float, c_void_p, c_void_p, i64, c_void_p]),
    "hypmi_gather_apply_seg": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, i64, c_void_p, c_void_p, c_void_p, i64,
                                              ctypes.c_int, i64, ctypes.c_float, c_void_p, i64, c_void_p, i64,
                                              c_void_p, i64, c_void_p]),
    "hypmi_gather_apply": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, i64, c_void_p, i64, ctypes.c_int, i64,
                                          ctypes.c_float, c_void_p, i64, c_void_p, c_void_p, i64, c_void_p, i64,
                                          c_void_p]),
    "hypmi_gather_stats_pairs": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, i64, c_void_p, i64, ctypes.c_int, i64,
                                                ctypes.c_float, c_void_p, i64, c_void_p]),
    "hypmi_gather_apply_pairs": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, i64, c_void_p, i64, ctypes.c_int, i64,
                                                ctypes.c_float, c_void_p, i64, c_void_p, i64, c_void_p, i64,
                                                c_void_p]),
    "hypmi_gather_redo_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]),
    "hypmi_gather_clock": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.c_int]),
    "hypmi_asnorm_apply": (ctypes.c_int, [c_void_p, i64, i64, i64, c_void_p, c_void_p, i64, c_void_p,
                                          c_void_p, i64, c_void_p, i64, c_void_p]),
    "hypmi_asnorm_refine": (ctypes.c_int, [c_void_p, i64, c_void_p, i64, c_void_p, i64, i64, i64, ctypes.c_int,
                                           c_void_p, i64, c_void_p, i64, ctypes.c_int, c_void_p, c_void_p, i64,
                                           c_void_p, c_void_p, i64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                           c_void_p, i64, c_void_p, c_void_p]),
    "hypmi_mask_zero": (ctypes.c_int, [c_void_p, i64, i64, i64, c_void_p, i64, c_void_p]),
    "hypmi_transpose": (ctypes.c_int, [c_void_p, i64, i64, i64, c_void_p, i64, c_void_p]),
    "hypmi_row_meanstd": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, i64, c_void_p, i64,
                                         ctypes.c_float, c_void_p, c_void_p, c_void_p]),
    "hypmi_norm_apply": (ctypes.c_int, [c_void_p, i64, i64, i64, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, i64, c_void_p]),
    "hypmi_gram_f64": (ctypes.c_int, [c_void_p, i64, i64, ctypes.c_int, c_void_p, c_void_p, i64, c_void_p,
                                      c_void_p]),
    "hypmi_gram_f64_workspace_bytes": (i64, [i64, ctypes.c_int]),
    "hypmi_gram_f64_ordered": (ctypes.c_int, [c_void_p, i64, i64, ctypes.c_int, c_void_p, c_void_p, i64, c_void_p,
                                              c_void_p, i64, c_void_p]),
    "hypmi_pool_rows_f64": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, c_void_p, c_void_p, i64, c_void_p,
                                           c_void_p, i64, c_void_p]),
    "hypmi_gemm_nt_f64": (ctypes.c_int, [c_void_p, i64, i64, c_void_p, i64, i64, ctypes.c_int, c_void_p,
                                         c_void_p, i64, c_void_p]),
    "hypmi_splda_posterior": (ctypes.c_int, [c_void_p, i64, i64, ctypes.c_int, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p]),
    "hypmi_ark_probe": (ctypes.c_int, [ctypes.c_char_p, i64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hypmi_ark_read_vectors": (ctypes.c_int, [c_void_p, i32, c_void_p, c_void_p, c_void_p, i64, i32, i32,
                                              c_void_p, i64, i32]),
    "hypmi_score_extract": (ctypes.c_int, [c_void_p, i64, i64, i64, c_void_p, c_void_p, c_void_p, c_void_p,
                                           i64, ctypes.c_float, ctypes.c_float, ctypes.c_int, c_void_p,
                                           c_void_p, i64, c_void_p, c_void_p]),
    "hypmi_score_hist": (ctypes.c_int, [c_void_p, i64, i64, i64, c_void_p, c_void_p, c_void_p, c_void_p,
                                        i64, ctypes.c_float, ctypes.c_float, ctypes.c_int, c_void_p,
                                        c_void_p, c_void_p]),
}

_lib = None


def load():
    """Load libhypmi.so once.  torch is imported first so that the library binds to the HIP
    runtime torch has already loaded (one runtime per process, shared device pointers)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `make -C hyperion_amd/csrc` (or __graft_entry__.build()); "
            "hyperion_amd has no CPU fallback" % LIB_PATH)
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().hypmi_last_error()
        raise HypmiError("hypmi error %d: %s" % (rc, msg.decode() if msg else "?"))
