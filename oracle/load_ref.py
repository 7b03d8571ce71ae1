"""Loader for the upstream reference package (in-container only; never shipped to the GPU box).

TEST INFRASTRUCTURE.  Used only by oracle/gen_golden.py to validate the numpy restatement in
oracle/plda_ref.py and to emit the fixtures under tests/golden/.  Nothing in the product package
imports this file, and nothing run on the GPU box reads /root/reference.

The reference's package __init__ pulls in audio/HDF5 I/O that is not installed here, so the
package object is created by hand with its __path__ pointing at the reference tree and only the
numeric sub-packages (pdfs.plda, transforms, score_norm, metrics, utils.math) are imported.
"""
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

REF_ROOT = "/root/reference"


def load():
    for a, t in [("str", str), ("float", float), ("int", int), ("bool", bool), ("object", object)]:
        if not hasattr(np, a):
            setattr(np, a, t)  # numpy>=1.24 removed the aliases used in hyperion/utils
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))  # only save/load need it
    ja = types.ModuleType("jsonargparse")
    for n in ["ArgumentParser", "ActionParser", "ActionConfigFile", "namespace_to_dict"]:
        setattr(ja, n, object)
    sys.modules.setdefault("jsonargparse", ja)  # CLI-only dependency
    pkg = types.ModuleType("hyperion")
    pkg.__path__ = [REF_ROOT + "/hyperion"]
    sys.modules["hyperion"] = pkg
    from hyperion.pdfs.plda import SPLDA, FRPLDA, PLDA
    from hyperion.transforms import CentWhiten, LNorm, LDA, PCA, TransformList
    from hyperion.score_norm import AdaptSNorm, SNorm, TNorm, ZNorm
    from hyperion.metrics import compute_eer, compute_min_dcf
    from hyperion.utils import math as hmath

    ns = types.SimpleNamespace(
        SPLDA=SPLDA, FRPLDA=FRPLDA, PLDA=PLDA, CentWhiten=CentWhiten, LNorm=LNorm, LDA=LDA,
        PCA=PCA, TransformList=TransformList, AdaptSNorm=AdaptSNorm, SNorm=SNorm, TNorm=TNorm,
        ZNorm=ZNorm, compute_eer=compute_eer, compute_min_dcf=compute_min_dcf, math=hmath)
    return ns
