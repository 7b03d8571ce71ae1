"""The small helpers of hyperion_amd/utils/math.py against the reference's outputs
(tests/golden/math_helpers.npz, made by oracle/gen_golden.py from hyperion/utils/math.py:102-347)."""
import numpy as np
import scipy.linalg as sla

from hyperion_amd.utils import math as M

TOL = dict(rtol=1e-11, atol=1e-12)


def test_elementwise_and_reductions(golden):
    g = golden("math_helpers")
    r, xs = g["r"], g["xs"]
    assert np.allclose(M.softmax(r), g["softmax"], **TOL) and np.allclose(M.softmax(r, axis=0), g["softmax0"], **TOL)
    assert np.allclose(M.logsumexp(r), g["logsumexp"], **TOL)
    assert np.allclose(M.logsumexp(r, axis=0), g["logsumexp0"], **TOL)
    keep = xs.copy()
    for name in ("logsigmoid", "neglogsigmoid", "sigmoid"):
        assert np.allclose(getattr(M, name)(xs), g[name], **TOL), name
    assert np.array_equal(xs, keep)  # inputs are left alone
    assert M.sigmoid(np.array([-150.0]))[0] == 0.0 and M.logsigmoid(np.array([-150.0]))[0] == -150.0


def test_fisher_ratio_and_triangles(golden):
    g = golden("math_helpers")
    assert np.isclose(M.fisher_ratio(g["mu1"], g["S1"], g["mu2"], g["S2"]), g["fisher"], rtol=1e-11)
    assert np.isclose(M.fisher_ratio_with_precs(g["mu1"], g["S1"], g["mu2"], g["S2"]), g["fisher_precs"], rtol=1e-10)
    A = g["A"]
    keep = A.copy()
    for lower, t in ((False, "_u"), (True, "_l")):
        assert np.array_equal(M.symmat2vec(A, lower=lower), g["sym2vec" + t])
        assert np.allclose(M.symmat2vec(A, lower=lower, diag_factor=0.5), g["sym2vec_df" + t], **TOL)
        assert np.array_equal(M.vec2symmat(g["sym2vec" + t], lower=lower), g["vec2sym" + t])
        assert np.allclose(M.vec2symmat(g["sym2vec_df" + t], lower=lower, diag_factor=2), g["vec2sym_df" + t], **TOL)
        tri = np.tril(A) if lower else np.triu(A)
        assert np.array_equal(M.trimat2vec(tri, lower=lower), g["tri2vec" + t])
        assert np.array_equal(M.vec2trimat(g["tri2vec" + t], lower=lower), g["vec2tri" + t])
        assert np.array_equal(M.vec2trimat(M.trimat2vec(tri, lower), lower), tri)
    assert np.array_equal(A, keep)
    assert np.array_equal(M.int2onehot(g["ids"]), g["onehot"]) and np.array_equal(M.int2onehot(g["ids"], 9), g["onehot9"])


def test_full_covariance_floor(golden):
    g = golden("math_helpers")
    S, F = g["vf_S"], g["vf_F"]
    tol = dict(rtol=1e-9, atol=1e-10)
    got = M.fullcov_varfloor(S, F)
    assert np.allclose(got, g["vf"], **tol)
    # what the floor means: no direction has less variance than under F, and S is kept where it is larger
    assert np.min(sla.eigvalsh(got, F)) >= 1 - 1e-9 and np.min(sla.eigvalsh(got - S)) >= -1e-9
    assert np.allclose(M.fullcov_varfloor(S, 0.7), g["vf_scalar"], **tol)
    for lower, t in ((False, "_u"), (True, "_l")):
        cF, cS = sla.cholesky(F, lower=lower), sla.cholesky(S, lower=lower)
        assert np.allclose(M.fullcov_varfloor(S, cF, F_is_chol=True, lower=lower), g["vf_chol" + t], **tol)
        assert np.allclose(M.fullcov_varfloor_from_cholS(cS, cF, lower=lower), g["vf_from_chol" + t], **tol)
        assert np.allclose(M.fullcov_varfloor_from_cholS(cS, np.sqrt(0.7), lower=lower),
                           g["vf_from_chol_scalar" + t], **tol)
