"""Seeded synthetic models and x-vectors for tests and benchmarks (no reference code involved).

The SPLDA model is built so that scores look like a real back-end's: targets around +8, non-targets
around -14, EER near 1.6 %.  A model with random V and W gives all-positive scores and EER 0 and
is useless for parity testing near zero.
"""
import numpy as np
import scipy.linalg as sla


def make_splda(d, K, seed=1234):
    """mu = 0, V = (U * sqrt(b)).T with U orthonormal d x K and b a geometric ladder
    0.8 .. 0.02 of between-class variances, W = inv(I - V.T V) (within-class precision)."""
    rng = np.random.RandomState(seed)
    U, _ = np.linalg.qr(rng.randn(d, K))
    b = np.geomspace(0.8, 0.02, K)
    V = (U * np.sqrt(b)).T
    Sw = np.eye(d) - V.T @ V
    W = np.linalg.inv(Sw)
    W = 0.5 * (W + W.T)
    return {"type": "splda", "mu": np.zeros(d), "V": V, "W": W}


def make_frplda(d, K, seed=1234):
    """Full-rank PLDA that shares W with make_splda: B = inv(V.T V + 0.05 I), mu = 0.01."""
    m = make_splda(d, K, seed)
    B = np.linalg.inv(m["V"].T @ m["V"] + 0.05 * np.eye(d))
    B = 0.5 * (B + B.T)
    return {"type": "frplda", "mu": np.full(d, 0.01), "B": B, "W": m["W"]}


def make_plda(d, K, z_dim=50, seed=1234):
    """Full PLDA: V from make_splda, channel loading U = 0.1 randn, diagonal precision D."""
    m = make_splda(d, K, seed)
    rng = np.random.RandomState(seed + 1)
    U = 0.1 * rng.randn(z_dim, d)
    D = 1.0 / (0.5 + rng.rand(d))
    return {"type": "plda", "mu": np.zeros(d), "V": m["V"], "U": U, "D": D}


def sample_splda(model, num_classes, per_class, seed):
    """Draw per_class vectors for each of num_classes speakers from an SPLDA model:
    x = mu + y V + z chol(inv(W)), with the within-class noise drawn before the speaker factors
    (the draw order matters for reproducing a given seed)."""
    rng = np.random.RandomState(seed)
    d = model["mu"].shape[0]
    Sw = np.linalg.inv(model["W"])
    chol_Sw = sla.cholesky(0.5 * (Sw + Sw.T), lower=False)
    z = rng.normal(size=(num_classes * per_class, d)) @ chol_Sw
    y = rng.normal(size=(num_classes, model["V"].shape[0])) @ model["V"] + model["mu"]
    return np.repeat(y, per_class, axis=0) + z


def length_norm(x):
    """sqrt(D) x / (|x| + 1e-10) per row."""
    mx = np.sqrt(np.sum(x ** 2, axis=1, keepdims=True)) + 1e-10
    return np.sqrt(x.shape[1]) * x / mx


def voxceleb_like(d=512, K=150, num_classes=40, per_class=122, n=4874, seed=7):
    """Model + length-normalised vectors at VoxCeleb1-O size; returns (model, x, class_of_row)."""
    model = make_splda(d, K)
    x = length_norm(sample_splda(model, num_classes, per_class, seed))[:n]
    cls = np.repeat(np.arange(num_classes), per_class)[:n]
    return model, x, cls


def chain_cfg2(d=512, lda_dim=200, seed=3):
    """CentWhiten(512x512) -> LNorm() -> LDA(512 -> lda_dim) as a list of (kind, mu, T)."""
    rng = np.random.RandomState(seed)
    mu_cw = 0.1 * rng.randn(d)
    T_cw = rng.randn(d, d) / np.sqrt(d)
    mu_lda = 0.1 * rng.randn(d)
    T_lda = rng.randn(d, lda_dim) / np.sqrt(d)
    return [("cw", mu_cw, T_cw), ("lnorm", None, None), ("lda", mu_lda, T_lda)]


def raw_xvectors(n, d=512, seed=21, num_centers=10000, center_seed=20):
    """Raw (pre-chain) x-vectors: 0.5 * class centre + unit noise; class = row % num_centers."""
    C = np.random.RandomState(center_seed).randn(num_centers, d)
    cls = np.arange(n) % num_centers
    x = 0.5 * C[cls] + np.random.RandomState(seed).randn(n, d)
    return x, cls
