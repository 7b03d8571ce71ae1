"""Common machinery of the PLDA models (host API of hyperion/pdfs/plda/plda_base.py, executed on
the GPU): scoring, sufficient statistics and the EM training loop."""
import numpy as np

from ... import _lib, engine
from ...engine import SIDE_ENROL, SIDE_TEST
from ...fold import fold_chain
from ...hyp_defs import float_cpu
from ...hyp_model import HypModel, fingerprint

_LNORM_STEP = (None, None, True)  # parameter-free length normalisation, LNorm()


def _group_index(class_ids):
    """Row order grouped by class and group offsets for classes 0..max(id)
    (plda_base.py:251: num_classes = max(class_ids) + 1; ids that never occur give empty groups)."""
    ids = np.asarray(class_ids)
    if ids.ndim != 1:
        raise ValueError("class ids must be a 1-D integer array")
    ids = ids.astype(np.int64)
    if ids.size and ids.min() < 0:
        raise ValueError("class ids must be non-negative")
    num_classes = int(ids.max()) + 1 if ids.size else 0
    order = np.argsort(ids, kind="stable").astype(np.int32)
    counts = np.bincount(ids, minlength=num_classes)
    offsets = np.zeros(num_classes + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    return order, offsets, counts


def pool_rows(x_dev, class_ids, mean):
    """Device class pooling.  x_dev: device fp32 [n, d].  Returns (pooled [C, d] device tensor,
    counts as a host int array)."""
    torch = engine.require_gpu()
    lib = _lib.load()
    order, offsets, counts = _group_index(class_ids)
    if order.shape[0] != x_dev.shape[0]:
        raise ValueError("%d class ids for %d rows" % (order.shape[0], x_dev.shape[0]))
    C, d = len(counts), x_dev.shape[1]
    ld = engine.round_up(d, 4)
    out = torch.zeros((C, ld), dtype=torch.float32, device=x_dev.device)
    if C:
        o_dev = torch.from_numpy(order).to(x_dev.device)
        f_dev = torch.from_numpy(offsets).to(x_dev.device)
        _lib.check(lib.hypmi_pool_rows(engine._ptr(x_dev), x_dev.stride(0), d, engine._ptr(o_dev),
                                       engine._ptr(f_dev), C, int(bool(mean)), engine._ptr(out), ld,
                                       None, engine.stream_ptr()))
    return out[:, :d], counts


class ClassStats:
    """Sufficient statistics of a labelled training set, resident on the device for the epochs of a
    training run: N [M] class counts (host float64), F [M, d] class sums (device float64), S [d, d]
    second-order statistic (device float64).  `ClassStats.of(D)` takes a host (N, F, S) tuple as it is
    passed around by the reference's API."""

    def __init__(self, N, F, S):
        self.N = np.asarray(N, dtype=np.float64)
        self.F = F
        self.S = S
        self._S_host = None

    @classmethod
    def of(cls, D):
        if isinstance(D, cls):
            return D
        N, F, S = D
        return cls(N, engine.upload64(F), None if S is None else engine.upload64(S))

    @property
    def S_host(self):
        if self._S_host is None and self.S is not None:
            self._S_host = self.S.cpu().numpy()
        return self._S_host

    def host(self):
        return (self.N.astype(float_cpu()), self.F.cpu().numpy().astype(float_cpu()),
                None if self.S is None else self.S_host.astype(float_cpu()))

    def __iter__(self):  # N, F, S = D
        return iter(self.host())


class PLDABase(HypModel):
    def __init__(self, y_dim=None, mu=None, update_mu=True, **kwargs):
        super().__init__(**kwargs)
        self.mu = mu
        self.y_dim = y_dim
        self.update_mu = update_mu
        if mu is not None:
            self.x_dim = mu.shape[0]
        self._plans = {}
        self._book_cache = (None, None)

    _param_names = ()
    _cache_defaults = {"_plans": dict, "_book_cache": lambda: (None, None)}

    def save_params(self, f):
        """The parameters named by `_param_names`, e.g. mu, V, W (hyperion/pdfs/plda/splda.py:253-255)."""
        self._save_params_from_dict(f, {k: getattr(self, k) for k in self._param_names})

    @classmethod
    def load_params(cls, f, config):
        """Parameters from an open model file; the configuration entries are constructor
        arguments (e.g. hyperion/pdfs/plda/splda.py:257-262)."""
        params = cls._load_params_to_dict(f, config["name"], list(cls._param_names))
        kwargs = dict(list(config.items()) + list(params.items()))
        kwargs.pop("class_name", None)
        return cls(**kwargs)

    # ---- to be provided by the concrete model ---------------------------------------------
    def _params(self):
        raise NotImplementedError

    def _fold(self):
        """-> (At, bt, An, bn, c) in float64, see hyperion_amd/fold.py."""
        raise NotImplementedError

    def _book_fold(self):
        """-> dict(P, row_bias, shift_t, shift_n, lam, c_extra), see llr_NvsM_book."""
        raise NotImplementedError

    def validate(self):
        raise NotImplementedError

    @engine.small_host_algebra
    def _book_fold_cached(self):
        """_book_fold() of the current parameters (an eigendecomposition; kept until they change)."""
        key = fingerprint(*self._params())
        if self._book_cache[0] != key:
            self._book_cache = (key, self._book_fold())
        return self._book_cache[1]

    @engine.small_host_algebra
    def _shared_fold(self):
        """The 1-vs-1 projection in the basis of by-the-book scoring, one segment a side: both
        gamma_tar and gamma_non are per-column maps of s = x P + row_bias there
        (|gamma(n)|^2 = sum_k (s_k + shift_k)^2 / (1 + n lam_k)), so the device needs K columns, not
        2K.  The constant must be the one of _fold()."""
        bf = self._book_fold_cached()
        lam = bf["lam"]
        c = 0.5 * (2.0 * float(np.sum(np.log1p(lam))) - float(np.sum(np.log1p(2.0 * lam)))) + bf["c_extra"]
        c_ref = float(self._fold()[4])
        if abs(c - c_ref) > 1e-8 * max(1.0, abs(c_ref)):
            raise AssertionError("shared-basis constant %r differs from %r" % (c, c_ref))
        return {"P": bf["P"], "pb": bf["row_bias"], "shift_t": bf["shift_t"], "shift_n": bf["shift_n"],
                "scale_t": 1.0 / np.sqrt(1.0 + 2.0 * lam), "scale_n": 1.0 / np.sqrt(1.0 + lam)}

    def _validate_for_scoring(self):
        self.validate()

    # ---- plans -----------------------------------------------------------------------------
    @engine.small_host_algebra
    def score_plan(self, preproc=None, lnorm_first=False):
        """Device plan for [preproc chain ->] [LNorm() ->] this model's projection.  `preproc` is
        a hyperion_amd TransformList (or a list of transforms) folded in front of the model."""
        self._validate_for_scoring()
        transforms = []
        if preproc is not None:
            transforms = list(getattr(preproc, "transforms", preproc))
        key = (fingerprint(*self._params()), tuple(t._params_key() for t in transforms),
               tuple(type(t).__name__ for t in transforms), bool(lnorm_first))
        plan = self._plans.get(key)
        if plan is None:
            steps = [t._step() for t in transforms]
            if lnorm_first:
                steps.append(_LNORM_STEP)
            d_in = self.x_dim
            for A, b, _ in steps:  # input dim of the chain = first matrix's rows, else model dim
                if A is not None:
                    d_in = A.shape[0]
                    break
                if b is not None:
                    d_in = b.shape[0]
                    break
            segments, tail, d_out = fold_chain(steps, d_in)
            if d_out != self.x_dim:
                raise ValueError("pre-processing gives %d-dim vectors, model expects %d"
                                 % (d_out, self.x_dim))
            if len(self._plans) > 8:
                self._plans.clear()
            plan = engine.Plan(d_in, segments, tail, self._fold(), shared=self._shared_fold())
            self._plans[key] = plan
        return plan

    # ---- 1 vs 1 -----------------------------------------------------------------------------
    def llr_1vs1(self, x1, x2, preproc=None, refine=None):
        """Log-likelihood ratio of every (x1[i], x2[j]) pair: [N1, N2] `float_cpu()` array.
        numpy in / numpy out; device tensors in give a device tensor out.

        refine: scores whose fp32 error bound is not below half of 1e-4 |s| (scores that cancel to
        about zero, ~1 % of a calibrated back-end's trials) are re-evaluated in float64 on the
        device, so that every score matches the float64 reference to 1e-4 relative.  Default: on for
        numpy inputs, off for device tensors."""
        torch = engine.require_gpu()
        plan = self.score_plan(preproc)
        dev_out = isinstance(x1, torch.Tensor) and isinstance(x2, torch.Tensor)
        if refine is None:
            refine = not dev_out
        G1 = plan.prologue(engine.upload(x1), SIDE_ENROL)
        G2 = plan.prologue(engine.upload(x2), SIDE_TEST)
        S = engine.score(G1, G2)
        if refine and S.numel():
            D1, b1 = plan.prologue_f64(engine.upload64(x1), SIDE_ENROL)
            D2, b2 = plan.prologue_f64(engine.upload64(x2), SIDE_TEST)
            engine.refine_scores(S, D1, b1, D2, b2, rel_tol=0.5e-4)
        return S if dev_out else engine.to_host(S, float_cpu())

    # ---- sufficient statistics ----------------------------------------------------------------
    @staticmethod
    def _stats_hard_device(x, class_ids, sample_weight=None, scale_factor=None):
        """compute_stats_hard with everything left on the device: a ClassStats."""
        X = engine.upload64(x)
        order, offsets, counts = _group_index(class_ids)
        if order.shape[0] != X.shape[0]:
            raise ValueError("%d class ids for %d rows" % (order.shape[0], X.shape[0]))
        if sample_weight is not None:
            ids = np.asarray(class_ids).astype(np.int64).reshape(-1)
            N = np.bincount(ids, weights=np.asarray(sample_weight, dtype=np.float64),
                            minlength=len(counts)).astype(np.float64)
        else:
            N = counts.astype(np.float64)
        F = engine.pool64(X, order, offsets, sample_weight)
        S = engine.gram64(X, sample_weight)[0]
        if scale_factor is not None:
            N, F, S = N * scale_factor, F * scale_factor, S * scale_factor
        return ClassStats(N, F, S)

    @staticmethod
    def compute_stats_hard(x, class_ids, sample_weight=None, scale_factor=None):
        """(N, F, S) per class for classes 0..max(id) (plda_base.py:249-274): counts, per-class
        sums (float64 segmented sums on the GPU) and the global second-order statistic S = x^T x
        (float64 matrix cores).  With sample_weight w: N = sum w, F = sum w x, S = (w x)^T x."""
        return PLDABase._stats_hard_device(x, class_ids, sample_weight, scale_factor).host()

    @staticmethod
    def compute_stats_hard_v0(x, class_ids, sample_weight=None, scal_factor=None):
        """The reference's earlier statement of compute_stats_hard through a one-hot posterior
        matrix (plda_base.py:276-282); the same statistics, so the same device path without the
        [n, classes] matrix."""
        return PLDABase.compute_stats_hard(x, class_ids, sample_weight, scal_factor)

    @staticmethod
    def center_stats(D, mu):
        """Statistics of the vectors shifted by -mu, from the statistics themselves
        (plda_base.py:284-290): Fc = F - N mu^T, Sc = S - (sum F) mu^T - mu (sum F)^T + (sum N) mu mu^T.
        Host algebra on the [classes, d] and [d, d] statistics; nothing is modified in place."""
        N, F, S = D
        N, F, S, mu = (np.asarray(a, dtype=np.float64) for a in (N, F, S, mu))
        Fmu = np.outer(np.sum(F, axis=0), mu)
        return N, F - np.outer(N, mu), S - Fmu - Fmu.T + np.sum(N) * np.outer(mu, mu)

    @staticmethod
    def compute_stats_soft(x, p_theta, sample_weight=None, scal_factor=None):
        """Statistics from class posteriors p_theta [n, C] (plda_base.py:236-246): N = sum_s p,
        F = p^T x, S = x^T diag(sum_c p) x.  (The reference scales the caller's p_theta in place
        when scal_factor is given; here it is left alone.)"""
        return PLDABase._stats_soft_device(x, p_theta, sample_weight, scal_factor).host()

    @staticmethod
    def _stats_soft_device(x, p_theta, sample_weight=None, scal_factor=None):
        X = engine.upload64(x)
        P = engine.upload64(p_theta)
        if sample_weight is not None:
            P = P * engine.upload64(np.asarray(sample_weight, dtype=np.float64)[:, None])
        if scal_factor is not None:
            P = P * scal_factor
        N = P.sum(dim=0).cpu().numpy()
        F = engine.gemm64(P.t().contiguous(), X.t().contiguous())  # [C, n] [d, n]^T
        S = engine.gram64(X, P.sum(dim=1))[0]
        return ClassStats(N, F, S)

    # ---- training ------------------------------------------------------------------------------
    def initialize(self, D):
        raise NotImplementedError

    def Estep(self, D):
        raise NotImplementedError

    def elbo(self, stats):
        raise NotImplementedError

    def MstepML(self, stats):
        raise NotImplementedError

    def MstepMD(self, stats):
        raise NotImplementedError

    def _train_stats(self, x, class_ids, ptheta, sample_weight):
        assert not (class_ids is None and ptheta is None)
        if class_ids is None:
            return self._stats_soft_device(x, ptheta)
        return self._stats_hard_device(x, class_ids)

    @engine.small_host_algebra
    def fit(self, x, class_ids=None, ptheta=None, sample_weight=None, x_val=None, class_ids_val=None,
            ptheta_val=None, sample_weight_val=None, epochs=20, ml_md="ml+md", md_epochs=None):
        """EM training (plda_base.py:34-88): statistics once, then per epoch an E-step on the device
        over the class statistics, the ELBO, and the maximum-likelihood and / or minimum-divergence
        M-steps on the small accumulators.  Returns (elbo, elbo / N) per epoch, and the same for the
        validation set when one is given.  As in the reference, sample weights are accepted and not
        used (plda_base.py:56)."""
        D = self._train_stats(x, class_ids, ptheta, sample_weight)
        D_val = None
        if x_val is not None:
            D_val = self._train_stats(x_val, class_ids_val, ptheta_val, sample_weight_val)
        if not self.is_init:
            self.initialize(D)
        return self._em_epochs(D, D_val, epochs, ml_md, md_epochs)

    def _posterior(self, D, want_logpy):
        """Device posterior means of the speaker factors of SPLDA and PLDA (`_WV()` gives W V^T and
        V W V^T, through the Woodbury form for the full model).  V W V^T = Q diag(lam) Q^T makes
        every L_i = I + N_i V W V^T diagonal in the basis Q, so the reference's loop over distinct
        counts with a Cholesky each (splda.py:112-142, plda.py:157-186) is two products and one
        scaling pass.  Returns (Z, y, logpy, logpy_sum, lam, Q): Z = [F | y] as one device buffer."""
        from scipy import linalg as sla

        torch = engine.require_gpu()
        mu = np.asarray(self.mu, dtype=np.float64)
        WV, VV = self._WV()
        lam, Q = sla.eigh(0.5 * (VV + VV.T))
        lam = np.maximum(lam, 0.0)
        B = WV @ Q  # [d, K]
        M, d = D.F.shape
        K = VV.shape[0]
        Z = torch.empty((M, d + K), dtype=torch.float64, device="cuda")
        Z[:, :d] = D.F
        T = engine.gemm64(D.F, np.ascontiguousarray(B.T), bias=-(mu @ B))  # (F - mu) W V^T Q
        logpy, total = engine.splda_posterior(T, D.N, lam)
        y = engine.gemm64(T, Q, out=Z[:, d:])  # back to the model's basis
        return Z, y, (logpy if want_logpy else None), total, lam, Q

    @staticmethod
    def _posterior_cov_sums(N, lam, Q):
        """sum_i L_i^-1 and sum_i N_i L_i^-1 (Py and Ry of splda.py:144-146) from the distinct counts."""
        uniq, cnt = np.unique(N, return_counts=True)
        inv = 1.0 / (1.0 + uniq[:, None] * lam[None, :])
        Py = (Q * (cnt[:, None] * inv).sum(axis=0)) @ Q.T
        Ry = (Q * ((cnt * uniq)[:, None] * inv).sum(axis=0)) @ Q.T
        return Ry, Py

    @engine.small_host_algebra
    def compute_py_g_x(self, D, return_cov=False, return_logpy_0=False, return_acc=False):
        """Posterior of the speaker factor of every class given its statistics (splda.py:73-158, plda.py:119-198):
        y [M, y_dim] and on request the posterior covariances [M, y_dim, y_dim], the log
        normalisers logpy [M], and the accumulators (Ry, Py).  Host arrays out."""
        D = ClassStats.of(D)
        _, y, logpy, _, lam, Q = self._posterior(D, return_logpy_0)
        r = [y.cpu().numpy().astype(float_cpu())]
        if return_cov:
            uniq, inverse = np.unique(D.N, return_inverse=True)
            cov = np.einsum("ak,uk,bk->uab", Q, 1.0 / (1.0 + uniq[:, None] * lam[None, :]), Q)
            r.append(cov[inverse].astype(float_cpu()))
        if return_logpy_0:
            r.append(logpy.cpu().numpy().astype(float_cpu()))
        if return_acc:
            r += list(self._posterior_cov_sums(D.N, lam, Q))
        return r[0] if len(r) == 1 else tuple(r)

    def _em_epochs(self, D, D_val, epochs, ml_md, md_epochs, after_epoch=None, also=None):
        """The epoch loop shared by fit and the adaptation variants."""
        use_ml = ml_md != "md"
        use_md = ml_md != "ml"
        elbo = np.zeros((epochs,), dtype=float_cpu())
        elbo_val = np.zeros((epochs,), dtype=float_cpu())
        for epoch in range(epochs):
            stats = self.Estep(D)
            if also is not None:
                self.Estep(also)
            elbo[epoch] = self.elbo(stats)
            if D_val is not None:
                elbo_val[epoch] = self.elbo(self.Estep(D_val))
            if use_ml:
                self.MstepML(stats)
            if use_md and (md_epochs is None or epoch in md_epochs):
                self.MstepMD(stats)
            if after_epoch is not None:
                after_epoch()
        elbo_norm = elbo / np.sum(D.N)
        if D_val is None:
            return elbo, elbo_norm
        return elbo, elbo_norm, elbo_val, elbo_val / np.sum(D_val.N)

    @engine.small_host_algebra
    def fit_adapt_weighted_avg_model(self, x, class_ids=None, ptheta=None, sample_weight=None, x_val=None,
                                     class_ids_val=None, ptheta_val=None, sample_weight_val=None, epochs=20,
                                     ml_md="ml+md", md_epochs=None, plda0=None, w_mu=1, w_B=0.5, w_W=0.5):
        """EM on adaptation data from the current model, pulled back towards `plda0` after every
        epoch by weighted_avg_model (plda_base.py:111-167)."""
        assert self.is_init
        D = self._train_stats(x, class_ids, ptheta, sample_weight)
        D_val = None if x_val is None else self._train_stats(x_val, class_ids_val, ptheta_val, sample_weight_val)
        return self._em_epochs(D, D_val, epochs, ml_md, md_epochs,
                               after_epoch=lambda: self.weighted_avg_model(plda0, w_mu, w_B, w_W))

    @engine.small_host_algebra
    def fit_adapt(self, x, class_ids=None, ptheta=None, sample_weight=None, x0=None, class_ids0=None, ptheta0=None,
                  sample_weight0=None, x_val=None, class_ids_val=None, ptheta_val=None, sample_weight_val=None,
                  epochs=20, ml_md="ml+md", md_epochs=None):
        """EM on adaptation data from the current model (plda_base.py:169-235).  As there, the
        statistics of the out-of-domain set (x0) go through the E-step every epoch and are not used
        further, and the call needs them: without x0 the reference stops with a NameError, this
        with an AssertionError."""
        assert self.is_init
        assert x0 is not None, "fit_adapt needs the out-of-domain set x0"
        D = self._train_stats(x, class_ids, ptheta, sample_weight)
        D0 = self._train_stats(x0, class_ids0, ptheta0, sample_weight0)
        D_val = None if x_val is None else self._train_stats(x_val, class_ids_val, ptheta_val, sample_weight_val)
        return self._em_epochs(D, D_val, epochs, ml_md, md_epochs, also=D0)

    # declared abstract by the reference (plda_base.py:353-355, 365-367); every model overrides
    # `sample`, and none overrides the misspelt `weigthed_avg_model` (they define weighted_avg_model)
    def sample(self, num_classes, num_samples_per_class, rng=None, seed=1024):
        raise NotImplementedError("%s has no sample()" % type(self).__name__)

    def weigthed_avg_model(self, plda):
        raise NotImplementedError("use weighted_avg_model(plda, w_mu, w_B, w_W)")

    def _row_logp(self, delta, Wq, logdet_W):
        """Row-wise Gaussian log-density (-d log 2 pi + log|W| - delta W delta^T) / 2 on the device in
        float64; delta a device [n, d] tensor, Wq the precision as [d, d] or a diagonal [d]."""
        torch = engine.require_gpu()
        if np.ndim(Wq) == 1:
            Wd = torch.from_numpy(np.ascontiguousarray(Wq, dtype=np.float64)).to(delta.device)
            quad = (delta * delta * Wd).sum(dim=1)
        else:
            quad = (engine.gemm64(delta, np.ascontiguousarray(np.asarray(Wq, dtype=np.float64).T)) * delta).sum(dim=1)
        logp = 0.5 * (-delta.shape[1] * np.log(2 * np.pi) + logdet_W - quad)
        return logp

    def _residual(self, x, *terms):
        """x - mu - sum_i y_i M_i on the device in float64 (terms = (y_i, M_i) pairs, M_i [K_i, d])."""
        torch = engine.require_gpu()
        Y = np.concatenate([np.asarray(y, dtype=np.float64) for y, _ in terms], axis=1)
        M = np.concatenate([np.asarray(m, dtype=np.float64) for _, m in terms], axis=0)  # [sum K, d]
        delta = engine.gemm64(Y, np.ascontiguousarray(-M.T), bias=-np.asarray(self.mu, dtype=np.float64))
        X = x if isinstance(x, torch.Tensor) else engine.upload64(np.asarray(x, dtype=np.float64))
        return delta.add_(X.to(torch.float64))

    def weigthed_avg_params(self, mu, w_mu):
        """The mean part of the model interpolation, under the reference's spelling
        (plda_base.py:362-363)."""
        self.mu = w_mu * np.asarray(mu, dtype=np.float64) + (1 - w_mu) * np.asarray(self.mu, dtype=np.float64)

    # ---- N vs M -----------------------------------------------------------------------------
    def _pooled_mean(self, x, ids):
        """Device class means F / N; `x` may already be a statistics tuple (ids is None)."""
        torch = engine.require_gpu()
        if ids is None:
            lib = _lib.load()
            F = engine.upload(x[1]).clone()
            N = torch.from_numpy(np.ascontiguousarray(x[0], dtype=np.float32)).to(F.device)
            _lib.check(lib.hypmi_div_rows(engine._ptr(F), F.stride(0), F.shape[0], F.shape[1],
                                          engine._ptr(N), engine.stream_ptr()))
            return F
        xm, _ = pool_rows(engine.upload(x), ids, mean=True)
        return xm

    @staticmethod
    def _result(S, *inputs):
        """Device tensors in give the device matrix back; anything else a `float_cpu()` array."""
        torch = engine.require_gpu()
        if all(isinstance(x, torch.Tensor) for x in inputs):
            return S
        return engine.to_host(S, float_cpu())

    @staticmethod
    def _book_result(S, *sides):
        """The caller gets a device matrix only for device inputs (vectors, or the F of a statistics
        tuple when ids is None); the statistics built here from host vectors do not count."""
        torch = engine.require_gpu()
        if not isinstance(S, torch.Tensor):
            return S
        for x, ids in sides:
            given = x[1] if ids is None else x
            if not isinstance(given, torch.Tensor):
                return engine.to_host(S, float_cpu())
        return S

    def llr_NvsM(self, x1, x2, ids1=None, ids2=None, method="vavg-lnorm"):
        """Multi-session scoring (plda_base.py:292-322).  numpy in / numpy out; device tensors in
        (methods vavg, vavg-lnorm, savg) give a device tensor out."""
        if method == "savg":
            return self._llr_savg(x1, ids1, x2, ids2)
        if method == "book":
            D1 = x1 if ids1 is None else self._stats_dev(x1, ids1)
            D2 = x2 if ids2 is None else self._stats_dev(x2, ids2)
            return self._book_result(self.llr_NvsM_book(D1, D2), (x1, ids1), (x2, ids2))
        if method not in ("vavg", "vavg-lnorm"):
            raise ValueError("unknown pooling method %r" % (method,))
        plan = self.score_plan(lnorm_first=(method == "vavg-lnorm"))
        G1 = plan.prologue(self._pooled_mean(x1, ids1), SIDE_ENROL)
        G2 = plan.prologue(self._pooled_mean(x2, ids2), SIDE_TEST)
        return self._result(engine.score(G1, G2), x1, x2)

    def llr_Nvs1(self, x1, x2, ids1=None, method="vavg-lnorm"):
        if method == "savg":
            return self._llr_savg(x1, ids1, x2, None)
        if method == "book":
            D1 = x1 if ids1 is None else self._stats_dev(x1, ids1)
            D2 = self._stats_dev(x2, np.arange(x2.shape[0]))
            return self._book_result(self.llr_NvsM_book(D1, D2), (x1, ids1), (x2, 0))
        if method not in ("vavg", "vavg-lnorm"):
            raise ValueError("unknown pooling method %r" % (method,))
        plan = self.score_plan(lnorm_first=(method == "vavg-lnorm"))
        G1 = plan.prologue(self._pooled_mean(x1, ids1), SIDE_ENROL)
        G2 = plan.prologue(engine.upload(x2), SIDE_TEST)
        return self._result(engine.score(G1, G2), x1, x2)

    # the pooling variants under the names the reference gives them (plda_base.py:306-351)
    def llr_NvsM_vavg(self, D1, D2, do_lnorm=True):
        """Scores of the class means F / N of two statistics tuples, length-normalised first when
        do_lnorm (plda_base.py:306-314)."""
        return self.llr_NvsM(D1, D2, method="vavg-lnorm" if do_lnorm else "vavg")

    def llr_NvsM_savg(self, x1, ids1, x2, ids2):
        """Segment-level scores averaged over the classes of both sides (plda_base.py:316-322)."""
        return self._llr_savg(x1, ids1, x2, ids2)

    def llr_Nvs1_vavg(self, D1, x2, do_lnorm=True):
        """plda_base.py:338-345."""
        return self.llr_Nvs1(D1, x2, method="vavg-lnorm" if do_lnorm else "vavg")

    def llr_Nvs1_savg(self, x1, ids1, x2):
        """plda_base.py:347-351."""
        return self._llr_savg(x1, ids1, x2, None)

    def _llr_savg(self, x1, ids1, x2, ids2):
        """Score averaging (plda_base.py:316-322, 347-351).  The score is bilinear in the packed
        rows, so averaging the segment-level scores over a class equals scoring the class mean of
        the packed rows; the [segments x segments] matrix is never formed."""
        plan = self.score_plan()
        G1 = plan.prologue(engine.upload(x1), SIDE_ENROL)
        G2 = plan.prologue(engine.upload(x2), SIDE_TEST)
        if ids1 is not None:
            G1 = pool_rows(G1, ids1, mean=True)[0].contiguous()
        if ids2 is not None:
            G2 = pool_rows(G2, ids2, mean=True)[0].contiguous()
        return self._result(engine.score(G1, G2), x1, x2)

    def _stats_dev(self, x, ids):
        """(N, F, None) of by-the-book scoring with F on the device: float64 class sums for host
        vectors (the reference pools and centres in float64), fp32 ones for device tensors."""
        torch = engine.require_gpu()
        if not isinstance(x, torch.Tensor):
            order, offsets, counts = _group_index(ids)
            X = engine.upload64(x)
            if order.shape[0] != X.shape[0]:
                raise ValueError("%d class ids for %d rows" % (order.shape[0], X.shape[0]))
            return counts.astype(np.float64), engine.pool64(X, order, offsets), None
        F_dev, counts = pool_rows(engine.upload(x), ids, mean=False)
        return counts.astype(np.float64), F_dev, None

    def llr_NvsM_book(self, D1, D2, refine=None):
        """By-the-book multi-session scoring (splda.py:318-379, frplda.py:305-366,
        plda.py:474-539) from statistics D = (N, F, S); F as device tensors on both sides gives a
        device matrix back.

        refine: as in llr_1vs1 -- every block's scores whose fp32 error bound is not below half of
        1e-4 |s| are re-evaluated in float64 on the device from float64 packed rows (classes of many
        segments carry |q| in the thousands next to scores of order ten: all of their scores).
        Default: on when the statistics of both sides are float64 (host arrays, or device float64
        sums), off for fp32 device statistics, whose own rounding is then the limit.

        Every count n needs L(n) = I + n VV (or B + n W); instead of one Cholesky per count the
        first-order statistics are projected once onto the eigen-basis that diagonalises all L(n)
        and each (n1, n2) block only rescales that projection.  Unlike the reference, the
        caller's F is not modified (the reference subtracts N mu from it in place)."""
        torch = engine.require_gpu()
        lib = _lib.load()
        bf = self._book_fold_cached()
        K = bf["P"].shape[1]
        kp = lib.hypmi_packed_width(K)
        # the result first: a cached block of that size, if the allocator holds one, is then not cut
        # up for the small temporaries below
        n1, n2 = len(D1[0]), len(D2[0])
        ldo = engine.round_up(max(n2, 1), 32)
        one_block = len(np.unique(D1[0])) == 1 and len(np.unique(D2[0])) == 1
        out = (torch.empty if one_block else torch.zeros)((n1, ldo), dtype=torch.float32, device="cuda")
        sides = []
        for D in (D1, D2):
            N = np.asarray(D[0], dtype=np.float64)
            F = D[1]
            lds_ = engine.round_up(K, 4)
            if not isinstance(F, torch.Tensor) or F.dtype == torch.float64:
                # Host statistics arrive in float64 (the reference centres them in float64,
                # splda.py:326-327): project them in float64 so that F - N mu does not cancel in fp32,
                # and round only the projected statistics.
                F64 = engine.upload64(F)
                n, d = F64.shape
                s = torch.zeros((max(n, 1), K), dtype=torch.float64, device="cuda")
                if n:
                    aug64 = torch.cat([F64, torch.from_numpy(N).to(F64.device)[:, None]], dim=1)
                    W64 = np.concatenate([bf["P"].T, np.asarray(bf["row_bias"], dtype=np.float64)[:, None]], axis=1)
                    engine.gemm64(aug64, engine.upload64(W64), out=s[:n])
                sides.append((N, s))
                continue
            F_dev = engine.upload(F)
            n, d = F_dev.shape
            # s = [F | N] [P ; row_bias]: one affine map with the class size as an extra input
            ld = engine.round_up(d + 1, 4)
            aug = torch.zeros((n, ld), dtype=torch.float32, device=F_dev.device)
            aug[:, :d] = F_dev
            aug[:, d] = torch.from_numpy(N.astype(np.float32)).to(F_dev.device)
            Wt = np.zeros((K, ld), dtype=np.float32)
            Wt[:, :d] = bf["P"].T
            Wt[:, d] = bf["row_bias"]
            Wt_dev = torch.from_numpy(Wt).to(F_dev.device)
            s = torch.zeros((n, lds_), dtype=torch.float32, device=F_dev.device)
            if n:
                _lib.check(lib.hypmi_affine(engine._ptr(aug), n, ld, d + 1, engine._ptr(Wt_dev), ld,
                                            None, K, engine._ptr(s), lds_, engine.stream_ptr()))
            sides.append((N, s))
        (N1, s1), (N2, s2) = sides
        lam = bf["lam"]
        both64 = s1.dtype == torch.float64 and s2.dtype == torch.float64
        if refine is None:
            refine = both64
        elif refine and not both64:
            raise ValueError("float64 refinement of by-the-book scores needs float64 statistics on both sides")

        def vec(a, like=None):  # a map in the precision of the statistics it is applied to
            dt = np.float64 if like is not None and like.dtype == torch.float64 else np.float32
            return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()

        u1, u2 = np.unique(N1), np.unique(N2)
        single = len(u1) == 1 and len(u2) == 1
        for a in u1:
            i = np.where(N1 == a)[0].astype(np.int32)
            for b in u2:
                j = np.where(N2 == b)[0].astype(np.int32)
                logL = lambda n: float(np.sum(np.log1p(n * lam)))  # noqa: E731
                c = 0.5 * (logL(a) + logL(b) - logL(a + b)) + bf["c_extra"]
                Gs = []
                for (own, rows, s, side) in ((a, i, s1, SIDE_ENROL), (b, j, s2, SIDE_TEST)):
                    sc_t = vec(1.0 / np.sqrt(1.0 + (a + b) * lam), s)
                    sc_n = vec(1.0 / np.sqrt(1.0 + own * lam), s)
                    sh_t, sh_n = vec(bf["shift_t"], s), vec(bf["shift_n"], s)  # alive until the launch
                    G = torch.empty((len(rows), kp), dtype=torch.float32, device="cuda")
                    r_dev = torch.from_numpy(rows).cuda()
                    G64 = bnd = None
                    if refine:
                        G64 = torch.empty((len(rows), K + 2), dtype=torch.float64, device="cuda")
                        bnd = torch.empty((len(rows), 2), dtype=torch.float32, device="cuda")
                        _lib.check(lib.hypmi_book_pack_f64_rows(
                            engine._ptr(s), s.stride(0), K, engine._ptr(r_dev), len(rows), engine._ptr(sh_t),
                            engine._ptr(sc_t), engine._ptr(sh_n), engine._ptr(sc_n), side, c, engine._ptr(G), kp,
                            engine._ptr(G64), K + 2, engine._ptr(bnd), engine.stream_ptr()))
                    else:
                        pack = lib.hypmi_book_pack_f64 if s.dtype == torch.float64 else lib.hypmi_book_pack
                        _lib.check(pack(engine._ptr(s), s.stride(0), K, engine._ptr(r_dev), len(rows),
                                        engine._ptr(sh_t), engine._ptr(sc_t), engine._ptr(sh_n), engine._ptr(sc_n),
                                        side, c, engine._ptr(G), kp, engine.stream_ptr()))
                    Gs.append((G, r_dev, G64, bnd))
                if single:
                    blk = engine.score(Gs[0][0], Gs[1][0], out=out[:, :n2])
                else:
                    blk = engine.score(Gs[0][0], Gs[1][0])
                if refine and blk.numel():
                    engine.refine_scores(blk, Gs[0][2], Gs[0][3], Gs[1][2], Gs[1][3], rel_tol=0.5e-4)
                if not single:
                    _lib.check(lib.hypmi_scatter_block(engine._ptr(blk), blk.stride(0),
                                                       engine._ptr(Gs[0][1]), len(i),
                                                       engine._ptr(Gs[1][1]), len(j), engine._ptr(out),
                                                       ldo, engine.stream_ptr()))
        if all(isinstance(D[1], torch.Tensor) for D in (D1, D2)):
            return out[:, :n2]  # device statistics in, device scores out
        return engine.to_host(out[:, :n2], float_cpu())

    def get_config(self):
        config = {"y_dim": self.y_dim, "update_mu": self.update_mu}
        base_config = super().get_config()
        return dict(list(base_config.items()) + list(config.items()))
