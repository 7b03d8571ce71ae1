"""Shared machinery of the affine transforms: each one reduces to x -> (x - mu) T, optionally
followed by a length normalisation; `predict` folds that into a device plan and runs it."""
import numpy as np

from .. import engine
from ..fold import affine_from_mu_T, fold_chain
from ..hyp_defs import float_cpu
from ..hyp_model import HypModel, fingerprint


class AffineTransform(HypModel):
    _lnorm = False
    _cache_defaults = {"_plan": lambda: None, "_plan_key": lambda: None}

    def __init__(self, mu=None, T=None, **kwargs):
        super().__init__(**kwargs)
        self.mu = mu
        self.T = T
        self._plan = None
        self._plan_key = None

    def _step(self):
        """(A, b, lnorm) of this transform."""
        A, b = affine_from_mu_T(self.mu, self.T)
        return A, b, self._lnorm

    def _params_key(self):
        return fingerprint(self.mu, self.T)

    def predict(self, x):
        return run_chain([self], x)

    def save_params(self, f):
        """mu and T under the transform's name (e.g. hyperion/transforms/cent_whiten.py:69-71)."""
        self._save_params_from_dict(f, {"mu": self.mu, "T": self.T})

    @classmethod
    def load_params(cls, f, config):
        """mu and T from an open model file (e.g. hyperion/transforms/cent_whiten.py:73-77)."""
        params = cls._load_params_to_dict(f, config["name"], ["mu", "T"])
        return cls(mu=params["mu"], T=params["T"], name=config["name"])


def chain_plan(transforms, d_in, cache_owner):
    key = (d_in,) + tuple(t._params_key() for t in transforms)
    if cache_owner._plan is None or cache_owner._plan_key != key:
        segments, tail, _ = fold_chain([t._step() for t in transforms], d_in)
        cache_owner._plan = engine.Plan(d_in, segments, tail, None)
        cache_owner._plan_key = key
    return cache_owner._plan


def run_chain(transforms, x, cache_owner=None):
    """numpy (or device tensor) in, same kind out."""
    owner = cache_owner if cache_owner is not None else transforms[0]
    torch = engine.require_gpu()
    is_dev = isinstance(x, torch.Tensor)
    if not is_dev:
        x = np.asarray(x)
        if x.ndim != 2:
            raise ValueError("expected [num_vectors, dim] input, got shape %s" % (x.shape,))
    plan = chain_plan(transforms, x.shape[1], owner)
    y = plan.transform(engine.upload(x))
    return y if is_dev else engine.to_host(y, float_cpu())
