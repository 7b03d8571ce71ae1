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


    # ---- bare matrix files and command-line plumbing ------------------------------------------
    @classmethod
    def load_mat(cls, file_path):
        """mu and T from the top level of an HDF5 file, at float32 precision as the reference reads
        them (e.g. hyperion/transforms/cent_whiten.py:79-84)."""
        from ..io.hdf5_lite import Hdf5File

        f = Hdf5File(file_path)
        return cls(np.asarray(f["mu"]).astype("float32"), np.asarray(f["T"]).astype("float32"))

    def save_mat(self, file_path):
        """mu and T as top-level datasets in their own precision (cent_whiten.py:86-89)."""
        from ..io.hdf5_write import File

        with File(file_path, "w") as f:
            f.create_dataset("mu", data=np.asarray(self.mu))
            f.create_dataset("T", data=np.asarray(self.T))

    _valid_args = ("update_mu", "update_T", "name")
    _default_name = "lnorm"  # what the reference's parsers say for CentWhiten and LNorm alike

    @classmethod
    def filter_args(cls, **kwargs):
        """The constructor arguments among a parsed command line (cent_whiten.py:91-94)."""
        return dict((k, kwargs[k]) for k in cls._valid_args if k in kwargs)

    @classmethod
    def add_class_args(cls, parser, prefix=None):
        """Options of the training scripts (cent_whiten.py:96-119); `type=bool` as in the
        reference, i.e. any non-empty value reads as True."""
        p1 = "--" if prefix is None else "--" + prefix + "."
        parser.add_argument(p1 + "update-mu", default=True, type=bool, help="updates centering parameter")
        parser.add_argument(p1 + "update-T", default=True, type=bool, help="updates whitening parameter")
        cls._add_own_args(parser, p1)

    @classmethod
    def _add_own_args(cls, parser, p1):
        parser.add_argument(p1 + "name", default=cls._default_name)

    @classmethod
    def add_argparse_args(cls, parser, prefix=None):
        return cls.add_class_args(parser, prefix)


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
