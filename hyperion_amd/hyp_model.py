"""Base class of the host API objects: a name plus a JSON-able config, and loading of the HDF5
model files the reference writes (hyperion/hyp_model.py:17-135: a "config" JSON string plus one
float dataset per parameter under the object's name).  Reading goes through the package's own HDF5
parser; writing model files is not provided."""
import json
import zlib

import numpy as np

from .hyp_defs import float_cpu


class HypModel(object):
    def __init__(self, name=None, **kwargs):
        self.name = name
        self._is_init = False

    @property
    def is_init(self):
        return self._is_init

    def get_config(self):
        return {"class_name": self.__class__.__name__, "name": self.name}

    # ---- model files (hyperion/hyp_model.py:86-135) --------------------------------------------
    @staticmethod
    def _config_of(f):
        return json.loads(str(np.asarray(f["config"]).astype("U")))

    @classmethod
    def load_config(cls, file_path):
        """The configuration dictionary of a model file (HDF5, or a plain JSON text file)."""
        from .io.hdf5_lite import Hdf5Error, Hdf5File

        try:
            return cls._config_of(Hdf5File(file_path))
        except Hdf5Error:
            with open(file_path, "r") as fh:
                return json.loads(fh.read())

    @classmethod
    def load(cls, file_path):
        from .io.hdf5_lite import Hdf5File

        f = Hdf5File(file_path)
        return cls.load_params(f, cls._config_of(f))

    @classmethod
    def load_params(cls, f, config):
        return cls(name=config["name"])

    @staticmethod
    def _load_params_to_dict(f, name, params, dtypes=None):
        """{parameter: array or None} from the datasets `name/parameter` of an open model file."""
        prefix = "" if name is None else name + "/"
        out = {}
        for k in params:
            p_name = prefix + k
            dtype = float_cpu() if dtypes is None else dtypes[k]
            out[k] = np.asarray(f[p_name]).astype(dtype, copy=False) if p_name in f else None
        return out


def fingerprint(*arrays):
    """Cheap content key for caching device plans against host parameters that users may
    replace or edit in place."""
    key = []
    for a in arrays:
        if a is None:
            key.append(None)
        else:
            a = np.ascontiguousarray(a)
            key.append((a.shape, str(a.dtype), zlib.crc32(a.view(np.uint8).reshape(-1))))
    return tuple(key)
