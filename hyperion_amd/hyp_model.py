"""Base class of the host API objects: a name plus a JSON-able config, and loading of the HDF5
model files the reference writes (hyperion/hyp_model.py:17-135: a "config" JSON string plus one
float dataset per parameter under the object's name).  Reading and writing go through the package's
own HDF5 parser and writer (hyperion_amd/io/hdf5_lite.py, hdf5_write.py); h5py is not needed."""
import json
import os
import zlib
from copy import deepcopy

import numpy as np

from .hyp_defs import float_cpu, float_save


class HypModel(object):
    def __init__(self, name=None, **kwargs):
        self.name = name
        self._is_init = False

    # attributes that hold device plans or other values derived from the parameters -> factory of
    # their empty value; they are rebuilt on first use and never copied or pickled
    _cache_defaults = {}

    def __getstate__(self):
        state = dict(self.__dict__)
        for k, make in self._cache_defaults.items():
            if k in state:
                state[k] = make()
        return state

    def copy(self):
        """An independent model with the same parameters (hyperion/hyp_model.py:23-24)."""
        return deepcopy(self)

    @property
    def is_init(self):
        return self._is_init

    def init_to_false(self):
        self._is_init = False

    # the reference declares these three abstract (hyperion/hyp_model.py:33-43); models that train
    # override them
    def initialize(self):
        raise NotImplementedError("%s has no initialize()" % type(self).__name__)

    def fit(self, x, sample_weights=None, x_val=None, sample_weights_val=None):
        raise NotImplementedError("%s has no fit()" % type(self).__name__)

    def fit_generator(self, x, x_val=None):
        raise NotImplementedError("%s has no fit_generator()" % type(self).__name__)

    def get_config(self):
        return {"class_name": self.__class__.__name__, "name": self.name}

    @staticmethod
    def load_config_from_json(json_str):
        """hyperion/hyp_model.py:133-135."""
        return json.loads(json_str)

    def to_json(self, **kwargs):
        """The configuration as JSON text; numpy scalars become Python numbers, types their names
        (hyperion/hyp_model.py:117-131)."""
        def get_json_type(obj):
            if type(obj).__module__ == np.__name__:
                return obj.item()
            if type(obj).__name__ == type.__name__:
                return obj.__name__
            raise TypeError("Not JSON Serializable:", obj)

        return json.dumps(self.get_config(), default=get_json_type, **kwargs)

    # ---- model files (hyperion/hyp_model.py:44-135) --------------------------------------------
    def save(self, file_path):
        """Writes the model file the reference writes: a scalar string dataset "config" with the
        JSON configuration, then the parameters (hyperion/hyp_model.py:44-52)."""
        from .io.hdf5_write import File

        file_dir = os.path.dirname(file_path)
        if file_dir and not os.path.isdir(file_dir):
            os.makedirs(file_dir, exist_ok=True)
        with File(file_path, "w") as f:
            f.create_dataset("config", data=np.array(self.to_json(), dtype="S"))
            self.save_params(f)

    def save_params(self, f):
        """Models without parameters write nothing beyond the configuration."""

    def _save_params_from_dict(self, f, params, dtypes=None):
        """One dataset `name/parameter` per array that is not None, stored as float_save()
        unless `dtypes` says otherwise (hyperion/hyp_model.py:58-73)."""
        prefix = "" if self.name is None else self.name + "/"
        for k, v in params.items():
            if v is None:
                continue
            dtype = float_save() if dtypes is None else dtypes[k]
            f.create_dataset(prefix + k, data=np.asarray(v).astype(dtype, copy=False))

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
