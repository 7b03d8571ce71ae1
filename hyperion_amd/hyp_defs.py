"""Global float types of the host API (same contract as hyperion/hyp_defs.py:9-38): results handed
back to numpy callers are `float_cpu()` arrays (float64 unless changed), while the device path
computes in fp32 on the MI355X matrix cores."""
import logging

import numpy as np

_FLOAT_CPU = "float64"
_FLOAT_KERAS = "float32"
_FLOAT_SAVE = "float32"


def float_cpu():
    return _FLOAT_CPU


def set_float_cpu(float_cpu):
    global _FLOAT_CPU
    _FLOAT_CPU = float_cpu


def float_save():
    return _FLOAT_SAVE


def set_float_save(float_save):
    global _FLOAT_SAVE
    _FLOAT_SAVE = float_save


def float_device():
    """Arithmetic type of the HIP kernels."""
    return np.float32


def float_keras():
    """Kept for scripts that ask for it (hyperion/hyp_defs.py:23-29); nothing here reads it."""
    return _FLOAT_KERAS


def set_float_keras(float_keras):
    global _FLOAT_KERAS
    _FLOAT_KERAS = float_keras


logging_levels = {0: logging.WARN, 1: logging.INFO, 2: logging.DEBUG, 3: 5}


def config_logger(verbose_level):
    """Root logger at the level the scripts' `-v` option names (hyperion/hyp_defs.py:41-49)."""
    logging.basicConfig(level=logging_levels[verbose_level],
                        format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s")
