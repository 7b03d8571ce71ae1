"""The C ABI from a caller that is not Python: examples/native_caller.c (plain C99, gcc) links
libhypmi.so through include/hypmi.h, builds a plan, runs prologue + scoring and checks the matrix
against its own double-precision evaluation of the closed form."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "native_caller")


def _ensure_built():
    """Built by __graft_entry__.build(); a tree that has the library but not the example gets it
    here (one gcc call on host code, a second or two)."""
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, capture_output=True)


def test_c_caller_is_built_against_the_library():
    """The header compiles as C and the executable resolves the library next to the package."""
    _ensure_built()
    out = subprocess.run(["ldd", EXE], capture_output=True, text=True, check=True).stdout
    assert "libhypmi.so" in out and "not found" not in out.split("libhypmi.so")[1].splitlines()[0]


@pytest.mark.gpu
def test_c_caller_scores_match_its_own_closed_form():
    _ensure_built()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "native caller: 70 x 131 trials" in r.stdout and "bad ld -> -1" in r.stdout
