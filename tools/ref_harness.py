"""Stub-import harness for the read-only reference.

The reference needs `gymnasium` and `numba`, neither of which is installed here.
This module injects two tiny in-memory stand-ins into ``sys.modules`` (an identity
``njit`` and a minimal ``gym.Env`` whose ``reset(seed=)`` seeds
``np.random.Generator(PCG64(SeedSequence(seed)))`` exactly like Gymnasium does) and
then imports the reference package from ``/root/reference/src``.

It is used ONLY by ``tools/gen_golden.py`` in the build container to produce the
fixtures under ``tests/golden/``.  Nothing in the product, the tests, ``bench.py`` or
``__graft_entry__`` imports this file; the reference never travels to the GPU box.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import types

import numpy as np

REFERENCE_SRC = "/root/reference/src"


def _install_stubs() -> None:
    if "numba" not in sys.modules:
        numba = types.ModuleType("numba")

        def njit(*args, **kwargs):
            if len(args) == 1 and callable(args[0]) and not kwargs:
                return args[0]
            return lambda fn: fn

        numba.njit = njit
        numba.prange = range
        sys.modules["numba"] = numba

    if "gymnasium" not in sys.modules:
        gym = types.ModuleType("gymnasium")
        spaces = types.ModuleType("gymnasium.spaces")

        class Env:
            _np_random = None

            @property
            def np_random(self):
                if self._np_random is None:
                    self._np_random = np.random.default_rng()
                return self._np_random

            @np_random.setter
            def np_random(self, value):
                self._np_random = value

            def reset(self, *, seed=None, options=None):
                if seed is not None:
                    self._np_random = np.random.Generator(
                        np.random.PCG64(np.random.SeedSequence(seed))
                    )

        class Box:
            def __init__(self, low, high, shape=None, dtype=np.float32):
                self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

        class Dict(dict):
            def __init__(self, spaces_dict=None):
                super().__init__(spaces_dict or {})

        gym.Env = Env
        spaces.Box = Box
        spaces.Dict = Dict
        gym.spaces = spaces
        sys.modules["gymnasium"] = gym
        sys.modules["gymnasium.spaces"] = spaces


def import_reference():
    """Return the reference's ``wedm`` package (imported with the stubs above)."""
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    _install_stubs()
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import wedm  # noqa: WPS433

    return wedm


def quiet(fn, *args, **kwargs):
    """Call ``fn`` swallowing the reference's constructor prints."""
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*args, **kwargs)
