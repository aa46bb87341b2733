"""pytest configuration: markers, repo root on sys.path, oracle build."""
from __future__ import annotations

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a machine without an AMD GPU skips the gpu-marked tests instead of failing them."""
    try:
        import torch

        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs an MI355X (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure), built on demand with oracle/Makefile."""
    from oracle import oracle as o

    o.lib()
    return o


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


@pytest.fixture(autouse=True)
def _poisoned_lds(request):
    """GPU tests start with every compute unit's LDS full of 1e30 (`wedm_debug_poison_lds`): the step kernels never
    write the rows of their LDS image that lie past a wire's end, and a result that leaned on one would otherwise depend
    on what the previous test's kernels left there (it did once: a regular tile let the interior-formula value of a
    last cell, computed from such a row, into the maximum temperature — caught only when the test order changed)."""
    if "gpu" in request.keywords:
        import torch

        if torch.cuda.is_available():
            from sparc_amd import _lib

            assert _lib.load().wedm_debug_poison_lds(1e30, None) == 0
            torch.cuda.synchronize()
    yield
