import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # the configuration the package runs in (maest_amd/__init__.py), set before any HIP call

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture
def emu():
    """Bind the host SIMT-emulator build of the kernel sources for the duration of one test."""
    from tests.emu import build_emu
    from maest_amd import _lib
    if not build_emu.available():
        pytest.skip("host clang for the emulator build is not available")
    path = build_emu.build()
    _lib._testing_override(path)
    yield "cpu"
    _lib._testing_restore()


@pytest.fixture
def gemm_options():
    """``gemm_options(gemm_min_m=512)``: set library switches (maest_set_option) for one test, restored afterwards.
    Request it AFTER `emu` so that it acts on the emulator build and is undone before that is unbound."""
    from maest_amd import ops
    stack = []

    def set_(**kw):
        o = ops.options(**kw)
        o.__enter__()
        stack.append(o)
    yield set_
    for o in reversed(stack):
        o.__exit__()
