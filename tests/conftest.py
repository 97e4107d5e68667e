import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session", autouse=True)
def _native_stack_on_abort():
    """The GPU runtimes abort() on fatal errors, not always with a message: have the library print the native call stack of the aborting
    thread first (pytest's faulthandler then adds the interpreter's).  Diagnostics only; never fails a run."""
    try:
        from dqnflappybird_amd import _lib
        out = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out):                   # (the GPU box: what is written there comes back with the call)
            os.environ.setdefault("FB_ABORT_LOG", os.path.join(out, "abort_trace.txt"))
        _lib.lib().fb_debug_abort_backtrace()
    except Exception:
        pass
    yield
