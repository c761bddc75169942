import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The shared libraries are build artefacts (git-ignored): build them in-tree when a fresh checkout lacks them
    (hipcc cross-compiles gfx950 without a GPU).  A failing build is reported by the tests that need the library."""
    lib = os.path.join(ROOT, "gauss_amd", "lib")
    if all(os.path.exists(os.path.join(lib, f)) for f in ("libgauss_hip.so", "libgauss_host.so")):
        return
    try:
        from gauss_amd import build
        build.build_host()
    except Exception as ex:      # pragma: no cover
        print(f"[conftest] could not build the libraries: {ex}", file=sys.stderr)


@pytest.fixture(scope="session")
def ctx():
    """One HIP context for the whole GPU session -- the process's default context, so that entry points called without
    ctx= share it: a second live context on the device makes the library queue its runs in the two-launch form (its streams
    would share hardware queues with the first one's, gauss_ctx.cpp), and the suite is meant to exercise the defaults.
    Fails loudly if the library is missing."""
    from gauss_amd import hotpath
    c = hotpath.default_context()
    yield c
    c.close()
    hotpath._default_ctx = None
