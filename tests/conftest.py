import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def T():
    """The package (directory trace.jl_amd) as module trace_jl_amd; builds libtracehip.so / liboracle.so if stale."""
    graft.build_library()
    graft.build_oracle()
    return graft.load_package()


@pytest.fixture(scope="session")
def ob(T):
    import oracle_bridge
    oracle_bridge.lib()
    return oracle_bridge


@pytest.fixture(scope="session")
def ctx(T):
    """GPU context; fails loudly (no CPU fallback) when the extension cannot see an MI355X."""
    return T.default_context()
