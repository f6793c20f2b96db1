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


EXPERIMENTS_NOTE = "needs the EXPERIMENTS build"


def supported(ctx, name, values):
    """The values of an option this build of the library honours (tests that sweep kernel variants sweep what exists).  Leaves the option at the last supported value."""
    return [v for v in values if ctx.has_option(name, v)]


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_setup(item):
    outcome = yield
    _skip_if_experiments_only(outcome)


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    outcome = yield
    _skip_if_experiments_only(outcome)


def _skip_if_experiments_only(outcome):
    """A test (or fixture) that asks for a kernel family the default library does not carry is SKIPPED, not failed: run it against the EXPERIMENTS build
    (TRHIP_LIB=trace.jl_amd/libtracehip_experiments.so python -m pytest tests -m gpu)."""
    exc = outcome.excinfo
    if exc is not None and EXPERIMENTS_NOTE in str(exc[1]):
        outcome.force_exception(pytest.skip.Exception(str(exc[1])[-160:], _use_item_location=True))
