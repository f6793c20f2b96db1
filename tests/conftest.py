import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experiments(option, value): the test (or this parameter of it) drives a kernel family that only the EXPERIMENTS build of the library "
                            "carries (traversal 4 / 6 / 7, leaf_queue, leaf_sorted, bvh_builder 1); DESELECTED at collection when the loaded binary lacks it")


def option_in_build(name, value) -> bool:
    """trhip_option_in_build: a property of the loaded binary (no context, no GPU)."""
    graft.build_library()
    return bool(graft.load_package().lib().trhip_option_in_build(name.encode(), int(value)))


def experiments(name, value):
    """Mark for a test or a pytest.param: exists only in a build that carries `name = value`."""
    return pytest.mark.experiments(name, value)


def pytest_collection_modifyitems(config, items):
    """Variants the loaded library does not carry are not in the run at all (deselected, reported as such) — decided from the BINARY at collection, not from the text of an
    exception at run time: a test of the default path that fails with whatever message FAILS.  Run them against the EXPERIMENTS build with
    TRHIP_LIB=trace.jl_amd/libtracehip_experiments.so python -m pytest tests -m gpu."""
    keep, drop = [], []
    for item in items:
        missing = [m.args for m in item.iter_markers("experiments") if not option_in_build(*m.args)]
        (drop if missing else keep).append(item)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep



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


def supported(ctx, name, values):
    """The values of an option this build of the library honours (tests that sweep kernel variants inside one test sweep what exists).  Leaves the options alone."""
    return [v for v in values if option_in_build(name, v)]
