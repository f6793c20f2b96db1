"""trace_jl_amd — MI355X-native wavefront path tracing behind Trace.jl's Integrator / Sampler / Film surface.

The directory is named ``trace.jl_amd`` (not importable by that name); load it as module ``trace_jl_amd`` with
``__graft_entry__.load_package()`` (tests/conftest.py and bench.py do).  Layout:

    csrc/        hand-written HIP kernels + the C ABI (libtracehip.so, include/tracehip.h)
    _ffi.py      ctypes binding of the C ABI
    api.py       Python mirror of the Trace.jl API that scene scripts use
    scenes.py    workload definitions: shadows (docs/src/shadows.md), Cornell, synthetic N-triangle meshes
    julia/       TraceHIP.jl: the ccall shim a Trace.jl user would load (cannot be executed in this image)
"""
from ._ffi import Context, Sensor, Stats, TraceHipError, default_context, lib  # noqa: F401
from .api import *  # noqa: F401,F403
from . import api, parallel, scenes  # noqa: F401
