"""CPU-side checks of the boundary: the C-ABI library loads without a GPU and exports every symbol include/tracehip.h
declares; computing without a GPU fails loudly (no CPU fallback)."""
import os
import re
import subprocess

import pytest


def test_every_declared_symbol_is_exported(T):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "tracehip.h")).read()
    declared = set(re.findall(r"\b(trhip_[a-z0-9_]+)\s*\(", header))
    declared -= {"trhip_ctx", "trhip_scene", "trhip_status", "trhip_sensor", "trhip_stats", "trhip_hit"}
    assert len(declared) >= 30
    out = subprocess.check_output(["nm", "-D", "--defined-only", T._ffi.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (trhip_[a-z0-9_]+)", out))
    assert declared <= exported, f"declared but not exported: {sorted(declared - exported)}"
    assert declared == set(T._ffi.SIGNATURES), f"ctypes table out of sync: {sorted(declared ^ set(T._ffi.SIGNATURES))}"
    lib = T.lib()
    assert lib.trhip_version() == 3001


def test_no_cpu_fallback(T):
    """Without a GPU trhip_init must fail with a message; nothing silently routes to the CPU."""
    import torch
    if torch.cuda.is_available() or os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    with pytest.raises(T.TraceHipError) as e:
        T.Context(0)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_product_does_not_reference_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "trace.jl_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".jl")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in src and "oracle_bridge" not in src and "/oracle/" not in src, f"{f} references the oracle"


def test_option_in_build_is_a_property_of_the_binary(T):
    """trhip_option_in_build needs no context and no GPU: what tests/conftest.py decides the collection on.  Options of the default path exist in every build; the kernel
    families behind traversal 4 / 6 / 7, leaf_queue, leaf_sorted and bvh_builder 1 exist together or not at all (-DTRHIP_EXPERIMENTS)."""
    lib = T.lib()
    for name, value in (("traversal", 1), ("traversal", 2), ("traversal", 3), ("leaf_queue", 0), ("leaf_sorted", 0), ("bvh_builder", 0), ("bvh_builder", 2), ("bvh_builder", -1),
                        ("overlap", 1), ("no_such_option", 5)):
        assert lib.trhip_option_in_build(name.encode(), value) == 1, (name, value)
    experiments = [lib.trhip_option_in_build(n.encode(), v) for n, v in (("traversal", 4), ("traversal", 6), ("traversal", 7), ("leaf_queue", 1), ("leaf_sorted", 1), ("bvh_builder", 1))]
    assert experiments in ([0] * 6, [1] * 6), experiments
    assert lib.trhip_option_in_build(None, 0) == 0
