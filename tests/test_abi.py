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
