#!/usr/bin/env python
"""tools/build_variant.py NAME [-DFLAG=V ...]: the library compiled with extra flags into _diag/lib_NAME.so (tuning A/B: TRHIP_LIB=$PWD/_diag/lib_NAME.so)."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

name = sys.argv[1]
out = graft.build_library(extra_flags=sys.argv[2:], out_name=f"libvariant_{name}.so")
os.makedirs(os.path.join(ROOT, "_diag"), exist_ok=True)
shutil.move(out, os.path.join(ROOT, "_diag", f"lib_{name}.so"))
if os.path.exists(out + ".flags"):
    os.remove(out + ".flags")
print(os.path.join("_diag", f"lib_{name}.so"))
shutil.rmtree(os.path.join(ROOT, "trace.jl_amd", "csrc", f"obj_libvariant_{name}_so"), ignore_errors=True)  # (11 MB of objects per variant, and gpurun ships the tree)
