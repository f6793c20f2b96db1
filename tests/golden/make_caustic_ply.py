#!/usr/bin/env python
"""Places the reference's one mesh asset — docs/src/assets/models/caustic-glass.ply, the model docs/code/caustic_glass.jl:21-24
loads — under tests/golden/ as DATA (a fixture: 44 034 vertices with normals, 88 064 triangles, binary little-endian PLY).

Run in the build container, where /root/reference exists; the GPU box only sees the copy.  The file is copied byte for byte
(so the PLY reader is exercised on exactly what the reference ships) and its SHA-256 is recorded next to it.

    python tests/golden/make_caustic_ply.py
"""
import hashlib
import json
import os
import shutil

SRC = "/root/reference/docs/src/assets/models/caustic-glass.ply"
HERE = os.path.dirname(os.path.abspath(__file__))
DST = os.path.join(HERE, "caustic-glass.ply")

if __name__ == "__main__":
    shutil.copyfile(SRC, DST)
    blob = open(DST, "rb").read()
    header = blob[: blob.index(b"end_header") + len(b"end_header")].decode("ascii", "replace").splitlines()
    meta = {"source": "docs/src/assets/models/caustic-glass.ply (pxl-th/Trace.jl @ 2024_10_08)", "bytes": len(blob), "sha256": hashlib.sha256(blob).hexdigest(),
            "header": header}
    json.dump(meta, open(os.path.join(HERE, "caustic-glass.ply.json"), "w"), indent=1)
    print(meta["bytes"], meta["sha256"])
