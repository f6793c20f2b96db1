"""load_triangle_mesh(path, core) (model_loader.jl:1-11) without Assimp: the PLY subset docs/src/assets/models uses
(binary little-endian, vertex x y z nx ny nz, faces as `list uint8 int`), plus ascii and big-endian variants."""
import struct

import numpy as np
import pytest

VERTS = np.float32([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0.5]])
NORMS = np.float32([[0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0.4472136, 0.8944272]])
FACES = [[0, 1, 2], [0, 2, 3]]


def write_ply(path, fmt):
    header = ["ply", f"format {fmt} 1.0", "comment made by a test", "element vertex 4", "property float x", "property float y", "property float z", "property float nx",
              "property float ny", "property float nz", "element face 2", "property list uchar int vertex_indices", "end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode())
        if fmt == "ascii":
            for v, n in zip(VERTS, NORMS):
                f.write((" ".join(repr(float(x)) for x in list(v) + list(n)) + "\n").encode())
            for t in FACES:
                f.write(("3 " + " ".join(map(str, t)) + "\n").encode())
        else:
            e = "<" if fmt == "binary_little_endian" else ">"
            for v, n in zip(VERTS, NORMS):
                f.write(struct.pack(e + "6f", *v, *n))
            for t in FACES:
                f.write(struct.pack(e + "B3i", 3, *t))


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian", "binary_big_endian"])
def test_read_ply_and_load_triangle_mesh(T, tmp_path, fmt):
    path = str(tmp_path / f"m_{fmt}.ply")
    write_ply(path, fmt)
    v, n, f = T.read_ply(path)
    assert np.array_equal(v, VERTS) and np.array_equal(n, NORMS) and f.tolist() == FACES
    meshes, tris = T.load_triangle_mesh(path, T.ShapeCore(T.translate([1, 2, 3]), False))
    assert len(meshes) == 1 and len(tris) == 2
    mesh = meshes[0]
    assert mesh.indices.reshape(-1, 3).tolist() == [[1, 2, 3], [1, 3, 4]]        # 1-based (model_loader.jl:36)
    assert np.array_equal(mesh.vertices, VERTS + np.float32([1, 2, 3]))          # vertices go to world space (triangle_mesh.jl:23)
    assert np.array_equal(mesh.normals, NORMS)                                   # normals do not (triangle_mesh.jl:23-28)
    assert [t.k for t in tris] == [0, 1] and all(t.mesh is mesh for t in tris)


def test_ply_errors(T, tmp_path):
    p = tmp_path / "bad.ply"
    p.write_bytes(b"not a ply\n")
    with pytest.raises(ValueError):
        T.read_ply(str(p))
    q = tmp_path / "quad.ply"
    q.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n"
                  b"0 0 0\n1 0 0\n1 1 0\n0 1 0\n4 0 1 2 3\n")
    with pytest.raises(ValueError, match="Only triangles supported"):
        T.read_ply(str(q))
    r = tmp_path / "nonormals.ply"
    r.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n"
                  b"0 0 0\n1 0 0\n1 1 0\n3 0 1 2\n")
    with pytest.raises(ValueError, match="normals"):
        T.load_triangle_mesh(str(r))


def test_reference_caustic_glass_ply(T):
    """The reference's one mesh asset (docs/src/assets/models/caustic-glass.ply, placed under tests/golden/ by make_caustic_ply.py),
    read by load_triangle_mesh as docs/code/caustic_glass.jl:21-24 does: 44 034 vertices with normals, 88 064 triangles, the
    bounding box SURVEY.md §8d quotes, then moved by the script's translate(5, -1.49, -100)."""
    import hashlib
    import json
    import os
    import numpy as np
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    path = os.path.join(here, "caustic-glass.ply")
    meta = json.load(open(path + ".json"))
    blob = open(path, "rb").read()
    assert len(blob) == meta["bytes"] and hashlib.sha256(blob).hexdigest() == meta["sha256"]
    verts, normals, faces = T.api.read_ply(path)
    assert verts.shape == (44034, 3) and normals.shape == (44034, 3) and faces.shape == (88064, 3)
    assert faces.dtype == np.uint32 and int(faces.max()) == 44033 and int(faces.min()) == 0
    np.testing.assert_allclose(verts.min(0), [-4.81, 1.50, 1.35], atol=0.01)
    np.testing.assert_allclose(verts.max(0), [-2.66, 3.50, 3.50], atol=0.01)
    np.testing.assert_allclose(np.linalg.norm(normals, axis=1), 1.0, atol=1e-3)
    meshes, triangles = T.load_triangle_mesh(path, T.ShapeCore(T.translate([5, -1.49, -100]), False))
    assert len(meshes) == 1 and len(triangles) == 88064
    mesh = triangles[0].mesh
    moved = (verts.astype(np.float32) + np.float32([5, -1.49, -100])).astype(np.float32)
    assert np.abs(mesh.vertices - moved).max() <= 1e-5  # world-space vertices (triangle_mesh.jl:23); normals stay untransformed (:23-28)
    assert np.array_equal(mesh.normals, normals)
    assert np.array_equal(mesh.indices.reshape(-1, 3), faces + 1)  # 1-based (model_loader.jl:36)
    scene = T.scenes.caustic_scene(path)
    assert len(scene.aggregate.primitives[0].mesh.indices) // 3 == 88064 and len(scene.lights) == 1
