#!/usr/bin/env python3
"""Golden vectors for OBJ faces with more than four vertices (tinyobjloader's built-in ear
clipping, tiny_obj_loader.h:1540-1819), produced by running the REFERENCE loader chain
(loadScenePrebuilt(Custom) → loadMesh → tinyobjloader, through oracle/_ref) on a synthetic
custom.obj written here. Container only; output: tests/golden/polygons.npz (the OBJ text and
the arrays the reference produced)."""
import sys, tempfile
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle.cpu_oracle import Reference, SCENE_TYPES

OBJ = """# polygons: convex pentagon (xy plane), concave hexagon (xz plane), octagon (tilted), an L-shaped
# concave 8-gon (yz plane), a star-like 10-gon, and a quad + triangle for the other paths
mtllib custom.mtl
usemtl a
v 0 0 0
v 2 0 0
v 2.6 1.4 0
v 1 2.5 0
v -0.7 1.3 0
f 1 2 3 4 5
v 0 1 0
v 3 1 0
v 3 1 3
v 1.5 1 1.2
v 0 1 3
v -1 1 1.5
f 6 7 8 9 10 11
v 1 0 0.1
v 0.7 0.7 0.2
v 0 1 0.35
v -0.7 0.7 0.3
v -1 0 0.2
v -0.7 -0.7 0.05
v 0 -1 0
v 0.7 -0.7 0.0
f 12 13 14 15 16 17 18 19
v 5 0 0
v 5 3 0
v 5 3 1
v 5 1 1
v 5 1 2
v 5 2 2
v 5 2 3
v 5 0 3
usemtl b
f 20 21 22 23 24 25 26 27
v 0.0 2.0 4
v 0.5 0.6 4
v 2.0 0.6 4
v 0.8 -0.3 4
v 1.2 -1.8 4
v 0.0 -0.9 4
v -1.2 -1.8 4
v -0.8 -0.3 4
v -2.0 0.6 4
v -0.5 0.6 4
f 28 29 30 31 32 33 34 35 36 37
f 1 2 3 4
f 5 4 3
"""
MTL = "newmtl a\nKd 0.8 0.3 0.2\nKs 0.5 0.5 0.5\nNs 20\nnewmtl b\nKd 0.1 0.6 0.9\nd 0.5\n"

ref = Reference()
with tempfile.TemporaryDirectory() as d:
    (Path(d) / "custom.obj").write_text(OBJ)
    (Path(d) / "custom.mtl").write_text(MTL)
    import oracle.cpu_oracle as co
    co.REFERENCE_ROOT = Path(d).parent / "__none__"  # not used below
    h = ref.lib.ref_scene_load(SCENE_TYPES["custom"], (d + "/").encode())
    assert h, "reference failed to load the synthetic custom.obj"
    ref._scenes["custom"] = h
    scene = ref.load_scene("custom")
out = {"obj": np.frombuffer(OBJ.encode(), np.uint8), "mtl": np.frombuffer(MTL.encode(), np.uint8),
       "n_meshes": np.int32(len(scene.meshes))}
for i, m in enumerate(scene.meshes):
    out[f"m{i}_positions"], out[f"m{i}_normals"], out[f"m{i}_texcoords"] = m.positions, m.normals, m.texcoords
    out[f"m{i}_triangles"] = m.triangles
    out[f"m{i}_material"] = np.float32(list(m.material.kd) + list(m.material.ks) + [m.material.shininess, m.material.transparency])
np.savez_compressed(ROOT / "tests" / "golden" / "polygons.npz", **out)
print("meshes", len(scene.meshes), [len(m.triangles) for m in scene.meshes])
