"""CPU: mesh-side host logic — OBJ loader mirror (examples/julia-raytracer.jl:466-602), transform order, generator."""
import numpy as np

from spira_hip import raytracer as R
from spira_hip import scenes


def test_icosphere_counts_and_closedness():
    for level, nf in [(0, 20), (1, 80), (3, 1280)]:
        v, f = scenes.icosphere(level)
        assert len(f) == nf and len(v) == nf // 2 + 2 and np.allclose(np.linalg.norm(v, axis=1), 1.0)
        edges = {}
        for a, b, c in f:
            for e in ((a, b), (b, c), (c, a)):
                edges[tuple(sorted(e))] = edges.get(tuple(sorted(e)), 0) + 1
        assert set(edges.values()) == {2}                      # closed 2-manifold
        n = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
        assert np.all(np.einsum("ij,ij->i", n, v[f].mean(axis=1)) > 0)   # outward winding (semantics A never flips normals)


def test_transform_order_matches_reference():
    v = np.array([[0.0, 0, 0], [2, 0, 0], [0, 4, 0], [0, 0, 1]])
    # centre (bbox mid = (1,2,.5)), normalise by 4, rotate Y 90 deg (x' = z, z' = -x), scale .5, translate z-1
    out = scenes.transform_vertices(v, scale=(0.5, 0.5, 0.5), rotation=(0, 90.0, 0), translation=(0, 0, -1.0), center=True, normalize_size=True)
    c = (v - [1, 2, 0.5]) / 4.0
    want = np.stack([c[:, 2], c[:, 1], -c[:, 0]], axis=1) * 0.5 + [0, 0, -1.0]
    assert np.allclose(out, want, atol=1e-15)
    assert np.array_equal(scenes.transform_vertices(v, center=False), v)


def test_obj_loader_roundtrip(tmp_path):
    v, f = scenes.bumpy_blob(2)
    path = tmp_path / "m.obj"
    with open(path, "w") as fh:
        fh.write("# test mesh\n")
        for p in v:
            fh.write("v %.17g %.17g %.17g\n" % tuple(p))
        fh.write("vn 0 0 1\n")
        for i, (a, b, c) in enumerate(f):
            if i % 2:
                fh.write("f %d/1/1 %d/1/1 %d/1/1\n" % (a + 1, b + 1, c + 1))   # v/vt/vn form (:493)
            else:
                fh.write("f %d %d %d\n" % (a + 1, b + 1, c + 1))
        fh.write("f 1 2 3 4\n")                                                # quad -> fan of two triangles (:500-505)
    m = R.Material(diffuse=R.Vec3(0.7, 0.3, 0.2), specular=0.2, roughness=0.4)
    tris = R.load_obj_mesh(str(path), m, center=True, normalize_size=True, scale=R.Vec3(0.5, 0.5, 0.5),
                           rotation=R.Vec3(0.0, 90.0, 0.0), translation=R.Vec3(0.0, 0.0, -1.0))
    assert len(tris) == len(f) + 2 and all(t.material is m for t in tris)
    tv = scenes.transform_vertices(v, scale=(0.5, 0.5, 0.5), rotation=(0, 90.0, 0), translation=(0, 0, -1.0), center=True, normalize_size=True)
    assert np.allclose(tris[5].vertices[1].tolist(), tv[f[5, 1]])
    assert [tris[-2].vertices[k].tolist() for k in range(3)] == [tv[0].tolist(), tv[1].tolist(), tv[2].tolist()]
    assert [tris[-1].vertices[k].tolist() for k in range(3)] == [tv[0].tolist(), tv[2].tolist(), tv[3].tolist()]
    scene, cam = R.create_scene_with_obj(str(path))
    sp, ma, tr = R.flatten_world(scene)
    assert len(sp) == 2 and len(ma) == 3 and len(tr) == len(tris) and set(tr[:, 9]) == {3.0}
    scene2, _ = R.create_scene_with_obj(str(tmp_path / "missing.obj"))        # sphere fallback (:687-691)
    sp2, ma2, tr2 = R.flatten_world(scene2)
    assert len(sp2) == 3 and tr2 is None


def test_scene_s4_shape():
    s = scenes.scene_s4(level=2)
    assert s["triangles10"].shape == (320, 10) and set(s["triangles10"][:, 9]) == {3.0}
    zs = s["triangles10"][:, [2, 5, 8]]
    assert -1.3 < zs.min() and zs.max() < -0.7          # scaled to a 0.5 box around z = -1
