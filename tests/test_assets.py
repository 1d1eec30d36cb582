"""Scene ingestion: the .glb container, morph targets, the OBJ / MTL reader, and the assets BASELINE config 5 and
SURVEY name — CesiumMan (skinned glTF), AnimatedMorphCube.glb, legocar.obj — through committed fixtures
(tests/golden/asset_*.npz, written by tests/golden/make_golden_assets.py from the reference checkout's assets).

CPU tier: the front ends on files the tests write themselves (so they also run where the reference checkout is absent), the
front ends on the REAL assets against the fixtures (skipped without /root/reference), and the device skinning / morphing code
(host-emulation build) on the CesiumMan and MorphCube rigs against the float64 poses stored in the fixtures.
GPU tier: the same two rigs posed by the HIP kernels, rendered, and compared with the oracle rendering the fixture's posed mesh.
"""
import json
import os
import struct

import numpy as np
import pytest

from conftest import ROOT, image_stats

GOLD = os.path.join(ROOT, "tests", "golden")
ASSETS = "/root/reference/assets/models"
needs_assets = pytest.mark.skipif(not os.path.isdir(ASSETS), reason="reference checkout absent (the GPU box): fixtures only")


# ---- files written by the test --------------------------------------------------------------------------------------------
def _write_morph_glb(path):
    """A quad with two morph targets (POSITION + NORMAL displacements) and a LINEAR "weights" animation, as a binary .glb."""
    pos = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (4, 1))
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint16)
    t0p = np.array([[0, 0, 0.5], [0, 0, 0], [0, 0, 0.5], [0, 0, 0]], np.float32)
    t0n = np.array([[0.1, 0, 0], [0, 0, 0], [0.1, 0, 0], [0, 0, 0]], np.float32)
    t1p = np.array([[0, 0, 0], [0.25, 0, 0], [0.25, 0, 0], [0, 0, 0]], np.float32)
    t1n = np.zeros((4, 3), np.float32)
    times = np.array([0.0, 1.0, 2.0], np.float32)
    wkeys = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]], np.float32).reshape(-1)
    blobs, views, accessors = [], [], []

    def add(arr, comp, typ):
        raw = np.ascontiguousarray(arr).tobytes()
        off = sum(len(b) for b in blobs)
        blobs.append(raw + b"\0" * ((-len(raw)) % 4))
        views.append({"buffer": 0, "byteOffset": off, "byteLength": len(raw)})
        accessors.append({"bufferView": len(views) - 1, "componentType": comp, "count": len(arr), "type": typ})
        return len(accessors) - 1

    a = {k: add(v, 5126, "VEC3") for k, v in (("p", pos), ("n", nrm), ("t0p", t0p), ("t0n", t0n), ("t1p", t1p), ("t1n", t1n))}
    ai, at, aw = add(idx, 5123, "SCALAR"), add(times, 5126, "SCALAR"), add(wkeys, 5126, "SCALAR")
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
           "nodes": [{"mesh": 0, "translation": [0.0, 0.5, 0.0]}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": a["p"], "NORMAL": a["n"]}, "indices": ai,
                                       "targets": [{"POSITION": a["t0p"], "NORMAL": a["t0n"]}, {"POSITION": a["t1p"], "NORMAL": a["t1n"]}]}],
                       "weights": [0.25, 0.5]}],
           "animations": [{"samplers": [{"input": at, "output": aw, "interpolation": "LINEAR"}],
                           "channels": [{"sampler": 0, "target": {"node": 0, "path": "weights"}}]}],
           "accessors": accessors, "bufferViews": views, "buffers": [{"byteLength": sum(len(b) for b in blobs)}]}
    js = json.dumps(doc).encode("utf-8")
    js += b" " * ((-len(js)) % 4)
    binary = b"".join(blobs)
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(binary)))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(binary), 0x004E4942) + binary)
    return pos, nrm, (t0p, t0n), (t1p, t1n)


def test_glb_container_and_morph_targets(pkg, tmp_path):
    path = str(tmp_path / "morph.glb")
    pos, nrm, t0, t1 = _write_morph_glb(path)
    g = pkg.gltf.Gltf(path)
    ni = g.mesh_nodes()[0]
    pr = g.primitive(g.nodes[ni]["mesh"])
    assert np.array_equal(pr["positions"], pos) and len(pr["targets"]) == 2 and pr["indices"].shape == (2, 3)
    assert np.allclose(g.W[ni], [0.25, 0.5])  # the mesh's default weights
    p, n = g.morphed(ni)
    assert np.allclose(p, pos + 0.25 * t0[0] + 0.5 * t1[0]) and np.allclose(n, nrm + 0.25 * t0[1])  # normals not renormalised
    for t, w in ((0.5, (0.5, 0.0)), (1.0, (1.0, 0.0)), (1.5, (0.5, 0.5)), (2.0, (0.0, 1.0)), (2.5, (0.5, 0.0))):  # wraps with fmod
        g.set_time(t)
        assert np.allclose(g.W[ni], w, atol=1e-6), (t, g.W[ni])
        p, _ = g.morphed(ni)
        assert np.allclose(p, pos + w[0] * t0[0] + w[1] * t1[0], atol=1e-6)
    with open(path, "r+b") as f:  # a version the format does not define
        f.seek(4)
        f.write(struct.pack("<I", 3))
    with pytest.raises(ValueError):
        pkg.gltf.Gltf(path)


def test_obj_reader_on_a_written_file(pkg, tmp_path):
    (tmp_path / "two.mtl").write_text("newmtl red\nKd 0.8 0.1 0.1\nNs 30\n\nnewmtl blue\nKd 0.1 0.1 0.9\nmap_Kd tex/blue.png\n")
    (tmp_path / "two.obj").write_text(
        "mtllib two.mtl\n"
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nv 1 0 1\n"
        "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
        "vn 0 0 1\n"
        "usemtl red\ns 1\nf 1/1/1 2/2/1 3/3/1 4/4/1\n"       # a quad with normals and uvs: a fan of two triangles
        "usemtl blue\ns off\nf -2 -1 3\nf 5 6 2 1\n"          # relative indices, no normals: generated; a second quad
        "usemtl red\nf 1/1/1 3/3/1 4/4/1\n")                 # back to the first material: same mesh
    mats, meshes = pkg.obj.load_obj(str(tmp_path / "two.obj"))
    assert [m[0] for m in mats] == ["red", "blue"] and mats[1][1]["map_Kd"] == "tex/blue.png"
    red, blue = meshes
    assert len(red["indices"]) == 3 and len(red["vertices"]) == 4          # joined: 4 unique corners for 3 triangles
    assert np.allclose(red["normals"], [0, 0, 1]) and np.allclose(red["uvs"][red["indices"][0]], [[0, 0], [1, 0], [1, 1]])
    assert len(blue["indices"]) == 3
    n = blue["normals"]
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-5)           # generated, unit length
    s, _, _ = pkg.obj.load_scene(str(tmp_path / "two.obj"), 32, 24)
    assert s.triangle_count() == 6 and np.allclose(s.host_materials[0]["color"], (0.8, 0.1, 0.1))


# ---- the real assets against the fixtures (development container) ------------------------------------------------------------
@needs_assets
def test_reference_assets_parse_to_the_committed_fixtures(pkg):
    fx = np.load(os.path.join(GOLD, "asset_cesiumman.npz"))
    g = pkg.gltf.Gltf(os.path.join(ASSETS, "CesiumMan", "CesiumMan.gltf"))
    ni = [i for i in g.mesh_nodes() if "skin" in g.nodes[i]][0]
    pr = g.primitive(g.nodes[ni]["mesh"])
    assert np.array_equal(pr["positions"], fx["positions"]) and np.array_equal(pr["indices"], fx["indices"])
    assert np.array_equal(pr["joints"], fx["joints"]) and np.allclose(pr["weights"], fx["weights"])
    for k, t in enumerate(fx["times"]):
        g.set_time(float(t))
        assert np.allclose(g.joint_matrices(ni), fx["joint_matrices"][k], atol=1e-6)
    fm = np.load(os.path.join(GOLD, "asset_morphcube.npz"))
    g = pkg.gltf.Gltf(os.path.join(ASSETS, "AnimatedMorphCube.glb"))
    ni = g.mesh_nodes()[0]
    for k, t in enumerate(fm["times"]):
        g.set_time(float(t))
        p, n = g.morphed(ni)
        assert np.allclose(g.W[ni], fm["weights"][k], atol=1e-6)
        assert np.allclose(p, fm["morphed_positions"][k], atol=1e-7) and np.allclose(n, fm["morphed_normals"][k], atol=1e-6)
    fo = np.load(os.path.join(GOLD, "asset_obj.npz"))
    for name in ("legocar", "sphere"):
        mats, meshes = pkg.obj.load_obj(os.path.join(ASSETS, name + ".obj"))
        assert [len(m["indices"]) for m in meshes] == list(fo[name + "_tris"])
        assert np.allclose([m[1]["Kd"] for m in mats], fo[name + "_kd"])
    assert int(fo["legocar_tris"].sum()) == 10992


def test_cesiumman_fixture_is_self_consistent():
    """What the file itself states pins the parse: the POSITION accessor's min / max; weights sum to 1; every joint index is
    a joint; and the float64 skinning keeps bones rigid (posed edge lengths of triangles bound to ONE joint do not change)."""
    fx = np.load(os.path.join(GOLD, "asset_cesiumman.npz"))
    p = fx["positions"]
    assert np.allclose(p.min(0), fx["file_min"], atol=1e-6) and np.allclose(p.max(0), fx["file_max"], atol=1e-6)
    assert p.shape == (3273, 3) and fx["indices"].shape == (4672, 3)
    # (this asset's weights do not all sum to 1: 6.4 % of its vertices carry a truncated influence set; the file's own accessor
    # bounds say so too — the sums are what they are, only their range is checked)
    ws = fx["weights"].sum(1)
    assert ws.max() <= 1.0 + 1e-5 and ws.min() > 0.1 and (np.abs(ws - 1.0) < 1e-5).mean() > 0.9
    assert fx["joints"].max() < fx["joint_matrices"].shape[1]
    rigid = (fx["weights"].max(1) > 0.999)
    tri = fx["indices"]
    one = rigid[tri].all(1) & (fx["joints"][tri, fx["weights"][tri].argmax(-1)].std(1) == 0)
    assert one.sum() > 500
    e0 = np.linalg.norm(p[tri[one, 1]] - p[tri[one, 0]], axis=1)
    for k in range(len(fx["times"])):
        q = fx["posed_positions"][k]
        e1 = np.linalg.norm(q[tri[one, 1]] - q[tri[one, 0]], axis=1)
        assert np.allclose(e0, e1, rtol=2e-3, atol=1e-5)
    assert np.abs(fx["posed_positions"][1] - fx["posed_positions"][0]).max() > 0.05  # the animation really moves it


# ---- the rigs on the device code ----------------------------------------------------------------------------------------------
def _rig_scene(pkg, positions, normals, indices, transform, w, h):
    s = pkg.scenes.Scene()
    s.name = "rig"
    body = s.add_material(color=(0.75, 0.55, 0.35), roughness=0.6)
    floor = s.add_material(color=(0.6, 0.6, 0.6), roughness=0.9)
    s.add_instance(s.add_mesh(positions, indices, normals=normals, material=body), transform)
    lo = (np.asarray(transform, np.float64)[:3, :3] @ positions.T.astype(np.float64)).T + np.asarray(transform, np.float64)[:3, 3]
    c, r = (lo.min(0) + lo.max(0)) / 2, float(np.linalg.norm(lo.max(0) - lo.min(0)))
    y0 = float(lo[:, 1].min()) - 0.02 * r
    fv = np.array([[c[0] - 2 * r, y0, c[2] - 2 * r], [c[0] + 2 * r, y0, c[2] - 2 * r], [c[0] + 2 * r, y0, c[2] + 2 * r], [c[0] - 2 * r, y0, c[2] + 2 * r]], np.float32)
    s.add_instance(s.add_mesh(fv, np.array([[0, 2, 1], [0, 3, 2]], np.uint32), material=floor))
    s.add_point_light((c[0] + r, c[1] + 1.5 * r, c[2] - 1.2 * r), (40.0 * r * r, 38.0 * r * r, 35.0 * r * r))
    s.add_area_light_quad((0.0, -1.0, 0.0), (c[0], c[1] + 2.0 * r, c[2]), r, r, (10.0, 10.0, 10.0))
    s.set_test_sky(64, 32)
    cam = pkg.Camera(aperture=0.0, FOV=40.0)
    cam.look_at((c[0] + 0.3 * r, c[1] + 0.2 * r, c[2] - 2.2 * r), tuple(c))
    cam.resize(w, h)
    s.camera = cam
    return s


def _posed_copy(pkg, scene, positions, normals, indices):
    import copy
    s = copy.copy(scene)
    s.meshes = [dict(m) for m in scene.meshes]
    v4 = np.ones((len(positions), 4), np.float32)
    v4[:, :3] = positions
    s.meshes[0]["vertices"] = v4
    s.meshes[0]["triangles"] = pkg.scenes.make_triangles(positions, indices, normals=normals, material=scene.meshes[0]["triangles"]["material"][0])
    return s


def _check_rig(pkg, live, make_ref, scene, w, h, poses, settings, tol_frac):
    """poses: iterable of (apply(live), posed positions, posed normals): the live context is posed on the device, a fresh
    oracle renders the same pose handed over as an ordinary mesh."""
    idx = scene.meshes[0]["indices"]
    for apply, pv, pn in poses:
        apply(live)
        live.update()
        live.render_frame(scene.camera, pkg.RESET)
        ref = make_ref()
        ref.init(w, h)
        _posed_copy(pkg, scene, pv, pn, idx).upload(ref)
        for k, v in settings.items():
            ref.set_setting(k, v)
        ref.render_frame(scene.camera, pkg.RESET)
        a, b = live.primary_hits(), ref.primary_hits()
        assert (a["prim"] != b["prim"]).mean() <= 2e-3 and (a["inst"] != b["inst"]).mean() <= 1e-3
        same = (a["prim"] == b["prim"]) & (a["prim"] >= 0)
        assert np.abs(a["t"][same] - b["t"][same]).max() <= 2e-4 * max(1.0, float(b["t"][same].max()))
        frac, rmse, _ = image_stats(live.framebuffer(), ref.framebuffer(), 1e-3)
        assert frac <= tol_frac, (frac, rmse)


def _cesium(pkg, make_live, make_ref, w, h, settings, tol_frac):
    fx = np.load(os.path.join(GOLD, "asset_cesiumman.npz"))
    scene = _rig_scene(pkg, fx["positions"], fx["normals"], fx["indices"], fx["node_transform"], w, h)
    live = make_live()
    live.init(w, h)
    scene.upload(live)
    for k, v in settings.items():
        live.set_setting(k, v)
    live.set_mesh_skin(0, fx["joints"], fx["weights"], fx["normals"])
    poses = [((lambda c, k=k: c.pose_mesh(0, fx["joint_matrices"][k])), fx["posed_positions"][k], fx["posed_normals"][k])
             for k in range(len(fx["times"]))]
    _check_rig(pkg, live, make_ref, scene, w, h, poses, settings, tol_frac)


def _morphcube(pkg, make_live, make_ref, w, h, settings, tol_frac):
    fm = np.load(os.path.join(GOLD, "asset_morphcube.npz"))
    scene = _rig_scene(pkg, fm["positions"], fm["normals"], fm["indices"], fm["node_transform"], w, h)
    live = make_live()
    live.init(w, h)
    scene.upload(live)
    for k, v in settings.items():
        live.set_setting(k, v)
    live.set_mesh_morph(0, fm["normals"], fm["target_positions"], fm["target_normals"])
    poses = [((lambda c, k=k: c.morph_mesh(0, fm["weights"][k])), fm["morphed_positions"][k], fm["morphed_normals"][k])
             for k in range(len(fm["times"]))]
    _check_rig(pkg, live, make_ref, scene, w, h, poses, settings, tol_frac)


PARITY = {"integrator": "parity", "jitter": "center"}


def test_cesiumman_posed_by_the_device_code_emulation(pkg, make_emu, make_oracle):
    _cesium(pkg, make_emu, make_oracle, 96, 128, PARITY, 3e-3)


def test_morph_cube_morphed_by_the_device_code_emulation(pkg, make_emu, make_oracle):
    _morphcube(pkg, make_emu, make_oracle, 96, 96, PARITY, 3e-3)


def test_morph_api_errors(pkg, make_emu):
    fm = np.load(os.path.join(GOLD, "asset_morphcube.npz"))
    scene = _rig_scene(pkg, fm["positions"], fm["normals"], fm["indices"], fm["node_transform"], 32, 32)
    ctx = make_emu()
    ctx.init(32, 32)
    scene.upload(ctx)
    with pytest.raises(RuntimeError):
        ctx.morph_mesh(0, [0.5, 0.5])  # no targets yet
    ctx.set_mesh_morph(0, fm["normals"], fm["target_positions"], fm["target_normals"])
    with pytest.raises(RuntimeError):
        ctx.morph_mesh(0, [0.5])  # two targets, one weight


@pytest.mark.gpu
def test_cesiumman_posed_on_the_gpu(pkg, make_hip, make_oracle):
    """BASELINE config 5's asset: skinning, shading normals, refit and the re-quantisation of the 4-wide nodes on the device,
    at three animation times, against the oracle rendering the fixture's float64-posed mesh."""
    _cesium(pkg, make_hip, make_oracle, 360, 480, PARITY, 3e-3)


@pytest.mark.gpu
def test_morph_cube_morphed_on_the_gpu(pkg, make_hip, make_oracle):
    _morphcube(pkg, make_hip, make_oracle, 256, 256, PARITY, 3e-3)
