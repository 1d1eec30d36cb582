"""BVH construction and refit of the product (through the host-emulation build)."""
import numpy as np

from conftest import image_stats


def _check_tree(nodes, prims, ntri, max_leaf=4):
    assert sorted(prims.tolist()) == list(range(ntri))          # every primitive exactly once
    seen = np.zeros(ntri, bool)
    stack = [(0, 0)]
    depth_max = 0
    while stack:
        i, d = stack.pop()
        depth_max = max(depth_max, d)
        n = nodes[i]
        if n["count"] >= 0:
            assert 1 <= n["count"] <= max_leaf
            sl = slice(n["left_first"], n["left_first"] + n["count"])
            assert not seen[sl].any()
            seen[sl] = True
        else:
            l = n["left_first"]
            assert l % 2 == 0                                    # child pairs are 64-byte aligned
            for c in (l, l + 1):
                assert np.all(nodes[c]["bmin"] >= n["bmin"] - 1e-4) and np.all(nodes[c]["bmax"] <= n["bmax"] + 1e-4)
                stack.append((c, d + 1))
    assert seen.all()
    return depth_max


def test_bvh_invariants_terrain(pkg, make_emu):
    scene = pkg.scenes.terrain(n=48, width=64, height_px=48, lights=False)
    e = make_emu()
    e.init(64, 48)
    scene.upload(e)
    nodes, prims = e.get_bvh(0)
    ntri = len(scene.meshes[0]["triangles"])
    depth = _check_tree(nodes, prims, ntri)
    assert depth <= 42
    # leaves bound their triangles
    v = scene.meshes[0]["vertices"][:, :3]
    idx = scene.meshes[0]["indices"]
    for n in nodes[nodes["count"] > 0][:200]:
        for p in prims[n["left_first"]:n["left_first"] + n["count"]]:
            tri = v[idx[p]]
            assert np.all(tri.min(0) >= n["bmin"] - 1e-6) and np.all(tri.max(0) <= n["bmax"] + 1e-6)


def test_degenerate_inputs_still_build(pkg, make_emu, make_oracle):
    """Many coincident triangles (identical centroids) force the median-split fallback."""
    s = pkg.scenes.Scene()
    s.add_material(color=(0.8, 0.8, 0.8))
    tri = np.array([[-1, 0, 3], [1, 0, 3], [0, 1.5, 3]], np.float32)
    v = np.tile(tri, (40, 1))
    s.add_instance(s.add_mesh(v, None))
    s.add_point_light((0, 1, -2), (10, 10, 10))
    s.set_test_sky(32, 16)
    cam = pkg.Camera(aperture=0.0)
    cam.look_at((0, 0.5, -3), (0, 0.5, 3))
    cam.resize(32, 24)
    s.camera = cam
    e, o = make_emu(), make_oracle()
    for c in (e, o):
        c.init(32, 24)
        s.upload(c)
        c.set_setting("jitter", "center")
        c.render_frame(cam, pkg.RESET)
    nodes, prims = e.get_bvh(0)
    _check_tree(nodes, prims, 40)
    a, b = e.primary_hits(), o.primary_hits()
    assert np.array_equal(a["prim"] >= 0, b["prim"] >= 0)
    assert np.abs(a["t"] - b["t"])[a["prim"] >= 0].max() < 1e-5


def test_refit_equals_rebuild(pkg, make_emu, make_oracle):
    """BASELINE config 5 logic: re-sending a mesh with unchanged counts refits on the device; the image must equal a
    fresh build of the new pose (and the oracle's)."""
    w, h = 96, 64
    s0 = pkg.scenes.skinned_tube(frame=0.0, rings=24, seg=16, width=w, height=h)
    s1 = pkg.scenes.skinned_tube(frame=3.0, rings=24, seg=16, width=w, height=h)
    e = make_emu()
    e.init(w, h)
    s0.upload(e)
    e.set_setting("jitter", "center")
    e.render_frame(s0.camera, pkg.RESET)
    before = e.framebuffer()
    m = s1.meshes[0]
    e.set_mesh(0, m["vertices"], m["triangles"], m["indices"])   # same counts => refit path
    e.update()
    e.render_frame(s1.camera, pkg.RESET)
    refit = e.framebuffer()
    nodes, prims = e.get_bvh(0)
    _check_tree(nodes, prims, len(m["triangles"]))
    fresh, o = make_emu(), make_oracle()
    for c in (fresh, o):
        c.init(w, h)
        s1.upload(c)
        c.set_setting("jitter", "center")
        c.render_frame(s1.camera, pkg.RESET)
    assert image_stats(refit, fresh.framebuffer(), 1e-4)[0] <= 1e-3
    assert image_stats(refit, o.framebuffer(), 1e-3)[0] <= 2e-3
    assert image_stats(refit, before, 1e-3)[0] > 0.01            # the pose really changed
    # refit boxes contain the moved triangles
    v = m["vertices"][:, :3]
    for n in nodes[nodes["count"] > 0][:100]:
        for p in prims[n["left_first"]:n["left_first"] + n["count"]]:
            tri = v[m["indices"][p]]
            assert np.all(tri.min(0) >= n["bmin"] - 1e-6) and np.all(tri.max(0) <= n["bmax"] + 1e-6)
