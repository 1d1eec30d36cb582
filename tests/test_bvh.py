"""BVH construction and refit of the product (through the host-emulation build)."""
import numpy as np
import pytest

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


def _device_vs_host(pkg, make_ctx, make_oracle, scene, w, h, check_tree=True):
    """builder=device (lbvh.hip: Morton order, Karras hierarchy, device fit) must give a valid tree in the reference's
    node layout and the same closest hits / image as the host-built SAH tree and the oracle."""
    dev, host, ref = make_ctx(), make_ctx(), make_oracle()
    dev.set_setting("builder", "device")
    out = []
    for c in (dev, host, ref):
        c.init(w, h)
        scene.upload(c)
        c.set_setting("jitter", "center")
        c.render_frame(scene.camera, pkg.RESET)
        out.append((c.primary_hits(), c.framebuffer()))
    if check_tree:
        for mi, m in enumerate(scene.meshes):
            ntri = len(m["triangles"])
            nodes, prims = dev.get_bvh(mi)
            if ntri > 4:
                assert len(nodes) % 2 == 0 and len(nodes) >= 2 * ((ntri + 3) // 4)  # the device layout: one node pair per chunk
            _check_tree(nodes, prims, ntri)
    (a, ia), (b, ib), (r, ir) = out
    for other in (b, r):
        assert (a["prim"] != other["prim"]).mean() <= 2e-3
        same = (a["prim"] == other["prim"]) & (a["prim"] >= 0)
        assert (np.abs(a["t"][same] - other["t"][same]) <= 1e-4 + 2e-5 * np.abs(other["t"][same])).all()
    assert image_stats(ia, ib, 1e-3)[0] <= 5e-3
    assert image_stats(ia, ir, 1e-3)[0] <= 5e-3
    return dev


def test_device_builder_terrain_and_instances(pkg, make_emu, make_oracle):
    scene = pkg.scenes.terrain(n=40, width=96, height_px=64, lights=False)
    # lights handed over through set_lights only: shadow rays that END on emitter geometry are a coin toss in fp32
    scene.add_area_light_quad((0.0, -1.0, 0.0), (0.0, 30.0, 0.0), 6.0, 6.0, (400.0, 380.0, 350.0))
    scene.add_point_light((10.0, 20.0, -10.0), (900.0, 900.0, 800.0))
    _device_vs_host(pkg, make_emu, make_oracle, scene, 96, 64)
    # instanced boxes + room: several small meshes (some with <= 4 triangles take the host path inside)
    _device_vs_host(pkg, make_emu, make_oracle, pkg.scenes.cornell(96, 64), 96, 64)


def test_device_builder_then_refit(pkg, make_emu, make_oracle):
    """A device-built mesh refits like a host-built one (same parent links / flags machinery)."""
    w, h = 64, 48
    base = pkg.scenes.skinned_tube(0.0, rings=20, seg=12, width=w, height=h)
    live = make_emu()
    live.set_setting("builder", "device")
    live.init(w, h)
    base.upload(live)
    live.set_setting("jitter", "center")
    pose = pkg.scenes.skinned_tube(3.0, rings=20, seg=12, width=w, height=h)
    m = pose.meshes[0]
    live.set_mesh(0, m["vertices"], m["triangles"], m["indices"])
    live.update()
    live.render_frame(pose.camera, pkg.RESET)
    ref = make_oracle()
    ref.init(w, h)
    pose.upload(ref)
    ref.set_setting("jitter", "center")
    ref.render_frame(pose.camera, pkg.RESET)
    assert image_stats(live.framebuffer(), ref.framebuffer(), 1e-3)[0] <= 5e-3
    nodes, prims = live.get_bvh(0)
    _check_tree(nodes, prims, len(m["triangles"]))


def _soup(kind, rng, n):
    if kind == "uniform":
        c = rng.uniform(-10, 10, (n, 1, 3)); e = rng.normal(0, 0.4, (n, 3, 3))
    elif kind == "clustered":
        centers = rng.uniform(-20, 20, (6, 3))
        c = centers[rng.integers(0, 6, n)][:, None, :] + rng.normal(0, 0.3, (n, 1, 3)); e = rng.normal(0, 0.05, (n, 3, 3))
    elif kind == "slivers":           # long thin triangles, all overlapping each other's boxes
        c = rng.uniform(-2, 2, (n, 1, 3)); e = rng.normal(0, 1.0, (n, 3, 3)) * np.array([30.0, 0.02, 0.02])
    elif kind == "comb":              # self-similar: positions and sizes grow geometrically -> SAH peels the big end off,
        g = 1.02 ** np.arange(n)      # one small group at a time: the deepest trees the builders make (depth limit 32)
        c = np.stack([g, 0.1 * g * rng.normal(0, 1, n), 0.1 * g * rng.normal(0, 1, n)], -1)[:, None, :]
        e = rng.normal(0, 0.15, (n, 3, 3)) * g[:, None, None]
    elif kind == "far_from_origin":   # fp32 cancellation: coordinates ~1e4, sizes ~1
        c = 1.0e4 + rng.uniform(-50, 50, (n, 1, 3)); e = rng.normal(0, 1.0, (n, 3, 3))
    else:                             # "duplicates": every triangle four times + a few degenerate (zero-area) ones
        base = rng.uniform(-5, 5, (n // 4, 1, 3)) + rng.normal(0, 0.5, (n // 4, 3, 3))
        v = np.concatenate([base] * 4)
        v[:3, 1] = v[:3, 0]
        return v.reshape(-1, 3).astype(np.float32)
    return (c + e).reshape(-1, 3).astype(np.float32)


@pytest.mark.parametrize("kind", ["uniform", "clustered", "slivers", "far_from_origin", "duplicates", "comb"])
@pytest.mark.parametrize("builder", ["host", "device"])
def test_random_soups_against_brute_force(pkg, make_emu, make_oracle, kind, builder):
    _soup_case(pkg, make_emu, make_oracle, kind, builder)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["uniform", "clustered", "slivers", "far_from_origin", "duplicates", "comb"])
@pytest.mark.parametrize("builder", ["host", "device"])
def test_random_soups_against_brute_force_gpu(pkg, make_hip, make_oracle, kind, builder):
    _soup_case(pkg, make_hip, make_oracle, kind, builder)


def _soup_case(pkg, make_emu, make_oracle, kind, builder):
    """Triangle soups the builders do not like, traced with rays aimed at the geometry: the product's BVH (either
    builder, then the 4-wide collapse) must find what the oracle finds by testing EVERY triangle (bvh=0)."""
    rng = np.random.default_rng(sum(map(ord, kind + builder)))
    verts = _soup(kind, rng, 600)
    s = pkg.scenes.Scene()
    s.add_material(color=(0.8, 0.8, 0.8))
    s.add_instance(s.add_mesh(verts, None))
    s.set_test_sky(16, 8)
    cam = pkg.Camera(aperture=0.0)
    lo, hi = verts.min(0), verts.max(0)
    cam.look_at(tuple(lo - (hi - lo)), tuple((lo + hi) / 2))
    cam.resize(16, 16)
    s.camera = cam
    core, ref = make_emu(), make_oracle()
    core.set_setting("builder", builder)
    ref.set_setting("bvh", 0)
    for c in (core, ref):
        c.init(16, 16)
        s.upload(c)
    nodes, prims = core.get_bvh(0)
    _check_tree(nodes, prims, len(verts) // 3)
    n = 4000
    tri_c = verts.reshape(-1, 3, 3).mean(1)
    org = (tri_c[rng.integers(0, len(tri_c), n)] + rng.normal(0, 1.0, (n, 3)) * (hi - lo) * 0.7).astype(np.float32)
    tgt = tri_c[rng.integers(0, len(tri_c), n)] + rng.normal(0, 0.05, (n, 3))
    d = tgt - org
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    a, b = core.trace_rays(org, d), ref.trace_rays(org, d)
    if kind == "comb":
        # rays travel up to 1e5 triangle sizes here: a hit within an ulp of a triangle's edge may fall either way
        assert ((a["prim"] >= 0) != (b["prim"] >= 0)).mean() <= 1e-3
    else:
        assert np.array_equal(a["prim"] >= 0, b["prim"] >= 0)
    hit = (a["prim"] >= 0) & (b["prim"] >= 0)
    assert hit.mean() > 0.1
    scale = float(np.abs(verts).max())
    # (comb: where the two sides settle on different teeth of an edge-on pair, the distances differ too — compared where they agree)
    same = hit & (a["prim"] == b["prim"]) if kind == "comb" else hit
    assert (np.abs(a["t"][same] - b["t"][same]) <= 1e-6 * scale + (5e-5 if kind == "comb" else 2e-5) * np.abs(b["t"][same])).all()
    # the same primitive unless two triangles tie at the hit distance (the duplicates soup is all ties)
    if kind != "duplicates":
        assert (a["prim"][hit] != b["prim"][hit]).mean() <= (1e-2 if kind == "comb" else 5e-3)
