"""A scripted editing session (what rfw::system::synchronize does over an application's lifetime, system.cpp:247-433)
replayed call for call on the core and on the oracle: after every edit the parity-integrator image and the primary
hits must agree.  Exercises the host logic that single-shot tests never reach: relayout after a topology change of
one mesh among several, refit of a resident mesh after others moved, instance edits, material / light / sky swaps,
target resize, integrator switches, a posed mesh surviving all of it."""
import numpy as np
import pytest

from conftest import image_stats


def _session(pkg, core, ref, w=96, h=64, check_hits=True):
    S = pkg.scenes
    scene = S.cornell(w, h)
    ctxs = (core, ref)
    log = []

    def both(fn):
        for c in ctxs:
            fn(c)

    def check(label, tol=1e-3, frac_max=5e-3, pt=False):
        for c in ctxs:
            c.update()
            c.render_frame(scene.camera, pkg.RESET)
        a, b = core.framebuffer(), ref.framebuffer()
        frac, rmse, _ = image_stats(a, b, 2e-2 if pt else tol)
        assert frac <= (2e-2 if pt else frac_max), (label, frac, rmse)
        if check_hits and not pt:
            ha, hb = core.primary_hits(), ref.primary_hits()
            assert (ha["prim"] != hb["prim"]).mean() <= 3e-3, label
            assert (ha["inst"] != hb["inst"]).mean() <= 3e-3, label
        assert np.isfinite(a).all(), label
        log.append((label, frac))

    both(lambda c: (c.init(w, h), scene.upload(c), c.set_setting("integrator", "parity"), c.set_setting("jitter", "center")))
    check("initial upload")

    # 1. one mesh among three changes topology (a finer box): relayout, the others keep their place
    bv, bi = S._box((-0.5, 0.0, -0.5), (0.5, 1.0, 0.5))
    def subdivide(v, idx):
        v = list(map(tuple, v)); out = []
        for a, b, c in idx:
            ab = tuple((np.array(v[a]) + np.array(v[b])) / 2); bc = tuple((np.array(v[b]) + np.array(v[c])) / 2)
            ca = tuple((np.array(v[c]) + np.array(v[a])) / 2)
            base = len(v); v += [ab, bc, ca]
            out += [(a, base, base + 2), (base, b, base + 1), (base + 2, base + 1, c), (base, base + 1, base + 2)]
        return np.array(v, np.float32), np.array(out, np.uint32)
    fv, fi = subdivide(bv, bi)
    mat_box = scene.meshes[1]["triangles"]["material"][0]
    v4 = np.ones((len(fv), 4), np.float32); v4[:, :3] = fv
    tris = S.make_triangles(v4, fi, material=mat_box)
    scene.meshes[1] = dict(vertices=v4, indices=fi, triangles=tris)
    both(lambda c: c.set_mesh(1, v4, tris, fi))
    check("box mesh rebuilt with 4x the triangles")

    # 2. instance edits: move one box, re-point the other instance at the room mesh scaled down
    scene.instances[1]["transform"] = S._translate(-2.4, 0.0, 0.4) @ S._rot_y(40) @ np.diag([2.0, 5.0, 2.0, 1.0])
    both(lambda c: c.set_instance(1, 1, scene.instances[1]["transform"]))
    check("instance moved")
    scene.instances[2] = dict(mesh=0, transform=S._translate(2.2, 0.0, -1.0) @ np.diag([0.12, 0.12, 0.12, 1.0]))
    both(lambda c: c.set_instance(2, 0, scene.instances[2]["transform"]))
    check("instance re-pointed at another mesh")

    # 3. same-count vertex edit of the big mesh: the refit path while other meshes are resident
    room = scene.meshes[0]
    rv = room["vertices"].copy(); rv[:, 1] *= 1.1
    rt = S.make_triangles(rv, room["indices"], material=room["triangles"]["material"])
    scene.meshes[0] = dict(vertices=rv, indices=room["indices"], triangles=rt)
    both(lambda c: c.set_mesh(0, rv, rt, room["indices"]))
    check("room refit (same counts)")

    # 4. a new mesh and instance at fresh indices
    qv = S.quad((0.0, 0.0, -1.0), (0.0, 4.0, 4.6), 3.0, 2.0)
    nm = scene.add_material(color=(0.2, 0.3, 0.8), roughness=0.8)
    mats, ids = S.pack_materials(scene.host_materials, scene.textures)
    both(lambda c: c.set_materials(mats, ids))
    qm = scene.add_mesh(qv, None, material=nm)
    qi = scene.add_instance(qm)
    both(lambda c: (c.set_mesh(qm, scene.meshes[qm]["vertices"], scene.meshes[qm]["triangles"], None),
                    c.set_instance(qi, qm, scene.instances[qi]["transform"])))
    check("mesh + instance added")

    # 5. lights and sky replaced
    scene.point_lights = [(np.array((2.5, 6.0, -2.0), np.float32), np.array((9.0, 7.0, 5.0), np.float32))]
    scene.set_test_sky(64, 32, base=0.3)
    a, p, s_, d = scene.light_arrays()
    both(lambda c: (c.set_lights(a, p, s_, d), c.set_sky(*scene.sky)))
    check("lights and sky replaced")

    # 6. resize, then the path tracer, then back
    scene.camera.resize(80, 52)
    both(lambda c: c.init(80, 52))
    check("resized target")
    both(lambda c: (c.set_setting("integrator", "pt"), c.set_setting("spp", 8)))
    check("path tracer after all edits", pt=True)
    both(lambda c: (c.set_setting("integrator", "parity"), c.set_setting("spp", 1)))
    check("back to the parity integrator")
    return log


def test_editing_session_emulated_core(pkg, make_emu, make_oracle):
    log = _session(pkg, make_emu(), make_oracle())
    assert len(log) == 10


@pytest.mark.gpu
def test_editing_session_gpu(pkg, make_hip, make_oracle):
    log = _session(pkg, make_hip(), make_oracle(), 480, 270)
    assert len(log) == 10


def test_material_index_out_of_range_fails_at_update(pkg, make_emu):
    """The shade kernels index the material table with the triangles' ids: rfwhip_update refuses a scene whose triangles
    point past the materials that were set (instead of a page fault on the device)."""
    import pytest
    s = pkg.scenes.cornell(32, 24)
    s.meshes[0]["triangles"]["material"][3] = 99
    ctx = make_emu()
    ctx.init(32, 24)
    with pytest.raises(RuntimeError, match="material"):
        s.upload(ctx)
