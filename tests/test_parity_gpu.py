"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances: the parity integrator is deterministic per pixel — RGB L2 <= 1e-3 on >= 99.9 % of pixels,
RMSE <= 1e-3, primitive ids identical on >= 99.95 % of pixels (silhouette flips counted, not excluded).  The
path-tracing integrator follows discrete random decisions, so a 1-ulp difference in sin/cos/exp can send a path
elsewhere: >= 99 % of pixels within 2e-2 at 16 spp and image RMSE <= 3e-2.
"""
import numpy as np
import pytest

from conftest import image_stats

pytestmark = pytest.mark.gpu


def _pair(pkg, make_hip, make_oracle, scene, w, h, settings):
    out = []
    for ctx in (make_hip(), make_oracle()):
        ctx.init(w, h)
        scene.upload(ctx)
        for k, v in settings.items():
            ctx.set_setting(k, v)
        ctx.render_frame(scene.camera, pkg.RESET)
        out.append(ctx)
    return out


@pytest.mark.parametrize("jitter", ["center", "xor128"])
def test_cornell_parity_integrator(pkg, make_hip, make_oracle, jitter):
    scene = pkg.scenes.cornell(512, 512)
    hip, ref = _pair(pkg, make_hip, make_oracle, scene, 512, 512, {"integrator": "parity", "jitter": jitter, "spp": 1})
    frac, rmse, _ = image_stats(hip.framebuffer(), ref.framebuffer(), 1e-3)
    assert frac <= 1e-3 and rmse <= 1e-3, (frac, rmse)
    a, b = hip.primary_hits(), ref.primary_hits()
    assert (a["prim"] != b["prim"]).mean() <= 5e-4
    assert (a["inst"] != b["inst"]).mean() <= 5e-4
    same = (a["prim"] == b["prim"]) & (a["prim"] >= 0)
    assert np.abs(a["t"][same] - b["t"][same]).max() <= 1e-3
    assert hip.get_probe_results()[:2] == ref.get_probe_results()[:2]


def test_cornell_parity_64spp_accumulate(pkg, make_hip, make_oracle):
    """BASELINE config 2 (reduced size for the oracle's sake): 64 jittered samples, xor128 stream from the default
    seed; HIP accumulates 4 batches of 16, the oracle 64 single frames."""
    scene = pkg.scenes.cornell(320, 240)
    hip, ref = make_hip(), make_oracle()
    for ctx, spp, calls in ((hip, 16, 4), (ref, 1, 64)):
        ctx.init(320, 240)
        scene.upload(ctx)
        ctx.set_setting("integrator", "parity")
        ctx.set_setting("spp", spp)
        for k in range(calls):
            ctx.render_frame(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
    frac, rmse, _ = image_stats(hip.framebuffer(), ref.framebuffer(), 1e-3)
    assert frac <= 2e-3 and rmse <= 1e-3, (frac, rmse)


def test_cornell_path_tracer(pkg, make_hip, make_oracle):
    scene = pkg.scenes.cornell(256, 192, geometric_emitter=True)
    hip, ref = _pair(pkg, make_hip, make_oracle, scene, 256, 192, {"integrator": "pt", "spp": 16})
    a, b = hip.framebuffer(), ref.framebuffer()
    assert np.isfinite(a).all()
    frac, rmse, _ = image_stats(a, b, 2e-2)
    assert frac <= 1e-2 and rmse <= 3e-2, (frac, rmse)
    assert abs(a[..., :3].mean() - b[..., :3].mean()) <= 2e-3 * b[..., :3].mean()
    sa, sb = hip.get_stats(), ref.get_counters()
    assert sa.primaryCount == 256 * 192 * 16


def test_terrain_parity_hits(pkg, make_hip, make_oracle):
    """A 20 k-triangle cut of the BASELINE config-3 mesh: closest-hit records against the oracle's own BVH."""
    # parity scenes keep emitters out of the geometry: the oracle's shadow rays end exactly on the light's centroid
    # (Context.cpp:231,241), so a geometric emitter self-occludes by last-ulp luck (SURVEY §7 hard part b)
    scene = pkg.scenes.terrain(n=100, width=480, height_px=270, lights=False)
    scene.add_area_light_quad((0.0, -1.0, 0.0), (5.0, 14.0, -8.0), 6.0, 6.0, (30.0, 28.0, 24.0))
    scene.add_point_light((-20.0, 12.0, 10.0), (300.0, 280.0, 260.0))
    hip, ref = _pair(pkg, make_hip, make_oracle, scene, 480, 270, {"integrator": "parity", "jitter": "center"})
    a, b = hip.primary_hits(), ref.primary_hits()
    assert (a["prim"] != b["prim"]).mean() <= 1e-3
    frac, rmse, _ = image_stats(hip.framebuffer(), ref.framebuffer(), 1e-3)
    assert frac <= 2e-3, (frac, rmse)


def test_deterministic_and_idempotent(pkg, make_hip):
    """Size-independent properties at the full BASELINE resolution: rendering twice gives bit-identical images
    (no atomics on radiance), RESET really resets, and CONVERGE of the same sample set averages."""
    scene = pkg.scenes.cornell(1920, 1080, geometric_emitter=True)
    ctx = make_hip()
    ctx.init(1920, 1080)
    scene.upload(ctx)
    ctx.set_setting("integrator", "pt")
    ctx.set_setting("spp", 2)
    ctx.render_frame(scene.camera, pkg.RESET)
    a = ctx.framebuffer()
    ctx.render_frame(scene.camera, pkg.RESET)
    b = ctx.framebuffer()
    assert np.array_equal(a, b)
    assert np.isfinite(a).all() and a[..., 3].min() == 1.0


def test_errors_are_loud(pkg, make_hip):
    ctx = make_hip()
    with pytest.raises(RuntimeError):
        ctx.render_frame(pkg.Camera(), pkg.RESET)  # no target yet
    with pytest.raises(RuntimeError):
        ctx.set_setting("integrator", "bogus")
    ctx.cleanup()
    ctx.cleanup()  # idempotent


def test_config2_cornell_1080p_64spp(pkg, make_hip, make_oracle):
    """BASELINE.json config 2 at its full size: Cornell scene, 1920x1080, 64 spp, parity integrator, xor128 jitter
    from the default seed — per-pixel L2 against the oracle (which has no accumulator: 64 single-sample frames).
    Stated tolerance: RGB L2 <= 1e-3 on >= 99.9 % of pixels and RMSE <= 1e-3."""
    scene = pkg.scenes.cornell(1920, 1080)
    hip, ref = make_hip(), make_oracle()
    for ctx, spp, calls in ((hip, 16, 4), (ref, 1, 64)):
        ctx.init(1920, 1080)
        scene.upload(ctx)
        ctx.set_setting("integrator", "parity")
        ctx.set_setting("spp", spp)
        for k in range(calls):
            ctx.render_frame(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
    a, b = hip.framebuffer(), ref.framebuffer()
    frac, rmse, d = image_stats(a, b, 1e-3)
    assert frac <= 1e-3 and rmse <= 1e-3, (frac, rmse, float(d.max()))


def test_atrium_textured_instanced(pkg, make_hip, make_oracle):
    """BASELINE config 4 stand-in (~264 k instanced triangles, 46 instances of 10 meshes, 25 materials, 12 mip-mapped
    textures) at reduced resolution: both integrators against the oracle."""
    scene = pkg.scenes.atrium(480, 270)
    assert 200_000 < scene.triangle_count() < 400_000
    # parity integrator: keep emitters out of the geometry (see test_terrain_parity_hits)
    par = pkg.scenes.atrium(480, 270)
    par.area_lights = par.area_lights[:0]
    par.instances = par.instances[:-1]          # drop the emissive quads' instance
    par.add_area_light_quad((0.0, -1.0, 0.0), (0.0, 12.0, 0.0), 3.0, 3.0, (25.0, 24.0, 20.0))
    hip, ref = _pair(pkg, make_hip, make_oracle, par, 480, 270, {"integrator": "parity", "jitter": "center"})
    a, b = hip.primary_hits(), ref.primary_hits()
    assert (a["prim"] != b["prim"]).mean() <= 2e-3 and (a["inst"] != b["inst"]).mean() <= 2e-3
    frac, rmse, _ = image_stats(hip.framebuffer(), ref.framebuffer(), 1e-3)
    assert frac <= 5e-3, (frac, rmse)
    hip, ref = _pair(pkg, make_hip, make_oracle, scene, 480, 270, {"integrator": "pt", "spp": 8})
    frac, rmse, _ = image_stats(hip.framebuffer(), ref.framebuffer(), 3e-2)
    assert frac <= 2e-2, (frac, rmse)
    assert abs(hip.framebuffer()[..., :3].mean() - ref.framebuffer()[..., :3].mean()) <= 5e-3 * ref.framebuffer()[..., :3].mean()


@pytest.mark.parametrize("faithful", [False, True])
def test_alpha_cards_layers_and_normal_maps(pkg, make_hip, make_oracle, faithful):
    """SURVEY §8 f1 on the GPU: alpha pass-through, additive diffuse layers, normal-map layers, per-slot texture ids."""
    scene = pkg.scenes.cards(480, 270, faithful=faithful)
    hip, ref = _pair(pkg, make_hip, make_oracle, scene, 480, 270, {"integrator": "pt", "spp": 8, "max_depth": 3})
    frac, rmse, _ = image_stats(hip.framebuffer(), ref.framebuffer(), 3e-2)
    assert frac <= 2e-2, (frac, rmse)
    m = ref.framebuffer()[..., :3].mean()
    assert abs(hip.framebuffer()[..., :3].mean() - m) <= 5e-3 * m
    st = hip.get_stats()
    oc = ref.get_counters()
    total = st.primaryCount + st.secondaryCount + st.deepCount
    assert abs(total - oc["rays_extend"]) <= 0.002 * oc["rays_extend"]


def test_device_skinning_equals_host_skinning(pkg, make_hip, make_oracle, orc):
    """SURVEY §8 f4 on the GPU: skin + shading normals + refit on the device == host skinning (oracle's restatement of
    SceneMesh::set_pose) + set_mesh, at config-5 size (30 720 triangles)."""
    from test_emu_parity import _host_skin
    w, h = 480, 270
    scene = pkg.scenes.skinned_tube(0.0, width=w, height=h)
    v, idx, vn, joints, weights = pkg.scenes.skinned_tube_rig()
    mat = scene.meshes[0]["triangles"]["material"][0]
    scene.meshes[0]["triangles"] = pkg.scenes.make_triangles(v, idx, normals=vn, material=mat)
    live = make_hip()
    live.init(w, h)
    scene.upload(live)
    live.set_setting("integrator", "pt")
    live.set_setting("spp", 4)
    live.set_setting("stage_timing", 1)
    live.set_mesh_skin(0, joints, weights, vn)
    for frame in (1.0, 3.5):
        mats = pkg.scenes.skinned_tube_joint_matrices(frame)
        live.pose_mesh(0, mats)
        live.update()
        live.render_frame(scene.camera, pkg.RESET)
        sv, sn = _host_skin(orc, v, vn, joints, weights, mats)
        posed = pkg.scenes.skinned_tube(0.0, width=w, height=h)
        m = posed.meshes[0]
        v4 = np.ones((len(sv), 4), np.float32)
        v4[:, :3] = sv
        m["vertices"] = v4
        m["triangles"] = pkg.scenes.make_triangles(sv, idx, normals=sn, material=mat)
        fresh, ref = _pair(pkg, make_hip, make_oracle, posed, w, h, {"integrator": "pt", "spp": 4})
        frac, rmse, _ = image_stats(live.framebuffer(), fresh.framebuffer(), 1e-3)
        assert frac <= 5e-3, (frame, frac, rmse)
        frac, rmse, _ = image_stats(live.framebuffer(), ref.framebuffer(), 3e-2)
        assert frac <= 2e-2, (frame, frac, rmse)
    ms, launches = live.get_kernel_time("refit")
    assert launches == 10 and ms > 0.0  # per pose: skin vertices, skin shading records, triangles, BVH2 boxes, Node4s


def test_stream_ordered_present_and_deinterleave(pkg, make_hip):
    """bench.py's N > 1 step without host synchronisation: two ranks' contexts on one device, each frame presented on a
    torch side stream (rfwhip_read_local_framebuffer_stream), "gathered" by a stream-ordered copy, de-interleaved on
    that stream while the next frame is already enqueued — the images of every step equal the synchronous path's."""
    import torch
    w, h, world, steps = 480, 270, 2, 4
    scene = pkg.scenes.cornell(w, h, geometric_emitter=True)
    ctxs = []
    for r in range(world):
        c = pkg.RenderContext(device=0, rank=r, world=world)
        c.init(w, h)
        scene.upload(c)
        c.set_setting("integrator", "pt")
        c.set_setting("spp", 4)
        ctxs.append(c)
    rows = ctxs[0].local_rows()
    dev = torch.device("cuda", 0)

    def run(pipelined):
        side = torch.cuda.Stream(device=dev)
        local = [torch.zeros((rows, w, 4), dtype=torch.float32, device=dev) for _ in range(world)]
        flat = torch.zeros((world, rows, w, 4), dtype=torch.float32, device=dev)
        fulls = [torch.zeros((h, w, 4), dtype=torch.float32, device=dev) for _ in range(steps)]
        with torch.cuda.stream(side):
            for k in range(steps):
                for r, c in enumerate(ctxs):
                    c.render_async(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
                    if pipelined:
                        c.read_local_framebuffer_stream(local[r].data_ptr(), side.cuda_stream)
                    else:
                        c.wait()
                        c.read_local_framebuffer_device(local[r].data_ptr())
                    flat[r].copy_(local[r], non_blocking=True)  # stands in for the gather's landing copy
                if pipelined:
                    ctxs[0].deinterleave_stream(flat.data_ptr(), fulls[k].data_ptr(), side.cuda_stream)
                else:
                    torch.cuda.synchronize()
                    ctxs[0].deinterleave_device(flat.data_ptr(), fulls[k].data_ptr())
        for c in ctxs:
            c.wait()
        torch.cuda.synchronize()
        return [f.cpu().numpy() for f in fulls]

    a, b = run(False), run(True)
    for k in range(steps):
        assert np.array_equal(a[k], b[k]), k
    assert a[steps - 1][..., :3].mean() > 0.01
    for c in ctxs:
        c.destroy()


def test_device_bvh_builder(pkg, make_hip, make_oracle):
    """SURVEY §8 f2 on the GPU: builder=device (Morton sort with rocPRIM, Karras hierarchy, device fit) gives a valid
    tree and the same hits / image as the host SAH build and the oracle; a later same-count set_mesh refits it."""
    from test_bvh import _device_vs_host, _check_tree
    scene = pkg.scenes.terrain(n=96, width=480, height_px=270, lights=False)
    scene.add_area_light_quad((0.0, -1.0, 0.0), (0.0, 30.0, 0.0), 6.0, 6.0, (400.0, 380.0, 350.0))
    scene.add_point_light((10.0, 20.0, -10.0), (900.0, 900.0, 800.0))
    _device_vs_host(pkg, make_hip, make_oracle, scene, 480, 270)
    # build time and traversal cost at the bench size, for DESIGN.md
    import time
    big = pkg.scenes.terrain(n=708, width=960, height_px=540)
    res = {}
    for builder in ("host", "device"):
        c = make_hip()
        c.set_setting("builder", builder)
        c.init(960, 540)
        t0 = time.perf_counter()
        big.upload(c)
        res[builder + "_upload_s"] = round(time.perf_counter() - t0, 4)
        c.set_setting("integrator", "pt")
        c.set_setting("spp", 8)
        c.render_frame(big.camera, pkg.RESET)
        t0 = time.perf_counter()
        for _ in range(3):
            c.render_frame(big.camera, pkg.CONVERGE)
        res[builder + "_ms_per_frame"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
        res[builder + "_mean"] = float(c.framebuffer()[..., :3].mean())
    print("DEVICE_BUILDER", res)
    assert abs(res["host_mean"] - res["device_mean"]) <= 2e-3 * res["host_mean"]


def test_blue_noise_primary_sampler(pkg, make_hip, make_oracle):
    """The blue-noise primary sampler on the GPU (synthetic table of the reference's layout) against the oracle."""
    table = pkg.scenes.synthetic_blue_noise()
    scene = pkg.scenes.cornell(480, 270, geometric_emitter=True)
    scene.camera.aperture = 0.05
    out = []
    for ctx in (make_hip(), make_oracle()):
        ctx.init(480, 270)
        scene.upload(ctx)
        ctx.set_blue_noise(table)
        for k, v in {"integrator": "pt", "spp": 8, "sampler": "bluenoise"}.items():
            ctx.set_setting(k, v)
        ctx.render_frame(scene.camera, pkg.RESET)
        out.append(ctx)
    hip, ref = out
    frac, rmse, _ = image_stats(hip.framebuffer(), ref.framebuffer(), 3e-2)
    assert frac <= 2e-2, (frac, rmse)
    m = ref.framebuffer()[..., :3].mean()
    assert abs(hip.framebuffer()[..., :3].mean() - m) <= 5e-3 * m


def test_skinned_tube_refit_on_device(pkg, make_hip, make_oracle):
    """BASELINE config 5 logic on the GPU: host skinning -> set_mesh with unchanged counts -> device refit; every
    frame's image equals a fresh build of that pose and the oracle's."""
    w, h = 480, 270
    base = pkg.scenes.skinned_tube(frame=0.0, rings=80, seg=48, width=w, height=h)
    live = make_hip()
    live.init(w, h)
    base.upload(live)
    live.set_setting("jitter", "center")
    live.set_setting("stage_timing", 1)
    for frame in (1.0, 2.5, 4.0):
        pose = pkg.scenes.skinned_tube(frame=frame, rings=80, seg=48, width=w, height=h)
        m = pose.meshes[0]
        live.set_mesh(0, m["vertices"], m["triangles"], m["indices"])
        live.update()
        live.render_frame(pose.camera, pkg.RESET)
        fresh, ref = _pair(pkg, make_hip, make_oracle, pose, w, h, {"integrator": "parity", "jitter": "center"})
        assert image_stats(live.framebuffer(), fresh.framebuffer(), 1e-4)[0] <= 1e-3
        assert image_stats(live.framebuffer(), ref.framebuffer(), 1e-3)[0] <= 2e-3
    ms, launches = live.get_kernel_time("refit")
    assert launches == 9 and ms > 0.0  # per refit: triangles, BVH2 boxes bottom-up, 4-wide node refresh


def test_two_ranks_on_one_device(pkg, make_hip):
    """The multi-GPU data path on one GPU: two contexts own the interleaved strips of one frame; their local
    framebuffers, concatenated the way the RCCL gather delivers them, de-interleave to the single-rank image."""
    import torch
    w, h = 640, 360
    scene = pkg.scenes.cornell(w, h, geometric_emitter=True)
    def render(rank, world):
        c = make_hip(rank, world)
        c.init(w, h)
        scene.upload(c)
        c.set_setting("integrator", "pt")
        c.set_setting("spp", 4)
        c.render_frame(scene.camera, pkg.RESET)
        return c
    single = render(0, 1)
    ranks = [render(r, 3) for r in range(3)]
    rows = ranks[0].local_rows()
    gathered = torch.empty((3, rows, w, 4), dtype=torch.float32, device="cuda:0")
    for r, c in enumerate(ranks):
        c.read_local_framebuffer_device(gathered[r].data_ptr())
    full = torch.empty((h, w, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    ranks[0].deinterleave_device(gathered.data_ptr(), full.data_ptr())
    assert np.array_equal(full.cpu().numpy(), single.framebuffer())


# ---- GPU-tier coverage of what round 1 checked on the emulation only ----------------------------------------------------------
def test_unrendered_remainder_pixels_on_the_gpu(pkg, make_hip, make_oracle):
    """EmbreeRT renders whole 4x2 packets only (Context.cpp:137-139); at 70 x 51 the columns beyond 68 and the
    last row keep their zeros, on the GPU as in the oracle."""
    scene = pkg.scenes.cornell(70, 51)
    for jitter in ("center", "xor128"):
        hip, ref = _pair(pkg, make_hip, make_oracle, scene, 70, 51, {"integrator": "parity", "jitter": jitter})
        a, b = hip.framebuffer(), ref.framebuffer()
        for img in (a, b):
            assert np.all(img[:, 68:] == 0) and np.all(img[50:, :] == 0)
            assert img[:50, :68, :3].max() > 0
        frac, rmse, _ = image_stats(a, b, 1e-3)
        assert frac <= 1e-3, (jitter, frac, rmse)


def test_parity_integrator_lens_sampling_on_the_gpu(pkg, make_hip, make_oracle):
    """aperture != 0 in the parity integrator (scalar form of the lens sample, Ray.cpp:16-47): 32 draws per
    packet instead of 16, origin on the 9-blade aperture."""
    scene = pkg.scenes.cornell(128, 96)
    scene.camera.aperture = 0.05
    hip, ref = _pair(pkg, make_hip, make_oracle, scene, 128, 96, {"integrator": "parity", "spp": 2})
    frac, rmse, _ = image_stats(hip.framebuffer(), ref.framebuffer(), 1e-3)
    assert frac <= 2e-3 and rmse <= 2e-3, (frac, rmse)


def test_spot_and_directional_lights_on_the_gpu(pkg, make_hip, make_oracle):
    """spot + directional lights in the path tracer (lights.h:48-76, :244-265)."""
    scene = pkg.scenes.cornell(256, 192, geometric_emitter=True)
    scene.add_spot_light((0.0, 9.0, 0.0), 20.0, (80.0, 80.0, 70.0), 35.0, (0.1, -1.0, 0.2))
    scene.add_directional_light((0.3, -1.0, 0.6), (1.5, 1.4, 1.2))
    hip, ref = _pair(pkg, make_hip, make_oracle, scene, 256, 192, {"integrator": "pt", "spp": 16})
    a, b = hip.framebuffer(), ref.framebuffer()
    frac, rmse, _ = image_stats(a, b, 2e-2)
    assert frac <= 1e-2 and rmse <= 3e-2, (frac, rmse)
    # the lights matter: without them the image is visibly darker
    plain = pkg.scenes.cornell(256, 192, geometric_emitter=True)
    hip2 = _pair(pkg, make_hip, make_oracle, plain, 256, 192, {"integrator": "pt", "spp": 16})[0]
    assert a[..., :3].mean() > 1.02 * hip2.framebuffer()[..., :3].mean() or a[..., :3].mean() < 0.98 * hip2.framebuffer()[..., :3].mean()


def test_more_lights_than_the_potential_cache_holds_on_the_gpu(pkg, make_hip, make_oracle):
    """26 lights of all four kinds: the loops over the lights read their records by scalar loads (rt::uniform_record), the first 16
    potentials wait in LDS for the selection pass, the others are recomputed there."""
    from test_emu_parity import many_lights_scene
    scene = many_lights_scene(pkg, 256, 192)
    hip, ref = _pair(pkg, make_hip, make_oracle, scene, 256, 192, {"integrator": "pt", "spp": 16})
    a, b = hip.framebuffer(), ref.framebuffer()
    frac, rmse, _ = image_stats(a, b, 2e-2)
    assert frac <= 1e-2 and rmse <= 3e-2, (frac, rmse)
    assert abs(a[..., :3].mean() - b[..., :3].mean()) <= 3e-3 * b[..., :3].mean()
    sa, sb = hip.get_stats(), ref.get_stats()
    assert abs(sa.shadowCount - sb.shadowCount) <= 2e-3 * sb.shadowCount, (sa.shadowCount, sb.shadowCount)


@pytest.mark.parametrize("scene_name", ["cornell", "cards"])
def test_per_depth_wave_counts_equal_the_oracle(pkg, make_hip, make_oracle, scene_name):
    """Wave sizes per depth (Kernels.cu:640,747,788: the compaction counters): extension rays of depth 1, of depths >= 2, and
    connections actually traced, 1 spp, depth 3 — equal to the oracle's up to the paths decided in the last bit."""
    scene = pkg.scenes.cornell(256, 192, geometric_emitter=True) if scene_name == "cornell" else pkg.scenes.cards(256, 192)
    hip, ref = _pair(pkg, make_hip, make_oracle, scene, 256, 192, {"integrator": "pt", "spp": 1, "max_depth": 3, "streams": 1})
    sa, sb = hip.get_stats(), ref.get_stats()
    for name in ("primaryCount", "secondaryCount", "deepCount", "shadowCount"):
        a, b = getattr(sa, name), getattr(sb, name)
        assert abs(a - b) <= max(3, 2e-4 * b), (name, a, b)
    assert sa.secondaryCount > 0 and sa.deepCount > 0 and sa.shadowCount > 0


def test_bench_workload_path_traced_image_vs_oracle(pkg, make_hip, make_oracle):
    """The bench workload itself — the 1 002 528-triangle terrain, HDR sky, 8 emissive light triangles, 2 point lights —
    with integrator=pt on both sides at 480 x 270 x 8 spp.

    The scene is +-50 units wide and the reference's geometric epsilon is 1e-5 (Kernels.cu:750, tools.h:119-123) — the size of
    an ulp of the hit points — so whether a shadow or bounce ray leaving a bumpy surface re-hits a neighbouring triangle is
    decided in the last bit, and sin/cos/rcp differ in the last bit between libm and the GPU.  Stated tolerance: >= 97 % of
    the pixels agree to 1e-3 (every decision of all 8 paths fell the same way), the rest are pixels where a path flipped
    (at most 1.5 % beyond 3e-2); nothing is biased: image mean within 1e-3, 8x8-block means within 0.5 % on average, and
    the per-depth wave sizes within 1.5e-3: about 2 % of the million paths re-decide somewhere (the flipped pixels above), each
    a coin toss for the depth-2 count, so two renders differ there by a standard deviation of ~30 of 95 000 = 3e-4 — the bound
    is five of them (2e-4 had held by luck: one change of operation order in the triangle test and the count moved by 41)."""
    scene = pkg.scenes.terrain(n=708, width=480, height_px=270)
    hip, ref = _pair(pkg, make_hip, make_oracle, scene, 480, 270, {"integrator": "pt", "spp": 8, "max_depth": 2})
    a, b = hip.framebuffer(), ref.framebuffer()
    assert np.isfinite(a).all()
    frac3, rmse, d = image_stats(a, b, 3e-2)
    assert (d > 1e-3).mean() <= 3e-2 and frac3 <= 1.5e-2 and rmse <= 8e-2, ((d > 1e-3).mean(), frac3, rmse)
    assert abs(a[..., :3].mean() - b[..., :3].mean()) <= 1e-3 * b[..., :3].mean()
    blk = lambda x: x[:264, :480, :3].astype(np.float64).reshape(33, 8, 60, 8, 3).mean((1, 3))  # noqa: E731
    ba, bb = blk(a), blk(b)
    rel = np.abs(ba - bb).max(-1) / np.maximum(bb.mean(-1), 1e-3)
    assert rel.mean() <= 5e-3 and rel.max() <= 0.12, (rel.mean(), rel.max())
    sa, sb = hip.get_stats(), ref.get_stats()
    for name in ("primaryCount", "secondaryCount", "deepCount", "shadowCount"):
        x, y = getattr(sa, name), getattr(sb, name)
        assert abs(x - y) <= 1.5e-3 * y, (name, x, y)
    assert sa.primaryCount == sb.primaryCount


def test_bench_workload_vs_the_reference_shaped_oracle(pkg, make_hip, make_oracle):
    """Round 4 gave the triangle test, the pt primary ray and the sky lookup a FIXED arithmetic shape and a total order on
    (t, prim), in the product and — so that the two agree to the last bit — in the oracle.  HIP against that oracle then no longer says
    how far those shapes are from the reference's own text (plain products and sums, contraction left to the compiler; strict
    `t > tt`, first triangle reached wins: bvh_tree.cpp:166-196, Kernels.cu:383-426,593-610).  The oracle keeps that form behind
    `arith=reference`; here the product meets it on the bench workload under ROUND 3's bounds, which were set before the shapes
    were fixed: the pixels of the existing test, and primary hits — same triangle on >= 99.95 % of the pixels, t within 1e-4
    relative, barycentrics within 5e-3."""
    scene = pkg.scenes.terrain(n=708, width=480, height_px=270)
    hip, ref = make_hip(), make_oracle()
    try:
        ref.set_setting("arith", "reference")
        for ctx in (hip, ref):
            ctx.init(480, 270)
            scene.upload(ctx)
            for k, v in {"integrator": "pt", "spp": 1, "max_depth": 2}.items():
                ctx.set_setting(k, v)
            ctx.render_frame(scene.camera, pkg.RESET)
        ha, hb = hip.primary_hits(), ref.primary_hits()  # (one sample per pixel: the record of THAT sample on both sides)
        for ctx in (hip, ref):
            ctx.set_setting("spp", 8)
            ctx.render_frame(scene.camera, pkg.RESET)
        a, b = hip.framebuffer(), ref.framebuffer()
        frac3, rmse, d = image_stats(a, b, 3e-2)
        assert (d > 1e-3).mean() <= 3e-2 and frac3 <= 1.5e-2 and rmse <= 8e-2, ((d > 1e-3).mean(), frac3, rmse)
        assert abs(a[..., :3].mean() - b[..., :3].mean()) <= 1e-3 * b[..., :3].mean()
        assert (ha["prim"] != hb["prim"]).mean() <= 5e-4 and (ha["inst"] != hb["inst"]).mean() <= 5e-4
        same = (ha["prim"] == hb["prim"]) & (ha["prim"] >= 0)
        assert (np.abs(ha["t"][same] - hb["t"][same]) <= 1e-4 * hb["t"][same]).all()
        for k in ("u", "v"):
            assert np.abs(ha[k][same] - hb[k][same]).max() <= 5e-3
        sa, sb = hip.get_stats(), ref.get_stats()
        for name in ("secondaryCount", "deepCount", "shadowCount"):
            x, y = getattr(sa, name), getattr(sb, name)
            assert abs(x - y) <= 1.5e-3 * y, (name, x, y)
    finally:
        ref.set_setting("arith", "product")  # (process-wide in the oracle)


def test_atrium_vs_the_reference_shaped_oracle(pkg, make_hip, make_oracle):
    """The same meeting on BASELINE config 4's scene: textures, 46 instances, and — what the terrain cannot show — the WORLD TREE's
    "M p instead of M^-1 o" measured against the reference-shaped TWO-LEVEL walk (plain products, strict `t > tt`, object-space
    triangle test: `arith=reference`), not only against the product-shaped one.  Bounds of the terrain's test; primary hits: the same
    triangle of the same instance on >= 99.9 % of the pixels (silhouettes of 45 transformed instances)."""
    scene = pkg.scenes.atrium(480, 270)
    hip, ref = make_hip(), make_oracle()
    try:
        ref.set_setting("arith", "reference")
        for ctx in (hip, ref):
            ctx.init(480, 270)
            scene.upload(ctx)
            for k, v in {"integrator": "pt", "spp": 1, "max_depth": 2}.items():
                ctx.set_setting(k, v)
            ctx.render_frame(scene.camera, pkg.RESET)
        ha, hb = hip.primary_hits(), ref.primary_hits()
        for ctx in (hip, ref):
            ctx.set_setting("spp", 8)
            ctx.render_frame(scene.camera, pkg.RESET)
        a, b = hip.framebuffer(), ref.framebuffer()
        frac3, rmse, d = image_stats(a, b, 3e-2)
        assert frac3 <= 2e-2 and rmse <= 8e-2, ((d > 1e-3).mean(), frac3, rmse)
        assert abs(a[..., :3].mean() - b[..., :3].mean()) <= 5e-3 * b[..., :3].mean()
        other = (ha["prim"] != hb["prim"]) | (ha["inst"] != hb["inst"])
        assert other.mean() <= 1e-3, other.mean()
        same = ~other & (ha["prim"] >= 0)
        assert (np.abs(ha["t"][same] - hb["t"][same]) <= 1e-4 * hb["t"][same]).all()
        for k in ("u", "v"):
            assert np.abs(ha[k][same] - hb[k][same]).max() <= 5e-3
        sa, sb = hip.get_stats(), ref.get_stats()
        for name in ("secondaryCount", "deepCount", "shadowCount"):
            x, y = getattr(sa, name), getattr(sb, name)
            assert abs(x - y) <= 3e-3 * y, (name, x, y)
    finally:
        ref.set_setting("arith", "product")  # (process-wide in the oracle)


def test_bench_workload_flips_isolated_gpu_arithmetic_vs_restatement(pkg, make_hip, make_emu, make_oracle):
    """Where do the per-pixel differences of the bench workload come from?  Three renders of the same 480 x 270 x 8 spp frame:
    HIP (the kernels), the host-emulation build of the SAME sources (same code, libm sin/cos/1/x instead of v_sin / v_cos /
    v_rcp: tests/emu) and the oracle (the independent C restatement, also libm).
      emulation vs HIP     = what the GPU's transcendental / reciprocal arithmetic alone flips,
      emulation vs oracle  = what differs between the two restatements on identical libm arithmetic.
    Both are asserted; the second must not be larger than the first by more than the noise of such a count — were the
    restatements to disagree in substance it would show here, without GPU arithmetic to hide behind."""
    scene = pkg.scenes.terrain(n=708, width=480, height_px=270)
    settings = {"integrator": "pt", "spp": 8, "max_depth": 2}
    imgs = {}
    for name, ctx in (("hip", make_hip()), ("emu", make_emu()), ("oracle", make_oracle())):
        ctx.init(480, 270)
        scene.upload(ctx)
        for k, v in settings.items():
            ctx.set_setting(k, v)
        ctx.render_frame(scene.camera, pkg.RESET)
        imgs[name] = ctx.framebuffer()
        ctx.destroy()
    def frac(a, b, tol):
        return float((image_stats(imgs[a], imgs[b], tol)[2] > tol).mean())
    gpu_arith = frac("emu", "hip", 1e-3)
    restate = frac("emu", "oracle", 1e-3)
    total = frac("hip", "oracle", 1e-3)
    print("flipped pixels (> 1e-3): emulation vs HIP %.4f, emulation vs oracle %.4f, HIP vs oracle %.4f" % (gpu_arith, restate, total))
    assert gpu_arith <= 3e-2, gpu_arith
    assert restate <= 1e-3, restate  # (measured: 0 of 129 600 pixels — the two restatements agree to the bit on libm)
    assert total <= 3e-2, total
    # none of the three pairs is biased
    for a, b in (("emu", "hip"), ("emu", "oracle")):
        ma, mb = imgs[a][..., :3].mean(), imgs[b][..., :3].mean()
        assert abs(ma - mb) <= 1e-3 * mb, (a, b, ma, mb)


@pytest.mark.parametrize("integrator", ["pt", "parity"])
def test_image_is_independent_of_how_calls_are_scheduled_gpu(pkg, make_hip, integrator):
    """The same on the real streams: ring of 1 / 2 / 4 buffer sets with eight calls in flight, calls cut into sub-batches,
    connection waves on a side stream, a wait every third call — bit-identical images."""
    from test_emu_parity import _pipelined
    scene = pkg.scenes.cornell(480, 272)
    base = {"integrator": integrator, "spp": 4, "max_depth": 2}
    ref = _pipelined(pkg, make_hip(), scene, 480, 272, dict(base, ring=1, streams=1), 8, 1)
    for extra, wait_every in (({"ring": 2}, 0), ({"ring": 4}, 0), ({"ring": 4}, 3), ({"ring": 4, "overlap": 1}, 0),
                              ({"streams": 4, "sub_batch_paths": 1}, 0), ({"streams": 3, "sub_batch_paths": 1, "overlap": 1}, 2),
                              ({"sample_group": 1}, 0), ({"sample_group": 2, "ring": 2}, 0), ({"sample_group": 64}, 1),
                              ({"sample_group": 4, "streams": 2, "sub_batch_paths": 1}, 0),
                              # round 4: one launch per depth or two;
                              # the primary wave per lane instead of as a packet
                              ({"fuse": 0}, 0), ({"fuse": 0, "ring": 4, "overlap": 1}, 0), ({"fuse": 0, "ring": 2}, 2),
                              ({"refill": 7, "sample_group": 64}, 0), ({"refill": 0}, 0)):
        img = _pipelined(pkg, make_hip(), scene, 480, 272, dict(base, **extra), 8, wait_every)
        assert np.array_equal(img, ref), (extra, wait_every)


@pytest.mark.parametrize("name,jitter", [("cornell96x64_center", "center"), ("cornell96x64_xor128", "xor128"), ("cards96x64_center", "center"), ("lens96x64_xor128", "xor128")])
def test_hip_matches_the_numpy_goldens_of_the_parity_integrator(pkg, make_hip, name, jitter):
    """The committed vectors of tests/golden/make_golden.py (independent numpy renderer) on the HIP kernels: primary hits,
    image, and — cards — retrieve_material's texture lookup incl. the FLOAT4 -> UINT fall-through."""
    import os
    import test_golden as TG
    g = np.load(os.path.join(TG.GOLD, name + ".npz"))
    img, hits = TG._render(pkg, make_hip(), jitter, name=name)
    TG._check(img, hits, g, name.startswith("cards"))


def test_batch_size_changes_between_pipelined_calls_gpu(pkg, make_hip):
    """spp changes while calls are in flight (the ring is re-laid out behind a synchronisation), 40 calls without a wait:
    the same 100 samples as one sample per call, up to the summation order of a batch."""
    scene = pkg.scenes.cornell(480, 272)
    a = make_hip()
    a.init(480, 272)
    scene.upload(a)
    a.set_setting("integrator", "pt")
    seq = [2, 4, 2, 1, 1, 8, 1, 1] * 5
    assert sum(seq) == 100
    for k, spp in enumerate(seq):
        a.set_setting("spp", spp)
        a.render_async(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
    a.wait()
    b = make_hip()
    b.init(480, 272)
    scene.upload(b)
    b.set_setting("integrator", "pt")
    b.set_setting("spp", 1)
    for k in range(100):
        b.render_async(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
    b.wait()
    ia, ib = a.framebuffer(), b.framebuffer()
    assert np.abs(ia - ib).max() <= 2e-5 * max(1.0, float(ib.max()))
    assert a.get_stats().primaryCount == 480 * 272


def test_random_barycentrics_closed_form_on_the_gpu(pkg, make_hip, make_oracle):
    """The integer closed form of lights.h:119-157 (rt_core.h: random_barycentrics) on the device against the oracle's loop:
    bit for bit (see tests/test_oracle_kat.py for the CPU tier)."""
    rng = np.random.default_rng(11)
    r0 = np.concatenate([rng.random(100000, dtype=np.float32), np.float32([0.0, 1.0, 0.5, 0.99999994, 2.3283064e-10])])
    rec = np.zeros((len(r0), 24), np.float32)
    rec[:, 20] = r0
    a = make_hip().kat("random_barycentrics", rec)[:, :3]
    b = make_oracle().kat("random_barycentrics", rec)[:, :3]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("scene_name", ["atrium", "terrain"])
def test_flat_instances_leave_every_result_alone_on_the_gpu(pkg, make_hip, scene_name):
    """Identity instances linked into the top-level tree directly (rfwhip_update, "flat" instances) against the two-level walk
    of every instance, on the device: atrium = an identity room beside 45 transformed instances of shared meshes, terrain = the
    bench scene's two identity instances.  Image, primary hits (instance ids included) and wave sizes are bit-equal; a mesh
    refit in between (it rewrites the triangle records that carry the instance index) changes nothing either.  Then the world
    tree (below)."""
    w, h = 480, 270
    scene = pkg.scenes.atrium(w, h) if scene_name == "atrium" else pkg.scenes.terrain(n=200, width=w, height_px=h)
    out = []
    for flat, flatten in ((1, 0), (0, 0), (1, 1 << 30)):  # (flatten_bytes = 0: no world tree, identity instances linked only)
        c = make_hip()
        c.init(w, h)
        c.set_setting("flat_instances", flat)
        c.set_setting("flatten_bytes", flatten)
        scene.upload(c)
        for k, v in {"integrator": "pt", "spp": 8, "max_depth": 3}.items():
            c.set_setting(k, v)
        c.render_frame(scene.camera, pkg.RESET)
        st = c.get_stats()
        out.append((c.framebuffer(), c.primary_hits(), (st.primaryCount, st.secondaryCount, st.deepCount, st.shadowCount)))
        if flat and not flatten:  # same vertices again = a refit of a flat instance's mesh (unchanged counts): the image must not move
            uses = [sum(1 for i in scene.instances if i["mesh"] == k) for k in range(len(scene.meshes))]
            mi = next(i["mesh"] for i in scene.instances if uses[i["mesh"]] == 1 and np.array_equal(i["transform"], np.eye(4)))
            m = scene.meshes[mi]
            c.set_mesh(mi, m["vertices"], m["triangles"], m["indices"])
            c.update()
            c.render_frame(scene.camera, pkg.RESET)
            assert np.array_equal(c.framebuffer(), out[0][0])
            assert np.array_equal(c.primary_hits()["inst"], out[0][1]["inst"])
    assert np.array_equal(out[0][0], out[1][0])
    for k in ("inst", "prim", "t"):
        assert np.array_equal(out[0][1][k], out[1][1][k]), k
    assert out[0][2] == out[1][2]
    insts = set(np.unique(out[0][1]["inst"])) - {-1}
    assert len(insts) >= 2, insts
    # The world tree (round 5, the default): static instances written out in world space under one tree.  The terrain's two
    # identity instances are linked as before (bit-equal); the atrium's 45 transformed columns are tested as M p instead of through
    # M^-1 o: the same triangle of the same instance on all but a handful of silhouette pixels, t to rounding, the image to the
    # statistics of a path tracer whose decisions see last-bit differences.
    wt, two = out[2], out[1]
    if scene_name != "atrium":
        assert np.array_equal(wt[0], two[0]) and wt[2] == two[2]
    else:
        same = (wt[1]["inst"] == two[1]["inst"]) & (wt[1]["prim"] == two[1]["prim"])
        assert (~same).mean() <= 2e-4, (~same).sum()
        hit = same & (wt[1]["prim"] >= 0)
        rel = np.abs(wt[1]["t"][hit] - two[1]["t"][hit]) / two[1]["t"][hit]
        print("world tree vs two-level walk: %d of %d primary hits on another triangle, t within %.2e relative" % ((~same).sum(), same.size, rel.max()))
        # (grazing hits are ill-conditioned: a last-bit difference of the vertices moves their t by far more than a last bit)
        assert np.quantile(rel, 0.999) <= 5e-6 and rel.max() <= 1e-3, (np.quantile(rel, 0.999), rel.max())
        frac3, rmse, d = image_stats(wt[0], two[0], 3e-2)
        # (most pixels of this scene see a transformed instance, at depth 3: 6 % of them hold a path that decided differently
        # somewhere — a texel, a light, a survival test; 0.3 % moved by more than 3e-2; nothing is biased, below)
        print("world tree vs two-level walk: %.4f of the pixels beyond 1e-3, %.4f beyond 3e-2" % ((d > 1e-3).mean(), frac3))
        assert (d > 1e-3).mean() <= 8e-2 and frac3 <= 1e-2, ((d > 1e-3).mean(), frac3)
        assert abs(wt[0][..., :3].mean() - two[0][..., :3].mean()) <= 2e-3 * two[0][..., :3].mean()
        for x, y in zip(wt[2], two[2]):
            assert abs(x - y) <= 2e-3 * max(y, 1), (wt[2], two[2])


def test_skipped_depth0_connections_leave_no_stale_terms_on_the_gpu(pkg, make_hip, make_oracle):
    """The depth-0 connection wave stores the first term of a slot's connection sum and the shade kernel zeroes only the slots
    that emit no shadow ray; when the reference's rule skips that wave (no path reaches depth 1), the device kernels still zero
    the slots left to it — at a size where the persistent-lane kernels run (test_emu_parity.py has the scene)."""
    from test_emu_parity import _gated_depth0_connections
    _gated_depth0_connections(pkg, make_hip, make_oracle, 480, 272)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["cornell_instances", "terrain", "axis_camera_mixed_signs", "lens", "scaled_atrium"])
def test_packet_form_of_the_primary_wave_on_the_gpu(pkg, make_hip, make_oracle, case):
    """k_primary_packet (wave-uniform traversal: scalar node fetches, one stack per wave) against the per-lane kernels on the
    MI355X: primary hit records, wave counts and images bit-equal for sample groups of 1, 8 and 64; hits against the oracle."""
    from test_emu_parity import packet_form_of_the_primary_wave
    packet_form_of_the_primary_wave(pkg, make_hip, make_oracle, case, 480, 270)
