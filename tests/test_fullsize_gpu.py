"""BASELINE config 3 at its full size on the GPU (1 002 536 triangles, 1920x1080, pt integrator depth 2), checked through
properties that do not need a full-size oracle render:
  * traversal: 300 000 random rays through the 1 M-triangle BVH against the oracle's own BVH (closest hit id, t, u, v);
  * accumulation: two CONVERGE batches of 8 spp == one batch of 16 spp (the sample index keys the RNG, the resolve
    sums a pixel's samples in a fixed order per batch) up to float summation order;
  * strips: 8 ranks' local framebuffers, gathered and de-interleaved, == the single-rank image, bit for bit;
  * sub-batch streams: the image does not depend on how a batch is cut into concurrent sub-batches;
  * energy: with the lights off the image is linear in the sky radiance, the firefly clamp bounds every sample, a black
    sky gives a black image.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 1920, 1080


@pytest.fixture(scope="module")
def terrain(pkg):
    return pkg.scenes.terrain(n=708, width=W, height_px=H)


def _ctx(pkg, make_hip, scene, rank=0, world=1, **settings):
    c = make_hip(rank, world)
    c.init(W, H)
    scene.upload(c)
    c.set_setting("integrator", "pt")
    for k, v in settings.items():
        c.set_setting(k, v)
    return c


def test_traversal_1m_triangles(pkg, make_hip, make_oracle, terrain):
    from test_trace_rays import _rays
    core, ref = make_hip(), make_oracle()
    for c in (core, ref):
        c.init(64, 64)
        terrain.upload(c)
    rng = np.random.default_rng(5)
    o, d = _rays(rng, 300000, 48.0)
    a, b = core.trace_rays(o, d), ref.trace_rays(o, d)
    # Product and oracle state the triangle test in the same fixed shape (csrc/rt_core.h: rounded()) with the same total order on
    # (t, prim); the ONE operation that differs is the reciprocal of the determinant (v_rcp_f32, 1 ulp, against an IEEE division).
    # Measured on the MI355X: no ray of the 300 000 gets another triangle, 90 % of the hit records are bit-equal, the rest differ
    # by one unit in the last place of t, u or v (round 3 had to allow 5e-3 on u and v: the compilers had contracted the cross
    # products of the two implementations differently).
    assert (a["prim"] != b["prim"]).mean() <= 1e-5 and (a["inst"] != b["inst"]).mean() <= 1e-5
    same = (a["prim"] == b["prim"]) & (a["prim"] >= 0)
    assert same.sum() > 0.5 * len(same)
    assert (np.abs(a["t"][same] - b["t"][same]) <= 5e-7 * np.abs(b["t"][same])).all()
    assert np.abs(a["u"][same] - b["u"][same]).max() <= 3e-7 and np.abs(a["v"][same] - b["v"][same]).max() <= 3e-7
    assert (a["t"][same] == b["t"][same]).mean() >= 0.8


def test_accumulation_and_subbatch_independence(pkg, make_hip, terrain):
    a = _ctx(pkg, make_hip, terrain, spp=16, streams=4, sub_batch_paths=1000000)
    a.render_frame(terrain.camera, pkg.RESET)
    img16 = a.framebuffer()
    b = _ctx(pkg, make_hip, terrain, spp=8, streams=1)
    b.render_frame(terrain.camera, pkg.RESET)
    b.render_frame(terrain.camera, pkg.CONVERGE)
    img8x2 = b.framebuffer()
    assert np.abs(img16 - img8x2).max() <= 1e-4 * max(1.0, float(img16.max()))
    c = _ctx(pkg, make_hip, terrain, spp=16, streams=3, sub_batch_paths=1000000)
    c.render_frame(terrain.camera, pkg.RESET)
    assert np.array_equal(c.framebuffer(), img16)           # 16 spp cut 4/4/4/4 or 5/5/6: same samples, same order
    st = a.get_stats()
    assert st.primaryCount == W * H * 16
    assert st.secondaryCount > 0 and st.deepCount > 0 and st.shadowCount > 0


def test_eight_strip_ranks_equal_single_rank(pkg, make_hip, terrain):
    import torch
    world = 8
    single = _ctx(pkg, make_hip, terrain, spp=4)
    single.render_frame(terrain.camera, pkg.RESET)
    rows = None
    gathered = None
    root = None
    for r in range(world):
        c = _ctx(pkg, make_hip, terrain, rank=r, world=world, spp=4)
        c.render_frame(terrain.camera, pkg.RESET)
        if gathered is None:
            rows = c.local_rows()
            gathered = torch.empty((world, rows, W, 4), dtype=torch.float32, device="cuda:0")
        c.read_local_framebuffer_device(gathered[r].data_ptr())
        if r == 0:
            root = c
        else:
            c.destroy()
    full = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    root.deinterleave_device(gathered.data_ptr(), full.data_ptr())
    assert np.array_equal(full.cpu().numpy(), single.framebuffer())


def test_sky_linearity_and_clamp(pkg, make_hip):
    """With the lights off every path ends on the sky (or dies): the image is linear in the sky's radiance, because no
    discrete decision of a path depends on it.  (It is NOT linear in the lights' radiance: every light pdf in
    lights.h carries 1 / energy, so a light's estimate grows with radiance x energy — restated as is.)  With the clamp on, every
    contribution is bounded by clampIntensity (tools.h:184-192)."""
    def scene(scale, clamp):
        s = pkg.scenes.terrain(n=708, width=W, height_px=H, lights=False)
        pix, w, h = s.sky
        s.sky = (pix * np.float32(scale), w, h)
        s.camera.clampValue = clamp
        return s
    imgs = []
    for scale in (1.0, 2.0):
        s = scene(scale, 1e30)
        c = _ctx(pkg, make_hip, s, spp=4)
        c.render_frame(s.camera, pkg.RESET)
        imgs.append(c.framebuffer()[..., :3].astype(np.float64))
        st = c.get_stats()
        assert st.shadowCount == 0                                  # no lights, no connections
        c.destroy()
    lit = imgs[0] > 1e-3
    assert lit.mean() > 0.3
    assert np.abs(imgs[1][lit] / imgs[0][lit] - 2.0).max() <= 1e-4
    s = scene(1.0, 0.25)
    c = _ctx(pkg, make_hip, s, spp=2)
    c.render_frame(s.camera, pkg.RESET)
    assert c.framebuffer()[..., :3].max() <= 0.25 * 1.001           # one sky contribution per path, clamped
    s = scene(0.0, 10.0)
    c = _ctx(pkg, make_hip, s, spp=1)
    c.render_frame(s.camera, pkg.RESET)
    assert c.framebuffer()[..., :3].max() == 0.0


def test_queue_blocks_and_kernel_forms_do_not_change_the_image(pkg, make_hip, terrain):
    """At this size the shade kernel fills its output queues in blocks and leaves void entries behind the last ones; the
    persistent-lane kernels (refill=7) and the one-ray-per-lane kernels (refill=0) both have to skip them.  Same image and
    per-depth ray counts up to floating-point contraction, and the counts are rays, not queue slots."""
    a = _ctx(pkg, make_hip, terrain, spp=4, max_depth=3)
    a.render_frame(terrain.camera, pkg.RESET)
    b = _ctx(pkg, make_hip, terrain, spp=4, max_depth=3, refill=0)
    b.render_frame(terrain.camera, pkg.RESET)
    # (the two kernel forms inline the same traversal into different kernels: the compiler contracts a few multiply-adds
    # differently, a hit moves by an ulp here and there — 2 % of the pixels differ in some bit, a handful visibly)
    d = np.abs(a.framebuffer() - b.framebuffer())[..., :3].max(-1)
    assert (d > 1e-3).mean() <= 1e-4 and np.median(d) == 0.0
    sa, sb = a.get_stats(), b.get_stats()
    for k in ("primaryCount", "secondaryCount", "deepCount", "shadowCount"):
        assert abs(getattr(sa, k) - getattr(sb, k)) <= 1e-5 * getattr(sa, k), k
    assert sa.primaryCount == W * H * 4 and 0 < sa.deepCount < sa.secondaryCount < sa.primaryCount
    # one sample at a time: small launches reserve exactly what they emit (no void entries) — the same rays in total
    c = _ctx(pkg, make_hip, terrain, spp=1, max_depth=3)
    tot = dict(secondaryCount=0, deepCount=0, shadowCount=0)
    for k in range(4):
        c.render_frame(terrain.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
        st = c.get_stats()
        for name in tot:
            tot[name] += getattr(st, name)
    for name in tot:
        assert tot[name] == getattr(sa, name), name
    assert np.abs(c.framebuffer() - a.framebuffer()).max() <= 1e-4 * max(1.0, float(a.framebuffer().max()))


def test_finished_group_flags_do_not_change_the_image(pkg, make_hip, terrain):
    """The packet form of the primary wave flags the 64-slot groups it finishes itself (no hit: sky terms written) and the shade
    kernel's scan passes them by (`group_flags`).  A quarter of the terrain's groups are such.  Same image bit for bit, same ray
    counts, same primary hits — with sub-batches that begin anywhere in the buffers (streams, a strip rank) as well."""
    ref = None
    for settings in (dict(group_flags=0), dict(group_flags=1), dict(group_flags=1, streams=3, sub_batch_paths=5_000_000)):
        c = _ctx(pkg, make_hip, terrain, spp=16, max_depth=2, **settings)
        c.render_frame(terrain.camera, pkg.RESET)
        st = c.get_stats()
        out = (c.framebuffer().copy(), (st.primaryCount, st.secondaryCount, st.deepCount, st.shadowCount), c.primary_hits())
        if ref is None:
            ref = out
            continue
        assert np.array_equal(out[0], ref[0]) and out[1] == ref[1], settings
        for key in ref[2]:
            assert np.array_equal(out[2][key], ref[2][key]), key
    # frames in flight: calls enqueued without waiting take turns through the ring of buffer sets — every set has flag bytes of its own
    acc = []
    for flags in (0, 1):
        c = _ctx(pkg, make_hip, terrain, spp=8, max_depth=2, group_flags=flags)
        for k in range(7):
            c.render_async(terrain.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
        c.wait()
        acc.append(c.framebuffer().copy())
        c.destroy()
    assert np.array_equal(acc[0], acc[1])
    # a strip rank: its slices are other rows, its groups other pixels
    import torch
    imgs = []
    for flags in (0, 1):
        c = _ctx(pkg, make_hip, terrain, rank=3, world=8, spp=16, max_depth=2, group_flags=flags)
        c.render_frame(terrain.camera, pkg.RESET)
        local = torch.empty((c.local_rows(), W, 4), dtype=torch.float32, device="cuda:0")
        c.read_local_framebuffer_device(local.data_ptr())
        torch.cuda.synchronize()
        imgs.append(local.cpu().numpy())
        c.destroy()
    assert np.array_equal(imgs[0], imgs[1]) and imgs[0].size > 0 and float(imgs[0][..., :3].max()) > 0.0


def test_sample_groups_at_full_size(pkg, make_hip, terrain):
    """The slot layout's sample groups (a wave = 64 / g pixels x g samples, rt_core.h) at the bench's own size: 1920 x 1080 x
    32 spp on the 1 M-triangle terrain with g = 1 (a wave = one 8 x 8 tile of one sample), 8 and 32 — the images, the primary
    hits read back through the layout and the wave sizes are identical, bit for bit."""
    out = []
    for g in (1, 8, 32):
        c = _ctx(pkg, make_hip, terrain, spp=32, streams=1, sample_group=g)
        c.render_frame(terrain.camera, pkg.RESET)
        st = c.get_stats()
        out.append((c.framebuffer(), c.primary_hits(), (st.primaryCount, st.secondaryCount, st.deepCount, st.shadowCount)))
        c.destroy()
    for img, hits, counts in out[1:]:
        assert np.array_equal(img, out[0][0])
        for k in ("t", "prim", "inst", "u", "v"):
            assert np.array_equal(hits[k], out[0][1][k]), k
        assert counts == out[0][2]


# ---------------------------------------------------------------------------------------------------------------------------------
# BASELINE config 4 at its full size: the atrium (263 k triangles in 46 instances, textures, normal maps, the world tree) — the scene
# BASELINE.json shards over 8 GPUs — through the same size-independent properties (round 5's verdict: until now it was compared with
# the oracle at 480 x 270 only, and the strip / scheduling properties ran on the terrain)
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def atrium(pkg):
    return pkg.scenes.atrium(W, H)


def test_atrium_eight_strip_ranks_equal_single_rank(pkg, make_hip, atrium):
    """8 ranks' local framebuffers, gathered and de-interleaved, == the single-rank image, bit for bit: textured shading, the
    world tree and the instance records give the same pixel whichever rank renders it."""
    import torch
    world = 8
    single = _ctx(pkg, make_hip, atrium, spp=4)
    single.render_frame(atrium.camera, pkg.RESET)
    st = single.get_stats()
    assert st.primaryCount == W * H * 4 and st.secondaryCount > 0.5 * st.primaryCount and st.deepCount > 0.25 * st.primaryCount  # (a room)
    gathered = root = None
    for r in range(world):
        c = _ctx(pkg, make_hip, atrium, rank=r, world=world, spp=4)
        c.render_frame(atrium.camera, pkg.RESET)
        if gathered is None:
            gathered = torch.empty((world, c.local_rows(), W, 4), dtype=torch.float32, device="cuda:0")
        c.read_local_framebuffer_device(gathered[r].data_ptr())
        if r == 0:
            root = c
        else:
            c.destroy()
    full = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    root.deinterleave_device(gathered.data_ptr(), full.data_ptr())
    assert np.array_equal(full.cpu().numpy(), single.framebuffer())


def test_atrium_scheduling_independence(pkg, make_hip, atrium):
    """Sub-batch cut, sample-group size and accumulation over calls do not change a bit of the atrium's image (16 spp cut 4/4/4/4
    or 5/5/6, g = 1 / 8; 8 + 8 spp against 16 up to the summation order), nor its per-depth ray counts."""
    a = _ctx(pkg, make_hip, atrium, spp=16, streams=4, sub_batch_paths=1000000)
    a.render_frame(atrium.camera, pkg.RESET)
    img = a.framebuffer()
    sa = a.get_stats()
    a.destroy()
    for settings in (dict(streams=3, sub_batch_paths=1000000), dict(streams=1, sample_group=1), dict(streams=1, sample_group=8)):
        c = _ctx(pkg, make_hip, atrium, spp=16, **settings)
        c.render_frame(atrium.camera, pkg.RESET)
        assert np.array_equal(c.framebuffer(), img), settings
        sc = c.get_stats()
        assert (sc.primaryCount, sc.secondaryCount, sc.deepCount, sc.shadowCount) == (sa.primaryCount, sa.secondaryCount, sa.deepCount, sa.shadowCount)
        c.destroy()
    b = _ctx(pkg, make_hip, atrium, spp=8, streams=1)
    b.render_frame(atrium.camera, pkg.RESET)
    b.render_frame(atrium.camera, pkg.CONVERGE)
    assert np.abs(b.framebuffer() - img).max() <= 1e-4 * max(1.0, float(img.max()))


def test_atrium_world_tree_on_and_off(pkg, make_hip, atrium):
    """The world tree (static instances written out in world space) against the two-level walk at full size: the same triangle of
    the same instance for every primary ray but silhouettes (M p is tested instead of M^-1 o), t to rounding, image statistics
    unbiased — the tolerances of tests/test_emu_parity.py::test_flat_instances_leave_every_result_alone, at 1920 x 1080."""
    out = []
    for flatten in (1 << 30, 0):
        c = _ctx(pkg, make_hip, atrium, spp=4, flatten_bytes=flatten)
        c.update()  # (flatten_bytes takes effect with the next update)
        c.render_frame(atrium.camera, pkg.RESET)
        st = c.get_stats()
        out.append((c.framebuffer()[..., :3].astype(np.float64), c.primary_hits(), (st.primaryCount, st.secondaryCount, st.deepCount, st.shadowCount)))
        c.destroy()
    (ia, ha, ca), (ib, hb, cb) = out
    other = (ha["prim"] != hb["prim"]) | (ha["inst"] != hb["inst"])
    assert other.mean() <= 1e-4, other.mean()
    same = ~other & (ha["prim"] >= 0)
    assert same.mean() > 0.5
    assert (np.abs(ha["t"][same] - hb["t"][same]) <= 5e-6 * hb["t"][same]).mean() >= 0.999
    d = np.sqrt(((ia - ib) ** 2).sum(-1))
    assert (d > 1e-3).mean() <= 3e-2, (d > 1e-3).mean()          # (a path here and there decides differently)
    assert abs(ia.mean() - ib.mean()) <= 2e-3 * ib.mean()
    for x, y in zip(ca, cb):
        assert abs(x - y) <= 3e-3 * max(y, 1), (ca, cb)


def test_connection_packets_do_not_change_the_image(pkg, make_hip, terrain, atrium):
    """Round 6: the connection wave of the primary vertices in packet form (k_shadow_packet: shadow rays carry their light's bin in the
    top bits of their slot word, a wave sorts runs of 256 by it and walks the tree once per 64 rays; setting shadow_packets, on the
    sub-batch's connection stream with shadow_side).  Which rays share a packet changes no ray's answer: image, primary hits and
    per-depth ray counts are those of the per-lane connection wave, bit for bit, on the bench scene and on the atrium at full size;
    the default (-1) measures the light bins per sorted run and keeps the packets while they average at most 8."""
    for scene, name in ((terrain, "terrain"), (atrium, "atrium")):
        out = []
        for settings in (dict(shadow_packets=0), dict(shadow_packets=1, shadow_side=0), dict(shadow_packets=1, shadow_side=1), dict()):
            c = _ctx(pkg, make_hip, scene, spp=16, **settings)
            c.render_frame(scene.camera, pkg.RESET)
            c.render_frame(scene.camera, pkg.CONVERGE)
            st = c.get_stats()
            out.append((c.framebuffer(), (st.primaryCount, st.secondaryCount, st.deepCount, st.shadowCount), float(c.get_setting("shadow_bins_per_run")),
                        c.get_setting("shadow_packets_on")))
            c.destroy()
        for img, counts, _, _ in out[1:]:
            assert np.array_equal(img, out[0][0]), name
            assert counts == out[0][1], name
        assert out[0][2] == 0.0 and out[1][2] > 1.0          # (no packets, no runs sorted; with them: a few bins per run)
        # the default decides by the measured bins per run (<= 8 of the 16): on for both (terrain ~3.6, atrium ~5.5)
        assert out[3][3] == "1" and 1.0 < out[3][2] <= 8.0, (name, out[3][2], out[3][3])
    # a sub-batch of more than 2^27 path slots (128 spp in ONE sub-batch: 265 M) leaves 3 bits for the bin instead of 4: 8 bins
    big = []
    for sp in (0, 1):
        c = _ctx(pkg, make_hip, terrain, spp=128, streams=1, shadow_packets=sp)
        c.render_frame(terrain.camera, pkg.RESET)
        st = c.get_stats()
        big.append((c.framebuffer(), (st.primaryCount, st.secondaryCount, st.deepCount, st.shadowCount), float(c.get_setting("shadow_bins_per_run"))))
        c.destroy()
    assert np.array_equal(big[0][0], big[1][0]) and big[0][1] == big[1][1] and big[1][2] > 1.0
