"""Known answers that pin the oracle's integer / scalar building blocks.

The reference ships no tests or vectors; these values follow by hand from its integer arithmetic
(utils/xor128.h:20-27, bsdf/tools.h:218-235) or from the acceptance rules of its intersection code
(bvh_tree.cpp:166-196, aabb.cpp:39-77), and tests/golden/rng_kat.npz holds an independent Python re-derivation."""
import ctypes as C
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def f3(x):
    return (C.c_float * 3)(*[float(v) for v in x])


def test_xor128_default_seed(orc):
    L = orc.load()
    s = (C.c_uint32 * 4)(123456789, 362436069, 521288629, 88675123)
    got = [L.rfwo_xor128_next(s) for _ in range(6)]
    assert got == [3701687786, 458299110, 2500872618, 3633119408, 516391518, 2377269574]
    kat = np.load(os.path.join(GOLD, "rng_kat.npz"))
    s = (C.c_uint32 * 4)(123456789, 362436069, 521288629, 88675123)
    draws = [L.rfwo_xor128_next(s) for _ in range(1000)]
    assert draws[:8] == [int(x) for x in kat["xor128_first8"]]
    assert draws[999] == int(kat["xor128_draw_1000"])
    assert list(s) == [int(x) for x in kat["xor128_state_after_1000"]]


def test_rand_scale_and_inclusive_one(orc):
    L = orc.load()
    s = (C.c_uint32 * 4)(123456789, 362436069, 521288629, 88675123)
    r = L.rfwo_rng_rand(s)
    assert r == np.float32(np.float32(3701687786) * np.float32(2.3283064365387e-10))
    # uint 0xFFFFFFFF maps to exactly 1.0f (rng.h:14): the jitter range is [0, 1] inclusive
    assert np.float32(np.float32(0xFFFFFFFF) * np.float32(2.3283064365387e-10)) == np.float32(1.0)


@pytest.mark.parametrize("draws", [0, 1, 31, 32, 1000, 259200 * 32, (1 << 40) + 12345])
def test_xor128_jump_equals_stepping(orc, draws):
    L = orc.load()
    a = (C.c_uint32 * 4)(123456789, 362436069, 521288629, 88675123)
    b = (C.c_uint32 * 4)(123456789, 362436069, 521288629, 88675123)
    L.rfwo_xor128_jump(a, draws)
    if draws <= 10_000_000:
        for _ in range(draws):
            L.rfwo_xor128_next(b)
        assert list(a) == list(b)
    else:  # composition property for distances too long to step
        L.rfwo_xor128_jump(b, draws - 777)
        L.rfwo_xor128_jump(b, 777)
        assert list(a) == list(b)


def test_wang_hash_and_xorshift(orc):
    L = orc.load()
    assert [L.rfwo_wang_hash(x) for x in (0, 1, 16789)] == [3232319850, 663891101, 4005165182]
    s = C.c_uint32(1)
    assert [L.rfwo_random_int(C.byref(s)), L.rfwo_random_int(C.byref(s))] == [270369, 67634689]
    s = C.c_uint32(1)
    assert L.rfwo_random_float(C.byref(s)) == np.float32(np.float32(270369) * np.float32(2.3283064365387e-10))


def test_half_to_float_all_patterns(orc):
    L = orc.load()
    bits = np.arange(65536, dtype=np.uint16)
    want = bits.view(np.float16).astype(np.float32)
    got = np.array([L.rfwo_half_to_float(int(b)) for b in bits], np.float32)
    ok = np.isnan(want) == np.isnan(got)
    assert ok.all()
    m = ~np.isnan(want)
    assert np.array_equal(want[m], got[m])


def _tri(orc, org, d, tmin, t0, p0, p1, p2):
    L = orc.load()
    t, u, v = C.c_float(t0), C.c_float(), C.c_float()
    hit = L.rfwo_intersect_triangle(f3(org), f3(d), tmin, C.byref(t), f3(p0), f3(p1), f3(p2), C.byref(u), C.byref(v))
    return hit, t.value, u.value, v.value


def test_moller_trumbore_acceptance_rules(orc):
    p0, p1, p2 = (0, 0, 5), (2, 0, 5), (0, 2, 5)
    hit, t, u, v = _tri(orc, (0.5, 0.25, 0), (0, 0, 1), 1e-5, 1e34, p0, p1, p2)
    assert hit == 1 and t == 5.0 and u == 0.25 and v == 0.125  # u weights p1, v weights p2
    assert _tri(orc, (0.5, 0.25, 0), (0, 0, 1), 1e-5, 5.0, p0, p1, p2)[0] == 0      # t must be < current t
    assert _tri(orc, (0.5, 0.25, 0), (0, 0, 1), 5.0, 1e34, p0, p1, p2)[0] == 0      # t must be > t_min
    assert _tri(orc, (-0.01, 0.5, 0), (0, 0, 1), 1e-5, 1e34, p0, p1, p2)[0] == 0    # u < 0
    assert _tri(orc, (0.5, -0.01, 0), (0, 0, 1), 1e-5, 1e34, p0, p1, p2)[0] == 0    # v < 0
    assert _tri(orc, (1.2, 1.2, 0), (0, 0, 1), 1e-5, 1e34, p0, p1, p2)[0] == 0      # u + v > 1
    assert _tri(orc, (0.5, 0.25, 0), (1, 0, 0), 1e-5, 1e34, p0, p1, p2)[0] == 0     # parallel: |a| < 1e-6
    assert _tri(orc, (0.5, 0.25, 10), (0, 0, 1), 1e-5, 1e34, p0, p1, p2)[0] == 0    # behind the origin
    # back faces are hit as well (no culling)
    assert _tri(orc, (0.5, 0.25, 10), (0, 0, -1), 1e-5, 1e34, p0, p1, p2)[0] == 1
    # |a| rejection is absolute: a triangle scaled down until |a| < 1e-6 disappears (bvh_tree.cpp:172-174)
    s = 5e-4
    assert _tri(orc, (0.25 * s, 0.25 * s, 0), (0, 0, 1), 1e-5, 1e34, (0, 0, 5), (s, 0, 5), (0, s, 5))[0] == 0


def test_slab_rule(orc):
    L = orc.load()
    tmin, tmax = C.c_float(), C.c_float()
    inv = lambda d: [1.0 / x if x != 0 else float("inf") for x in d]
    box = (f3((-1, -1, 4)), f3((1, 1, 6)))
    assert L.rfwo_intersect_aabb(box[0], box[1], f3((0, 0, 0)), f3(inv((0, 0, 1))), 1e34, C.byref(tmin), C.byref(tmax)) == 1
    assert tmin.value == 4.0 and tmax.value == 6.0
    # tmin must be < t: a closer hit already found culls the box
    assert L.rfwo_intersect_aabb(box[0], box[1], f3((0, 0, 0)), f3(inv((0, 0, 1))), 3.9, C.byref(tmin), C.byref(tmax)) == 0
    assert L.rfwo_intersect_aabb(box[0], box[1], f3((3, 0, 0)), f3(inv((0, 0, 1))), 1e34, C.byref(tmin), C.byref(tmax)) == 0
    # degenerate (tmax == tmin) is a miss: hit iff tmax > tmin (aabb.cpp:76)
    flat = (f3((-1, -1, 5)), f3((1, 1, 5)))
    assert L.rfwo_intersect_aabb(flat[0], flat[1], f3((0, 0, 0)), f3(inv((0, 0, 1))), 1e34, C.byref(tmin), C.byref(tmax)) == 0


def test_heron_area_and_normal_packing(orc):
    L = orc.load()
    assert abs(L.rfwo_triangle_area(f3((0, 0, 0)), f3((3, 0, 0)), f3((0, 4, 0))) - 6.0) < 1e-6
    rng = np.random.default_rng(3)
    for _ in range(200):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        if n[2] < -0.9:  # the 16:16 packing is singular at -z and loses precision towards it (tools.h:14)
            continue
        out = (C.c_float * 3)()
        L.rfwo_unpack_normal(L.rfwo_pack_normal(f3(n)), out)
        assert np.abs(np.array(out[:]) - n).max() < 5e-4


def test_camera_view_matches_formula(pkg, orc, make_oracle):
    cam = pkg.Camera(aperture=0.0, FOV=40.0, focalDistance=5.0)
    cam.look_at((1.0, 2.0, -7.0), (0.5, 1.0, 3.0))
    cam.resize(640, 360)
    o = make_oracle()
    v = o.camera_view(cam)
    d = np.asarray(cam.direction, np.float64)
    right = np.cross(d, [0, 1, 0]); right /= np.linalg.norm(right)
    up = np.cross(right, d)
    s = np.tan(np.radians(20.0))
    c = np.asarray(cam.position) + 5.0 * d
    p1 = c - s * 5.0 * (640 / 360) * right + s * 5.0 * up
    p3 = c - s * 5.0 * (640 / 360) * right - s * 5.0 * up
    assert np.abs(np.array(v.p1[:]) - p1).max() < 1e-5
    assert np.abs(np.array(v.p3[:]) - p3).max() < 1e-5
    assert abs(v.spreadAngle - np.radians(40.0) / 360) < 1e-9
    assert v.aperture == 0.0


def test_bsdf_energy_and_pdf_sanity(orc):
    """Disney BSDF restatement: non-negative, finite, and the cosine-weighted diffuse lobe integrates to its albedo
    share (Monte-Carlo over the hemisphere, roughness 1, no specular)."""
    L = orc.load()
    params = (C.c_uint32 * 4)(0xFF000000, 0, 0x7F00FF00, 0)  # metallic 0, subsurface 0, specular 0, roughness 1
    rng = np.random.default_rng(5)
    wo = np.array([0.3, 0.2, 0.93]); wo /= np.linalg.norm(wo)
    acc, n = 0.0, 4000
    for _ in range(n):
        z = rng.random(); phi = 2 * np.pi * rng.random(); r = np.sqrt(1 - z * z)
        wi = np.array([r * np.cos(phi), r * np.sin(phi), z])
        rgb, pdf = (C.c_float * 3)(), C.c_float()
        L.rfwo_evaluate_bsdf(f3((0.5, 0.5, 0.5)), params, f3((0, 0, 1)), f3(wo), f3(wi), rgb, C.byref(pdf))
        assert all(np.isfinite(rgb[:])) and min(rgb[:]) >= 0 and pdf.value >= 0
        acc += rgb[0] * z * 2 * np.pi  # uniform hemisphere pdf = 1/2pi
    assert 0.3 < acc / n < 0.75  # albedo 0.5 diffuse + the unavoidable grazing Fresnel lobe


def test_random_barycentrics_closed_form_equals_the_loop_bit_for_bit(pkg, make_emu, make_oracle):
    """lights.h:119-157: the product computes the sixteen sub-triangle rounds in closed form on integers (rt_core.h:
    random_barycentrics), the oracle walks the reference's loop.  Every intermediate of the loop is exact in float, so the two
    must agree to the last bit: 200 000 random r0, the ends of the range, and every pattern of the first four rounds."""
    rng = np.random.default_rng(7)
    r0 = np.concatenate([rng.random(200000, dtype=np.float32), np.float32([0.0, 1.0, 0.5, 0.25, 0.75, 0.99999994, 1e-9, 2.3283064e-10]),
                         (np.arange(256, dtype=np.float32) + np.float32(0.37)) / np.float32(256.0)])
    rec = np.zeros((len(r0), 24), np.float32)
    rec[:, 20] = r0
    a = make_emu().kat("random_barycentrics", rec)[:, :3]
    b = make_oracle().kat("random_barycentrics", rec)[:, :3]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.all(a >= -1e-6) and np.all(np.abs(a.sum(1) - 1.0) < 1e-6)
