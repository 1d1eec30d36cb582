"""The strict-arithmetic validation build (csrc/rt_strict_math.h, build.py: build_strict): the product sources compiled with
-DRT_STRICT_MATH -ffp-contract=off — every transcendental function one shared float implementation, the triangle test's
reciprocal an IEEE division, no fma contraction — for the GPU and, the same way, for the host emulation.  The two must then
agree to the BIT: the 2 % of pixels in which the shipped kernels differ from the emulation on the bench scene are thereby
proven to be arithmetic mode (v_sin / v_cos / v_rcp, library functions, contraction), not a difference between two programs.
Neither library is the product: tests/_strict/ and tests/_emu/ are test infrastructure."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT, image_stats

STRICT_FLAGS = ("-DRT_STRICT_MATH", "-ffp-contract=off")


@pytest.fixture(scope="module")
def emu_strict_lib():
    import build_emu
    return ctypes.CDLL(build_emu.build(defines=STRICT_FLAGS, tag="_strict"))


def _render(pkg, ctx, scene, w, h, settings):
    ctx.init(w, h)
    scene.upload(ctx)
    for k, v in settings.items():
        ctx.set_setting(k, v)
    ctx.render_frame(scene.camera, pkg.RESET)
    st = ctx.get_stats()
    return ctx.framebuffer(), ctx.primary_hits(), (st.primaryCount, st.secondaryCount, st.deepCount, st.shadowCount)


def test_strict_emulation_against_the_oracle(pkg, emu_strict_lib, make_oracle):
    """CPU tier: the strict emulation is still the same path tracer — against the oracle (glibc) it differs only where the
    choice of math library flips a decision."""
    scene = pkg.scenes.cornell(96, 64, geometric_emitter=True)
    settings = {"integrator": "pt", "spp": 8, "max_depth": 2}
    a = _render(pkg, pkg._binding.CoreBinding(emu_strict_lib, "rfwhip_", 0, 0, 1), scene, 96, 64, settings)
    b = _render(pkg, make_oracle(), scene, 96, 64, settings)
    frac, rmse, _ = image_stats(a[0], b[0], 2e-2)
    assert frac <= 2e-2 and rmse <= 3e-2, (frac, rmse)
    settings["spp"] = 1  # (hit records are compared sample for sample)
    a = _render(pkg, pkg._binding.CoreBinding(emu_strict_lib, "rfwhip_", 0, 0, 1), scene, 96, 64, settings)
    b = _render(pkg, make_oracle(), scene, 96, 64, settings)
    assert (a[1]["prim"] != b[1]["prim"]).mean() <= 1e-3 and a[2] == b[2]


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["bench_terrain", "cornell", "atrium_textured", "cornell_lens"])
def test_strict_hip_equals_strict_emulation_bit_for_bit(pkg, emu_strict_lib, make_oracle, scene_name):
    """480 x 270 x 8 spp, pt depth 2: images, primary hit records and per-depth wave counts of the strict GPU build and the strict
    emulation are IDENTICAL (0 differing pixels), on the bench workload (1 002 528-triangle terrain), Cornell with instances,
    the textured instanced atrium and a thin-lens camera."""
    w, h = 480, 270
    if scene_name == "bench_terrain":
        scene = pkg.scenes.terrain(n=708, width=w, height_px=h)
    elif scene_name == "cornell":
        scene = pkg.scenes.cornell(w, h, geometric_emitter=True)
    elif scene_name == "cornell_lens":
        scene = pkg.scenes.cornell(w, h, geometric_emitter=True)
        scene.camera.aperture = 0.05
    else:
        scene = pkg.scenes.atrium(w, h, columns=6, tex_size=64)
    settings = {"integrator": "pt", "spp": 8, "max_depth": 2}
    strict_so = os.path.join(ROOT, "tests", "_strict", "librfwhip_strict.so")
    assert os.path.exists(strict_so), "build it with __graft_entry__.build() (build.py: build_strict)"
    hip = _render(pkg, pkg._binding.CoreBinding(ctypes.CDLL(strict_so), "rfwhip_", 0, 0, 1), scene, w, h, settings)
    emu = _render(pkg, pkg._binding.CoreBinding(emu_strict_lib, "rfwhip_", 0, 0, 1), scene, w, h, settings)
    differing = int((np.abs(hip[0] - emu[0]).max(-1) > 0).sum())
    print("%s: strict HIP vs strict emulation: %d of %d pixels differ; wave counts %s / %s" % (scene_name, differing, w * h, hip[2], emu[2]))
    assert hip[2] == emu[2], (hip[2], emu[2])
    for k in hip[1]:
        assert np.array_equal(hip[1][k], emu[1][k]), (k, int((hip[1][k] != emu[1][k]).sum()))
    assert differing == 0 and np.array_equal(hip[0], emu[0])
    if scene_name == "bench_terrain":
        # what the choice of math library flips: emulation-strict (polynomials) against the oracle (glibc)
        ora = _render(pkg, make_oracle(), scene, w, h, settings)
        frac = float((image_stats(emu[0], ora[0], 1e-3)[2] > 1e-3).mean())
        print("bench_terrain: strict emulation vs oracle (glibc): %.4f of the pixels beyond 1e-3" % frac)
        assert frac <= 3e-2, frac


@pytest.mark.gpu
def test_shipped_kernels_move_shading_not_geometry(pkg, make_hip):
    """The SHIPPED kernels against the strict build on the bench workload (480 x 270, 8 samples, pt depth 2), sample by sample:
    the shipped arithmetic (v_rcp_f32 in the triangle test and the slab set-up, v_sin / v_cos / v_exp, 1-ulp division and square
    root behind the hit record) may flip shading DECISIONS — at most 3 % of the pixels differ at all — but it must not move
    GEOMETRY: every primary ray of every sample finds the same triangle of the same instance at a distance within 5e-7 relative,
    barycentrics within 3e-7.  A future fast-math change that lets a ray through a crack would show here as another triangle, where
    the image tolerances would have absorbed it."""
    w, h, spp = 480, 270, 8
    scene = pkg.scenes.terrain(n=708, width=w, height_px=h)
    strict_so = os.path.join(ROOT, "tests", "_strict", "librfwhip_strict.so")
    assert os.path.exists(strict_so), "build it with __graft_entry__.build() (build.py: build_strict)"
    out = []
    for ctx in (make_hip(), pkg._binding.CoreBinding(ctypes.CDLL(strict_so), "rfwhip_", 0, 0, 1)):
        ctx.init(w, h)
        scene.upload(ctx)
        for k, v in {"integrator": "pt", "spp": 1, "max_depth": 2}.items():
            ctx.set_setting(k, v)
        hits = []
        for s in range(spp):  # one sample per call: the hit records of sample s, the image accumulates
            ctx.render_frame(scene.camera, pkg.RESET if s == 0 else pkg.CONVERGE)
            hits.append(ctx.primary_hits())
        out.append((ctx.framebuffer(), hits))
        ctx.destroy()
    (img_a, hits_a), (img_b, hits_b) = out
    differing = np.abs(img_a[..., :3] - img_b[..., :3]).max(-1) > 0
    print("shipped vs strict: %.4f of the pixels differ, %.4f beyond 1e-3" % (differing.mean(), (np.abs(img_a[..., :3] - img_b[..., :3]).max(-1) > 1e-3).mean()))
    assert (np.abs(img_a[..., :3] - img_b[..., :3]).max(-1) > 1e-3).mean() <= 3e-2
    other = 0
    for a, b in zip(hits_a, hits_b):
        same = (a["prim"] == b["prim"]) & (a["inst"] == b["inst"])
        other += int((~same).sum())
        hit = same & (a["prim"] >= 0)
        assert (np.abs(a["t"][hit] - b["t"][hit]) <= 5e-7 * b["t"][hit]).all()
        assert np.abs(a["u"][hit] - b["u"][hit]).max() <= 3e-7 and np.abs(a["v"][hit] - b["v"][hit]).max() <= 3e-7
    assert other == 0, "%d of %d primary rays found another triangle" % (other, w * h * spp)
