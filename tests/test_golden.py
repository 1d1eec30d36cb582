"""The oracle against the committed golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py: an
independent numpy float32 brute-force renderer — no BVH, instancing by transforming vertices, its own xor128).
Also the oracle against itself in brute-force mode, and — through the host-emulation build — the product's own host
logic + device arithmetic against the same vectors."""
import os

import numpy as np
import pytest

from conftest import image_stats

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = [("cornell96x64_center", "center"), ("cornell96x64_xor128", "xor128"), ("cards96x64_center", "center"), ("lens96x64_xor128", "xor128")]


def _render(pkg, ctx, jitter, w=96, h=64, name="cornell", **extra):
    import sys
    sys.path.insert(0, GOLD)
    import golden_scenes
    # cards: textured materials — retrieve_material's nearest lookup, fmod wrap, FLOAT4 -> UINT fall-through
    scene = (golden_scenes.cards_parity(pkg, w, h) if name.startswith("cards") else
             golden_scenes.cornell_lens_parity(pkg, w, h) if name.startswith("lens") else pkg.scenes.cornell(w, h))
    ctx.init(w, h)
    scene.upload(ctx)
    ctx.set_setting("integrator", "parity")
    ctx.set_setting("jitter", jitter)
    for k, v in extra.items():
        ctx.set_setting(k, v)
    ctx.render_frame(scene.camera, pkg.RESET)
    return ctx.framebuffer(), ctx.primary_hits()


def _check(img, hits, g, textured=False):
    assert (hits["prim"] != g["prim"]).sum() == 0
    assert (hits["inst"] != g["inst"]).sum() == 0
    m = g["prim"] >= 0
    assert (np.abs(hits["t"] - g["t"])[m] < 1e-4 + 1e-5 * np.abs(g["t"][m])).all()  # (lens: the origin carries cos / sin rounding)
    assert np.abs(hits["u"] - g["u"])[m].max() < 1e-4 and np.abs(hits["v"] - g["v"])[m].max() < 1e-4
    frac, rmse, d = image_stats(img, g["image"], 1e-3)
    # (textured: a texel index is uint(t * (size - 1)) — a pixel whose t lands on a texel edge may take the neighbour)
    assert frac <= (2e-3 if textured else 0.0) and rmse < (2e-3 if textured else 2e-4), (frac, rmse, d.max())
    assert np.array_equal(img[..., 3], g["image"][..., 3])  # alpha: 1 on hits, 0 on sky (Context.cpp:194,281)


@pytest.mark.parametrize("name,jitter", CASES)
def test_oracle_matches_golden(pkg, make_oracle, name, jitter):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    img, hits = _render(pkg, make_oracle(), jitter, name=name)
    _check(img, hits, g, name.startswith("cards"))


@pytest.mark.parametrize("name,jitter", CASES)
def test_oracle_bruteforce_mode_matches_golden(pkg, make_oracle, name, jitter):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    img, hits = _render(pkg, make_oracle(), jitter, name=name, bvh="0")
    _check(img, hits, g, name.startswith("cards"))


@pytest.mark.parametrize("name,jitter", CASES)
def test_emulated_core_matches_golden(pkg, make_emu, name, jitter):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    img, hits = _render(pkg, make_emu(), jitter, name=name)
    _check(img, hits, g, name.startswith("cards"))


def test_golden_covers_sky_and_all_instances():
    g = np.load(os.path.join(GOLD, "cornell96x64_center.npz"))
    assert 0.2 < (g["prim"] < 0).mean() < 0.5
    assert set(np.unique(g["inst"])) == {-1, 0, 1, 2}
