"""Scenes of the committed golden vectors (tests/golden/make_golden_pt.py) — shared by the generator and the tests, so both
sides render exactly the same description."""


def cornell_pt(pkg, W=96, H=64):
    """Cornell room, emissive quad as geometry (area lights derived like rfw::system does) + one point light."""
    return pkg.scenes.cornell(W, H, geometric_emitter=True, point_light=True)


def cornell_lights(pkg, W=96, H=64):
    """The same room lit by every light type of lights.h: 2 area triangles, a point, a spot and a directional light."""
    s = pkg.scenes.cornell(W, H, geometric_emitter=True, point_light=True)
    s.name = "cornell_lights"
    s.add_spot_light((2.5, 8.5, -2.0), 18.0, (60.0, 40.0, 30.0), 32.0, (-0.35, -1.0, 0.45))
    s.add_directional_light((0.3, -1.0, 0.6), (1.5, 1.6, 2.0))
    return s


def terrain_small(pkg, W=96, H=64):
    """The bench workload (BASELINE config 3) at 12 x 12 cells = 288 triangles: smooth vertex normals, a glossy metallic and
    a rough material, 8 emissive light triangles + 2 point lights, the synthetic 2048 x 1024 HDR sky."""
    return pkg.scenes.terrain(n=12, width=W, height_px=H)


def cards_pt(pkg, W=96, H=64):
    """The pt-integrator feature scene (scenes.cards): a card with alpha holes (paths pass through), a floor with two
    normal-map layers and a detail colour layer, a card with three diffuse layers and three normal-map layers — mip-mapped
    RGBA8 textures — in the Cornell room."""
    return pkg.scenes.cards(W, H)
