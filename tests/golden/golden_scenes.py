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


def cards_parity(pkg, W=96, H=64):
    """Textured scene for the parity integrator (EmbreeRT's retrieve_material reads the first diffuse map only, nearest
    texel): the Cornell room of the untextured fixture plus three cards — an RGBA8 checker with scale 3, an RGBA8 pattern
    with scale 2 and a FLOAT4 texture (the case that falls through into the UINT case there) with a scale and a negative
    offset, so that the fmod wrap and its sign fix are exercised."""
    import numpy as np
    s = pkg.scenes.cornell(W, H)
    s.name = "cards_parity"
    n = 64
    yy, xx = np.mgrid[0:n, 0:n]
    chk = ((xx // 8) + (yy // 8)) % 2
    base = np.zeros((n, n, 4), np.uint8)
    base[..., 0], base[..., 1], base[..., 2], base[..., 3] = np.where(chk, 230, 120), np.where(chk, 220, 110), np.where(chk, 200, 100), 255
    pat = np.zeros((n, n, 4), np.uint8)
    pat[..., 0], pat[..., 1], pat[..., 2], pat[..., 3] = 40 + (xx * 3) % 50, 150 + (yy * 2) % 90, 40 + ((xx + yy) % 16) * 8, 255
    yy, xx = np.mgrid[0:16, 0:16]
    img = np.zeros((16, 16, 4), np.float32)
    img[..., 0], img[..., 1], img[..., 2], img[..., 3] = 0.25 + xx / 20.0, 0.9 - yy / 24.0, 0.3 + ((xx + yy) % 4) * 0.15, 1.0
    t_base = s.add_texture(pkg.scenes.make_texture_rgba8(base))
    t_pat = s.add_texture(pkg.scenes.make_texture_rgba8(pat, mips=False))
    t_f4 = s.add_texture(pkg.scenes.make_texture_float4(img))
    m_base = s.add_material(color=(0.9, 0.9, 0.9), roughness=0.7, texture=t_base, uvscale=(3.0, 3.0))
    m_pat = s.add_material(color=(1.0, 1.0, 1.0), roughness=0.9, texture=t_pat, uvscale=(2.0, 2.0), uvoffset=(0.125, 0.0))
    m_f4 = s.add_material(color=(0.9, 0.8, 0.7), roughness=0.8, texture=t_f4, uvscale=(1.5, 2.0), uvoffset=(0.25, -0.75))
    def card(p0, ex, ey, mat):
        p0, ex, ey = (np.asarray(v, np.float32) for v in (p0, ex, ey))
        v = np.array([p0, p0 + ex, p0 + ex + ey, p0 + ey], np.float32)
        idx = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
        uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
        s.add_instance(s.add_mesh(v, idx, uvs=uv, material=mat))
    L = 5.0
    card((-L + 0.02, 0.02, -L + 0.02), (0.0, 0.0, 2 * L - 0.04), (2 * L - 0.04, 0.0, 0.0), m_base)  # just above the floor
    card((-4.5, 0.5, -2.5), (3.5, 0.0, 0.4), (0.0, 4.5, 0.0), m_pat)
    card((4.2, 0.3, -3.5), (-3.5, 0.0, 0.2), (0.0, 2.5, 0.0), m_f4)   # in front of the short box, facing the camera
    return s


def cornell_lens(pkg, W=96, H=64):
    """cornell_pt seen through a thin lens (aperture 0.35, focus on the tall box): generatePrimaryRay's nine-blade aperture
    sampling (Kernels.cu:401-416)."""
    s = cornell_pt(pkg, W, H)
    s.name = "cornell_lens"
    s.camera.aperture = 0.35
    s.camera.focalDistance = 17.0
    return s


def cornell_lens_parity(pkg, W=96, H=64):
    """The untextured parity fixture's room through a thin lens (aperture 0.35): Ray::generateFromView's aperture sampling."""
    s = pkg.scenes.cornell(W, H)
    s.name = "cornell_lens_parity"
    s.camera.aperture = 0.35
    s.camera.focalDistance = 17.0
    return s
