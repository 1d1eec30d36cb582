"""Caller-side scene assembly: what rfw::system does before it talks to a RenderContext, restated in numpy.

  material packing        RFW/system/src/rfw/material_list.cpp:318-481 (HostMaterial -> 192 B device material)
  host material defaults  RFW/system/src/rfw/material_list.h:48-66
  per-face records        RFW/system/context/rfw/context/structs.h:24-60, geometry/quad.cpp:6-42
  area lights             RFW/system/src/rfw/system.cpp:967-1032 (update_area_lights), context.cpp:6-15 (Heron)
  point/spot/dir lights   system.cpp:720-758
  synchronize order       system.cpp:247-433
  test sky                RFW/system/src/rfw/skybox.cpp:31-52

Also the synthetic workloads BASELINE.json names (there are no usable assets for them in the reference checkout):
cornell (configs 1/2), terrain_1m (config 3), atrium (config 4, Sponza-scale), skinned_tube (config 5).
"""
import math

import numpy as np

from . import abi
from .camera import Camera


# ----------------------------------------------------------------------------------------------------------------------
# materials
# ----------------------------------------------------------------------------------------------------------------------
def _tochar(a):  # TOCHAR, material_list.cpp:318
    return np.uint32(np.float32(a) * np.float32(255.0))


def _touint4(a, b, c, d):
    return np.uint32(_tochar(a) + (_tochar(b) << np.uint32(8)) + (_tochar(c) << np.uint32(16)) + (_tochar(d) << np.uint32(24)))


def host_material(color=(1.0, 1.0, 1.0), roughness=0.5, metallic=0.0, subsurface=0.0, specular=0.5, specularTint=0.0,
                  anisotropic=0.0, sheen=0.0, sheenTint=0.0, clearcoat=0.0, clearcoatGloss=1.0, transmission=0.0,
                  eta=1.0, absorption=(0.0, 0.0, 0.0), smooth=True, texture=-1, uvscale=(1.0, 1.0),
                  uvoffset=(0.0, 0.0), texture1=-1, texture2=-1, normalmap=-1, normalmap1=-1, normalmap2=-1,
                  alpha=False):
    """HostMaterial with the reference's defaults (material_list.h:48-66).  texture/texture1/texture2 are the diffuse
    layers (map[TEXTURE0..2]), normalmap/normalmap1/normalmap2 the normal-map layers (map[NORMALMAP0..2]); all layers
    share uvscale / uvoffset here.  alpha = the HASALPHA host flag."""
    return dict(color=color, roughness=roughness, metallic=metallic, subsurface=subsurface, specular=specular,
                specularTint=specularTint, anisotropic=anisotropic, sheen=sheen, sheenTint=sheenTint,
                clearcoat=clearcoat, clearcoatGloss=clearcoatGloss, transmission=transmission, eta=eta,
                absorption=absorption, smooth=smooth, texture=texture, uvscale=uvscale, uvoffset=uvoffset,
                texture1=texture1, texture2=texture2, normalmap=normalmap, normalmap1=normalmap1,
                normalmap2=normalmap2, alpha=alpha)


def pack_materials(host_materials, textures, faithful=False):
    """HostMaterial::convertToDeviceMaterial (material_list.cpp:320-481) for a list of host materials: the 192-byte
    device material and the per-slot texture ids (MaterialTexIds) a backend resolves the map addresses from.

    faithful=True reproduces two quirks of the reference's packer: a second diffuse layer sets the Has2ndNormalMap
    bit instead of Has2ndDiffuseMap (:372), and every present map writes its texture id into texaddr0 (:385-455), so
    only MaterialTexIds says which texture belongs to which slot.  The default packs what was meant."""
    out = np.zeros(len(host_materials), dtype=abi.MATERIAL_DTYPE)
    ids = np.full((len(host_materials), 11), -1, dtype=np.int32)
    for i, m in enumerate(host_materials):
        o = out[i]
        o["diffuse"] = np.asarray(m["color"], np.float32).astype(np.float16)
        o["transmittance"] = np.asarray(m["absorption"], np.float32).astype(np.float16)
        o["parameters"][0] = _touint4(m["metallic"], m["subsurface"], m["specular"], m["roughness"])
        o["parameters"][1] = _touint4(m["specularTint"], m["anisotropic"], m["sheen"], m["sheenTint"])
        o["parameters"][2] = _touint4(m["clearcoat"], m["clearcoatGloss"], m["transmission"], m["eta"] * 0.5)
        o["parameters"][3] = 0
        flags = 0
        if m["eta"] > 0:
            flags |= 1 << abi.MAT_IS_DIELECTRIC
        if m["smooth"]:
            flags |= 1 << abi.MAT_HAS_SMOOTH_NORMALS
        if m.get("alpha", False):
            flags |= 1 << abi.MAT_HAS_ALPHA
        # (host field, map slot = MaterialTexIds index, flag)
        layers = [("texture", 0, abi.MAT_HAS_DIFFUSE_MAP),
                  ("texture1", 1, abi.MAT_HAS_2ND_NORMAL_MAP if faithful else abi.MAT_HAS_2ND_DIFFUSE_MAP),
                  ("texture2", 2, abi.MAT_HAS_3RD_DIFFUSE_MAP),
                  ("normalmap", 3, abi.MAT_HAS_NORMAL_MAP), ("normalmap1", 4, abi.MAT_HAS_2ND_NORMAL_MAP),
                  ("normalmap2", 5, abi.MAT_HAS_3RD_NORMAL_MAP)]
        for field, slot, flag in layers:
            t = m.get(field, -1)
            if t < 0:
                continue
            tex = textures[t]
            flags |= 1 << flag
            if slot == 0 and tex["type"] == abi.TEX_FLOAT4:
                flags |= 1 << abi.MAT_DIFFUSE_MAP_IS_HDR
            d = o["map"][slot]
            d["width"], d["height"] = tex["width"], tex["height"]
            d["uscale"], d["vscale"] = np.float16(m["uvscale"][0]), np.float16(m["uvscale"][1])
            d["uoffs"], d["voffs"] = np.float16(m["uvoffset"][0]), np.float16(m["uvoffset"][1])
            if faithful:
                o["map"][0]["addr"] = t
            else:
                d["addr"] = t
            ids[i, slot] = t
        o["flags"] = flags
    return out, ids.view(abi.MATERIAL_TEX_IDS_DTYPE).reshape(-1)


def make_texture_rgba8(rgba_u8, mips=True):
    """UINT texture: texel = r | g<<8 | b<<16 | a<<24 (texture.cpp:77-81) with MIPLEVELCOUNT(5) box-filtered levels
    appended (texture.cpp:163-225 semantics: each level halves width and height)."""
    img = np.ascontiguousarray(rgba_u8, dtype=np.uint8)
    h, w, _ = img.shape
    levels = [img]
    if mips:
        cur = img.astype(np.float32)
        for _ in range(4):
            hh, ww = max(1, cur.shape[0] // 2), max(1, cur.shape[1] // 2)
            cur = cur[: hh * 2, : ww * 2].reshape(hh, 2, ww, 2, 4).mean(axis=(1, 3)) if cur.shape[0] >= 2 and cur.shape[1] >= 2 else cur[:hh, :ww]
            levels.append(np.clip(np.rint(cur), 0, 255).astype(np.uint8))
    flat = np.concatenate([l.reshape(-1, 4) for l in levels]).astype(np.uint32)
    data = flat[:, 0] | (flat[:, 1] << 8) | (flat[:, 2] << 16) | (flat[:, 3] << 24)
    return {"type": abi.TEX_UINT, "width": w, "height": h, "data": data.astype(np.uint32)}


def synthetic_blue_noise(seed=7):
    """A table with the LAYOUT of the reference's createBlueNoiseBuffer() (blue_noise.h:8204: 256 x 256 sequence values,
    then a 128 x 128 x 8 scrambling tile at word 65536 and a 128 x 128 x 8 ranking tile at word 3 * 65536, every word
    in 0..255) filled with our own numbers: per dimension a stratified permutation of 0..255 as the "sequence", random
    bytes as the tiles.  It exercises blueNoiseSampler's indexing exactly; it is NOT the published blue-noise data
    (that table belongs to the reference tree and reaches the core through rfwhip_set_blue_noise)."""
    rng = np.random.default_rng(seed)
    t = np.zeros(5 * 65536, np.uint32)
    seq = np.stack([rng.permutation(256) for _ in range(256)], axis=1)  # [sample][dimension]
    t[:65536] = seq.reshape(-1)
    t[65536:65536 + 131072] = rng.integers(0, 256, 131072)
    t[3 * 65536:3 * 65536 + 131072] = rng.integers(0, 256, 131072)
    return t


def make_texture_float4(rgba_f32):
    img = np.ascontiguousarray(rgba_f32, dtype=np.float32)
    h, w, _ = img.shape
    return {"type": abi.TEX_FLOAT4, "width": w, "height": h, "data": img.reshape(-1)}


# ----------------------------------------------------------------------------------------------------------------------
# geometry
# ----------------------------------------------------------------------------------------------------------------------
def triangle_area(v0, v1, v2):
    """Triangle::calculateArea (context.cpp:6-15), Heron's formula in fp32, vectorised."""
    v0, v1, v2 = (np.asarray(x, np.float32) for x in (v0, v1, v2))
    a = np.linalg.norm(v1 - v0, axis=-1).astype(np.float32)
    b = np.linalg.norm(v2 - v1, axis=-1).astype(np.float32)
    c = np.linalg.norm(v0 - v2, axis=-1).astype(np.float32)
    s = (a + b + c) * np.float32(0.5)
    return np.sqrt(np.maximum(s * (s - a) * (s - b) * (s - c), 0)).astype(np.float32)


def make_triangles(vertices, indices=None, normals=None, uvs=None, material=0):
    """Per-face Triangle records for a mesh; vertices (N,3|4), indices (M,3) or None (non-indexed).
    Built as a flat (M, 40) float32 table (the 160-byte record is 40 dwords) and viewed as TRIANGLE_DTYPE."""
    v = np.asarray(vertices, np.float32)[:, :3]
    if indices is None:
        idx = np.arange(len(v), dtype=np.uint32).reshape(-1, 3)
    else:
        idx = np.asarray(indices, np.uint32).reshape(-1, 3)
    m = len(idx)
    p0, p1, p2 = v[idx[:, 0]], v[idx[:, 1]], v[idx[:, 2]]
    n = np.cross(p1 - p0, p2 - p0).astype(np.float32)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    n = (n / np.maximum(ln, 1e-30)).astype(np.float32)
    # filled column-wise in a (40, M) table (contiguous rows), transposed once at the end
    tt = np.zeros((40, m), np.float32)
    it, ut = tt.view(np.int32), tt.view(np.uint32)
    it[3] = -1                                             # lightTriIdx
    ut[7] = np.asarray(material, np.uint32)                # material
    if uvs is not None:
        uv = np.asarray(uvs, np.float32)
        for k in range(3):
            tt[k] = uv[idx[:, k], 0]                       # u0,u1,u2
            tt[4 + k] = uv[idx[:, k], 1]                   # v0,v1,v2
    if normals is None:
        tt[8:11] = tt[12:15] = tt[16:19] = n.T             # vN0, vN1, vN2
    else:
        vn = np.asarray(normals, np.float32)
        tt[8:11], tt[12:15], tt[16:19] = vn[idx[:, 0]].T, vn[idx[:, 1]].T, vn[idx[:, 2]].T
    tt[11], tt[15], tt[19] = n[:, 0], n[:, 1], n[:, 2]     # Nx, Ny, Nz
    e = p1 - p0
    tang = (e / np.maximum(np.linalg.norm(e, axis=1, keepdims=True), 1e-30)).astype(np.float32)
    tt[20:23] = tang.T                                     # T
    tt[23] = triangle_area(p0, p1, p2)                     # area
    tt[24:27] = np.cross(n, tang).T                        # B
    tt[28:31], tt[32:35], tt[36:39] = p0.T, p1.T, p2.T
    tt[31] = tt[35] = tt[39] = 1.0
    tab = np.ascontiguousarray(tt.T)
    return tab.view(abi.TRIANGLE_DTYPE).reshape(m)


def quad(normal, pos, width, height):
    """geometry::Quad (quad.cpp:6-18): 6 non-indexed vertices, two triangles."""
    N = np.asarray(normal, np.float64)
    pos = np.asarray(pos, np.float64)
    tmp = np.array([0.0, 1.0, 0.0]) if N[0] > 0.9 else np.array([1.0, 0.0, 0.0])
    T = np.cross(N, tmp)
    T = 0.5 * width * T / np.linalg.norm(T)
    B = np.cross(T / np.linalg.norm(T), N)
    B = 0.5 * height * B / np.linalg.norm(B)
    return np.array([pos - B - T, pos + B - T, pos - B + T, pos + B - T, pos + B + T, pos - B + T], np.float32)


def _vec4(v3):
    v3 = np.asarray(v3, np.float32).reshape(-1, 3)
    return np.concatenate([v3, np.ones((len(v3), 1), np.float32)], 1)


# ----------------------------------------------------------------------------------------------------------------------
# scene container
# ----------------------------------------------------------------------------------------------------------------------
class Scene:
    """Host copy of everything rfw::system owns, plus upload() = system::synchronize (system.cpp:247-433)."""

    def __init__(self):
        self.meshes = []      # dict(vertices (N,4), indices (M,3)|None, triangles)
        self.instances = []   # dict(mesh, transform 4x4)
        self.host_materials = []
        self.textures = []
        self.point_lights = []
        self.spot_lights = []
        self.directional_lights = []
        self.area_lights = np.zeros(0, dtype=abi.AREA_LIGHT_DTYPE)
        self.sky = (np.zeros((1, 3), np.float32), 1, 1)
        self.camera = Camera()
        self.name = "scene"

    # -- building ----------------------------------------------------------------------------------------------------
    def add_material(self, **kw):
        self.host_materials.append(host_material(**kw))
        return len(self.host_materials) - 1

    def add_texture(self, tex):
        self.textures.append(tex)
        return len(self.textures) - 1

    def add_mesh(self, vertices, indices=None, normals=None, uvs=None, material=0):
        v4 = _vec4(np.asarray(vertices, np.float32)[:, :3])
        tris = make_triangles(v4, indices, normals, uvs, material)
        idx = None if indices is None else np.asarray(indices, np.uint32).reshape(-1, 3)
        self.meshes.append(dict(vertices=v4, indices=idx, triangles=tris))
        return len(self.meshes) - 1

    def add_instance(self, mesh, transform=None):
        t = np.eye(4) if transform is None else np.asarray(transform, np.float64).reshape(4, 4)
        self.instances.append(dict(mesh=mesh, transform=t))
        return len(self.instances) - 1

    def add_point_light(self, position, radiance):  # system.cpp:720-731
        self.point_lights.append((np.asarray(position, np.float32), np.asarray(radiance, np.float32)))

    def add_spot_light(self, position, inner_deg, radiance, outer_deg, direction):  # system.cpp:733-747
        d = np.asarray(direction, np.float64)
        self.spot_lights.append((np.asarray(position, np.float32), math.cos(math.radians(inner_deg)),
                                 np.asarray(radiance, np.float32), math.cos(math.radians(outer_deg)),
                                 (d / np.linalg.norm(d)).astype(np.float32)))

    def add_directional_light(self, direction, radiance):  # system.cpp:749-758
        d = np.asarray(direction, np.float64)
        self.directional_lights.append(((d / np.linalg.norm(d)).astype(np.float32), np.asarray(radiance, np.float32)))

    def add_area_light_quad(self, normal, pos, width, height, radiance):
        """An emitter that exists only in the light list (no geometry), as the parity scenes need
        (SURVEY §7 hard part b)."""
        q = quad(normal, pos, width, height)
        lights = np.zeros(2, dtype=abi.AREA_LIGHT_DTYPE)
        n = np.asarray(normal, np.float32)
        for k in range(2):
            v0, v1, v2 = q[3 * k], q[3 * k + 1], q[3 * k + 2]
            l = lights[k]
            l["vertex0"], l["vertex1"], l["vertex2"] = v0, v1, v2
            l["position"] = (v0 + v1 + v2) * np.float32(1.0 / 3.0)
            l["radiance"] = np.asarray(radiance, np.float32)
            l["energy"] = np.float32(np.linalg.norm(np.asarray(radiance, np.float32)))
            l["normal"] = n
            l["area"] = triangle_area(v0, v1, v2)
            l["triIdx"], l["instIdx"] = -1, -1
        self.area_lights = np.concatenate([self.area_lights, lights])

    def set_test_sky(self, width=512, height=256, base=0.1):
        """skybox::generate_test_sky (skybox.cpp:31-52) at a reduced resolution: grey + three 10x patches."""
        px = np.full((height, width, 3), base, np.float32)
        y0, y1 = int(height * 900 / 2560), int(height * 1100 / 2560)
        for k, x0 in enumerate((0, 2000, 4000)):
            xa, xb = int(width * x0 / 5120), int(width * (x0 + 200) / 5120)
            px[y0:y1, xa:xb] = 0
            px[y0:y1, xa:xb, k] = 10.0
        self.sky = (px.reshape(-1, 3), width, height)

    def set_gradient_sky(self, width=2048, height=1024):
        """Synthetic HDR equirect: horizon gradient + a sun disc + two bright patches."""
        v = ((np.arange(height, dtype=np.float32) + 0.5) / height)[:, None]
        u = ((np.arange(width, dtype=np.float32) + 0.5) / width)[None, :]
        hz = 1.0 - np.clip(1.0 - v * 2.0, 0, 1)
        dim = np.where(v > 0.5, np.float32(0.15), np.float32(1.0))
        px = np.empty((height, width, 3), np.float32)
        px[..., 0] = (0.35 + 0.25 * hz) * dim
        px[..., 1] = (0.45 + 0.30 * hz) * dim
        px[..., 2] = (0.95 - 0.25 * hz) * dim
        y0, y1 = int(height * 0.18), int(height * 0.26)
        x0, x1 = int(width * 0.28), int(width * 0.32)
        du, dv = u[:, x0:x1] - 0.30, v[y0:y1] - 0.22
        sun = (du * du * 4 + dv * dv) < 0.0004
        px[y0:y1, x0:x1][sun] = (20.0, 18.0, 13.0)
        px[int(height * 0.30):int(height * 0.34), int(width * 0.70):int(width * 0.74)] = (8.0, 2.0, 1.0)
        px[int(height * 0.10):int(height * 0.13), int(width * 0.55):int(width * 0.58)] = (1.0, 6.0, 9.0)
        self.sky = (px.reshape(-1, 3), width, height)

    # -- what rfw::system derives ----------------------------------------------------------------------------------------
    def update_area_lights(self):
        """system::update_area_lights (system.cpp:967-1032) for identity mesh transforms: every triangle whose material
        is emissive (any(color > 1)) becomes an area light in world space; writes lightTriIdx / area back."""
        emissive = [any(c > 1.0 for c in m["color"]) for m in self.host_materials]
        lights = []
        base = len(self.area_lights)
        for ii, inst in enumerate(self.instances):
            mesh = self.meshes[inst["mesh"]]
            tris = mesh["triangles"]
            em = np.nonzero(np.asarray(emissive, bool)[tris["material"]])[0]
            if not len(em):
                continue
            M = inst["transform"]
            Nm = np.linalg.inv(M[:3, :3]).T
            for ti in em:
                tri = tris[ti]
                v = [(M[:3, :3] @ tri[k].astype(np.float64) + M[:3, 3]).astype(np.float32) for k in ("vertex0", "vertex1", "vertex2")]
                l = np.zeros((), dtype=abi.AREA_LIGHT_DTYPE)
                l["vertex0"], l["vertex1"], l["vertex2"] = v
                l["position"] = (v[0] + v[1] + v[2]) * np.float32(1.0 / 3.0)
                col = np.asarray(self.host_materials[tri["material"]]["color"], np.float32)
                l["energy"] = np.float32(np.linalg.norm(col))
                l["radiance"] = col
                l["normal"] = (Nm @ np.array([tri["Nx"], tri["Ny"], tri["Nz"]], np.float64)).astype(np.float32)
                l["triIdx"], l["instIdx"] = int(ti), ii
                tris["lightTriIdx"][ti] = base + len(lights)
                tris["area"][ti] = triangle_area(tri["vertex0"], tri["vertex1"], tri["vertex2"])
                l["area"] = tris["area"][ti]
                lights.append(l)
        if lights:
            self.area_lights = np.concatenate([self.area_lights, np.array(lights, dtype=abi.AREA_LIGHT_DTYPE)])

    def triangle_count(self):
        return sum(len(self.meshes[i["mesh"]]["triangles"]) for i in self.instances)

    def light_arrays(self):
        p = np.zeros(len(self.point_lights), dtype=abi.POINT_LIGHT_DTYPE)
        for i, (pos, rad) in enumerate(self.point_lights):
            p[i]["position"], p[i]["radiance"] = pos, rad
            p[i]["energy"] = np.float32(np.sqrt(np.dot(rad, rad)))
        s = np.zeros(len(self.spot_lights), dtype=abi.SPOT_LIGHT_DTYPE)
        for i, (pos, ci, rad, co, d) in enumerate(self.spot_lights):
            s[i]["position"], s[i]["cosInner"], s[i]["radiance"], s[i]["cosOuter"], s[i]["direction"] = pos, ci, rad, co, d
            s[i]["energy"] = np.float32(np.sqrt(np.dot(rad, rad)))
        d_ = np.zeros(len(self.directional_lights), dtype=abi.DIRECTIONAL_LIGHT_DTYPE)
        for i, (d, rad) in enumerate(self.directional_lights):
            d_[i]["direction"], d_[i]["radiance"] = d, rad
            d_[i]["energy"] = np.float32(np.sqrt(np.dot(rad, rad)))
        return self.area_lights, p, s, d_

    # -- system::synchronize ------------------------------------------------------------------------------------------
    def upload(self, ctx):
        pix, w, h = self.sky
        ctx.set_sky(pix, w, h)
        ctx.set_textures(self.textures)
        mats, ids = pack_materials(self.host_materials, self.textures, faithful=getattr(self, "pack_faithful", False))
        ctx.set_materials(mats, ids)
        for i, m in enumerate(self.meshes):
            ctx.set_mesh(i, m["vertices"], m["triangles"], m["indices"])
        for i, inst in enumerate(self.instances):
            ctx.set_instance(i, inst["mesh"], inst["transform"])
        a, p, s, d = self.light_arrays()
        ctx.set_lights(a, p, s, d)
        ctx.update()


# ----------------------------------------------------------------------------------------------------------------------
# synthetic workloads
# ----------------------------------------------------------------------------------------------------------------------
def _box(lo, hi, skip_bottom=True):
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    c = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                  [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]], np.float32)
    faces = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 4, 7, 3), (1, 2, 6, 5), (3, 7, 6, 2)]
    if not skip_bottom:
        faces.append((0, 1, 5, 4))
    idx = []
    for a, b, c_, d in faces:
        idx += [(a, b, c_), (a, c_, d)]
    return c, np.asarray(idx, np.uint32)


def _rot_y(deg):
    a = math.radians(deg)
    m = np.eye(4)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = math.cos(a), math.sin(a), -math.sin(a), math.cos(a)
    return m


def _translate(x, y, z):
    m = np.eye(4)
    m[:3, 3] = (x, y, z)
    return m


def cornell(width=512, height=512, geometric_emitter=False, point_light=True):
    """BASELINE.json configs 1/2: five walls (10 tris) + two boxes without bottoms (20 tris) = 30 indexed triangles,
    one area light handed over through set_lights as two light triangles (radiance (20,20,20), cf.
    Examples/imgui_app/main.cpp:103), diffuse materials (system.cpp:672-680), camera FOV 40 / aperture 0.
    The two boxes are instances of ONE unit-box mesh with different transforms, so the two-level path is exercised.
    geometric_emitter=True adds an emissive quad (material colour > 1) as real geometry and derives the area lights
    from it the way rfw::system does — the path-tracing configuration."""
    s = Scene()
    s.name = "cornell"
    white = s.add_material(color=(0.73, 0.73, 0.73), roughness=1.0)
    red = s.add_material(color=(0.65, 0.05, 0.05), roughness=1.0)
    green = s.add_material(color=(0.12, 0.45, 0.15), roughness=1.0)
    boxm = s.add_material(color=(0.70, 0.70, 0.40), roughness=0.6)
    L = 5.0
    v = np.array([[-L, 0, -L], [L, 0, -L], [L, 0, L], [-L, 0, L], [-L, 2 * L, -L], [L, 2 * L, -L], [L, 2 * L, L],
                  [-L, 2 * L, L]], np.float32)
    # winding chosen so geometric normals face the room interior
    idx = np.array([[0, 2, 1], [0, 3, 2],      # floor   (+y)
                    [4, 5, 6], [4, 6, 7],      # ceiling (-y)
                    [3, 6, 2], [3, 7, 6],      # back    (-z facing camera at -z side looking +z)
                    [0, 4, 7], [0, 7, 3],      # left    (+x)
                    [1, 2, 6], [1, 6, 5]], np.uint32)  # right (-x)
    mats = np.array([white, white, white, white, white, white, red, red, green, green], np.uint32)
    room = s.add_mesh(v, idx, material=mats)
    s.add_instance(room)
    bv, bi = _box((-0.5, 0.0, -0.5), (0.5, 1.0, 0.5))
    box = s.add_mesh(bv, bi, material=boxm)
    sc1, sc2 = np.diag([3.0, 6.0, 3.0, 1.0]), np.diag([3.0, 3.0, 3.0, 1.0])
    s.add_instance(box, _translate(-1.8, 0, 1.5) @ _rot_y(18) @ sc1)
    s.add_instance(box, _translate(1.7, 0, -1.2) @ _rot_y(-17) @ sc2)
    if geometric_emitter:
        em = s.add_material(color=(20.0, 20.0, 20.0), roughness=1.0)
        q = quad((0.0, -1.0, 0.0), (0.0, 2 * L - 0.01, 0.0), 3.0, 3.0)
        qm = s.add_mesh(q, None, material=em)
        s.add_instance(qm)
        s.update_area_lights()
    else:
        s.add_area_light_quad((0.0, -1.0, 0.0), (0.0, 2 * L - 0.02, 0.0), 3.0, 3.0, (20.0, 20.0, 20.0))
    if point_light:
        s.add_point_light((-3.0, 7.0, -3.0), (6.0, 5.0, 4.0))
    s.set_test_sky(256, 128)
    cam = Camera(aperture=0.0, FOV=40.0, focalDistance=5.0)
    cam.look_at((0.37, L + 0.21, -3.6 * L), (0.0, L, 0.0))  # off-axis: no rays exactly along shared edges
    cam.resize(width, height)
    s.camera = cam
    return s


def cards(width=480, height=270, faithful=False):
    """PT-integrator feature scene: the Cornell room with
      * a "leaf card" in front of the tall box whose RGBA8 texture has alpha holes (HasAlpha: paths pass through),
      * a floor with a tangent-space normal map (two layers) and a detail colour layer added on top of the base map,
      * a third card with three diffuse layers.
    faithful=True packs the materials with the reference packer's quirks (pack_materials)."""
    s = cornell(width, height, geometric_emitter=True, point_light=True)
    s.name = "cards"
    s.pack_faithful = faithful
    n = 64
    yy, xx = np.mgrid[0:n, 0:n]
    # leaf card: green with round holes (alpha 0) on a 4x4 lattice
    cx, cy = (xx % 16) - 7.5, (yy % 16) - 7.5
    hole = (cx * cx + cy * cy) < 30.0
    leaf = np.zeros((n, n, 4), np.uint8)
    leaf[..., 0] = 40 + (xx * 3) % 50
    leaf[..., 1] = 150 + (yy * 2) % 90
    leaf[..., 2] = 40
    leaf[..., 3] = np.where(hole, 0, 255)
    t_leaf = s.add_texture(make_texture_rgba8(leaf))
    # base colour: checker; detail layers: faint stripes
    base = np.zeros((n, n, 4), np.uint8)
    chk = ((xx // 8) + (yy // 8)) % 2
    base[..., 0] = np.where(chk, 230, 120)
    base[..., 1] = np.where(chk, 220, 110)
    base[..., 2] = np.where(chk, 200, 100)
    base[..., 3] = 255
    t_base = s.add_texture(make_texture_rgba8(base))
    det = np.zeros((n, n, 4), np.uint8)
    det[..., 0] = np.where((xx // 2) % 2, 30, 0)
    det[..., 2] = np.where((yy // 2) % 2, 40, 0)
    det[..., 3] = 255
    t_det = s.add_texture(make_texture_rgba8(det))
    det2 = np.zeros((n, n, 4), np.uint8)
    det2[..., 1] = ((xx + yy) % 8) * 6
    det2[..., 3] = 255
    t_det2 = s.add_texture(make_texture_rgba8(det2))
    # normal maps: sinusoidal bumps, encoded n * 0.5 + 0.5
    def nmap(fx, fy, amp):
        dx = amp * np.cos(2 * np.pi * fx * xx / n)
        dy = amp * np.cos(2 * np.pi * fy * yy / n)
        nn = np.stack([-dx, -dy, np.ones_like(dx)], -1)
        nn /= np.linalg.norm(nn, axis=-1, keepdims=True)
        img = np.zeros((n, n, 4), np.uint8)
        img[..., :3] = np.clip(np.rint((nn * 0.5 + 0.5) * 255.0), 0, 255).astype(np.uint8)
        img[..., 3] = 255
        return img
    t_n0 = s.add_texture(make_texture_rgba8(nmap(4, 4, 0.8), mips=False))
    t_n1 = s.add_texture(make_texture_rgba8(nmap(16, 1, 0.3), mips=False))
    m_leaf = s.add_material(color=(1.0, 1.0, 1.0), roughness=0.9, texture=t_leaf, alpha=True, uvscale=(2.0, 2.0))
    m_floor = s.add_material(color=(0.9, 0.9, 0.9), roughness=0.7, texture=t_base, texture1=t_det, normalmap=t_n0,
                             normalmap1=t_n1, uvscale=(3.0, 3.0))
    m_three = s.add_material(color=(0.8, 0.8, 0.8), roughness=0.8, texture=t_base, texture1=t_det, texture2=t_det2,
                             normalmap=t_n1, normalmap1=t_n0, normalmap2=t_n0)
    L = 5.0
    def card(p0, ex, ey, mat):
        p0, ex, ey = np.asarray(p0, np.float32), np.asarray(ex, np.float32), np.asarray(ey, np.float32)
        v = np.array([p0, p0 + ex, p0 + ex + ey, p0 + ey], np.float32)
        idx = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
        uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
        m = s.add_mesh(v, idx, uvs=uv, material=mat)
        s.add_instance(m)
    card((-4.5, 0.5, -2.5), (3.5, 0.0, 0.4), (0.0, 4.5, 0.0), m_leaf)    # in front of the tall box, slightly turned
    card((-L + 0.02, 0.02, -L + 0.02), (0.0, 0.0, 2 * L - 0.04), (2 * L - 0.04, 0.0, 0.0), m_floor)  # just above the floor
    card((1.2, 3.2, 0.5), (2.6, 0.0, -0.8), (0.0, 2.6, 0.0), m_three)
    return s


def _accumulate_vertex_normals(nverts, idx, face_normals):
    """Area-weighted smooth vertex normals (sum of the incident un-normalised face normals)."""
    vn = np.zeros((nverts, 3), np.float64)
    for k in range(3):
        for a in range(3):
            vn[:, a] += np.bincount(idx[:, k], weights=face_normals[:, a], minlength=nverts)
    return (vn / np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-30)).astype(np.float32)


def _hash01(ix, iz, seed):
    """Deterministic uniform [0,1) per lattice point (integer hash; seed 0x5EED for the displaced grid)."""
    h = (ix.astype(np.uint64) * np.uint64(0x9E3779B1) + iz.astype(np.uint64) * np.uint64(0x85EBCA77) +
         np.uint64(seed)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(12)
    h = (h * np.uint64(0x297A2D39)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    return (h.astype(np.float64) / 4294967296.0)


def terrain(n=708, extent=100.0, height=6.0, seed=0x5EED, width=1920, height_px=1080, lights=True):
    """BASELINE.json config 3: displaced grid, n x n cells x 2 = 2 n^2 triangles (n = 708 -> 1 002 528), one indexed
    mesh, smooth per-vertex normals, synthetic HDR sky, a handful of emissive quads and point lights so that
    next-event estimation and the shadow wave have work."""
    s = Scene()
    s.name = "terrain_%dk" % (2 * n * n // 1000)
    ground = s.add_material(color=(0.55, 0.50, 0.42), roughness=0.9)
    glossy = s.add_material(color=(0.80, 0.80, 0.85), roughness=0.25, metallic=0.6)
    g = np.arange(n + 1)
    ix, iz = np.meshgrid(g, g, indexing="xy")
    x = (ix / n - 0.5) * extent
    z = (iz / n - 0.5) * extent
    # multi-octave value noise from the lattice hash + a fine uniform displacement in [-h,h]/8
    y = np.zeros_like(x, dtype=np.float64)
    for octave, amp in ((8, 1.0), (32, 0.45), (96, 0.18)):
        cx, cz = ix * octave / n, iz * octave / n
        x0, z0 = np.floor(cx).astype(np.int64), np.floor(cz).astype(np.int64)
        fx, fz = cx - x0, cz - z0
        fx, fz = fx * fx * (3 - 2 * fx), fz * fz * (3 - 2 * fz)
        h00, h10 = _hash01(x0, z0, seed + octave), _hash01(x0 + 1, z0, seed + octave)
        h01, h11 = _hash01(x0, z0 + 1, seed + octave), _hash01(x0 + 1, z0 + 1, seed + octave)
        y += amp * ((h00 * (1 - fx) + h10 * fx) * (1 - fz) + (h01 * (1 - fx) + h11 * fx) * fz - 0.5)
    y = y * height + (_hash01(ix, iz, seed) - 0.5) * (height / 40.0)
    verts = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    i00 = (iz[:-1, :-1] * (n + 1) + ix[:-1, :-1]).ravel()
    i10, i01, i11 = i00 + 1, i00 + (n + 1), i00 + (n + 2)
    idx = np.empty((2 * n * n, 3), np.uint32)
    idx[0::2] = np.stack([i00, i01, i10], 1)
    idx[1::2] = np.stack([i10, i01, i11], 1)
    # smooth normals
    p0, p1, p2 = verts[idx[:, 0]], verts[idx[:, 1]], verts[idx[:, 2]]
    fn = np.cross(p1 - p0, p2 - p0)
    vn = _accumulate_vertex_normals(len(verts), idx, fn)
    cellx = (np.arange(2 * n * n) // 2) % n
    cellz = (np.arange(2 * n * n) // 2) // n
    mats = np.where(((cellx // max(1, n // 12)) + (cellz // max(1, n // 12))) % 5 == 0, glossy, ground).astype(np.uint32)
    mesh = s.add_mesh(verts, idx, normals=vn, material=mats)
    s.add_instance(mesh)
    if lights:
        em = s.add_material(color=(12.0, 10.4, 8.0), roughness=1.0)
        qs = []
        for (px, pz) in ((-0.25, -0.2), (0.22, 0.05), (-0.05, 0.3), (0.3, -0.3)):
            qs.append(quad((0.0, -1.0, 0.0), (px * extent, height * 2.2, pz * extent), extent * 0.04, extent * 0.04))
        qv = np.concatenate(qs)
        qm = s.add_mesh(qv, None, material=em)
        s.add_instance(qm)
        s.update_area_lights()
        s.add_point_light((0.0, height * 3.0, 0.0), (60.0, 57.0, 52.0))
        s.add_point_light((extent * 0.3, height * 2.0, extent * 0.25), (25.0, 33.0, 43.0))
    s.set_gradient_sky(2048, 1024)
    cam = Camera(aperture=0.0, FOV=40.0, focalDistance=5.0)
    cam.look_at((0.0, height * 3.0, -extent * 0.62), (0.0, height * 0.6, 0.0))
    cam.resize(width, height_px)
    s.camera = cam
    return s


def _checker_texture(size, c0, c1, cells=8, seed=1):
    yy, xx = np.mgrid[0:size, 0:size]
    m = (((xx * cells) // size + (yy * cells) // size) % 2).astype(np.float32)[..., None]
    rng = np.random.default_rng(seed)
    noise = rng.uniform(-12, 12, (size, size, 1)).astype(np.float32)
    img = m * np.asarray(c0, np.float32) + (1 - m) * np.asarray(c1, np.float32) + noise
    rgba = np.concatenate([np.clip(img, 0, 255), np.full((size, size, 1), 255, np.float32)], -1)
    return make_texture_rgba8(rgba.astype(np.uint8))


def atrium(width=1920, height=1080, columns=10, tex_size=256):
    """BASELINE.json config 4 stand-in (sponza.obj is absent from the reference checkout): a two-storey colonnade —
    instanced tessellated columns and arches, textured floor/walls/drapes, ~26 materials, ~260 k triangles."""
    s = Scene()
    s.name = "atrium"
    rng = np.random.default_rng(7)
    texs = [s.add_texture(_checker_texture(tex_size, rng.integers(90, 230, 3), rng.integers(40, 160, 3), cells=int(rng.integers(4, 17)), seed=i)) for i in range(12)]
    mats = []
    for i in range(24):
        mats.append(s.add_material(color=tuple(float(x) for x in rng.uniform(0.45, 0.95, 3)),
                                   roughness=float(rng.uniform(0.2, 1.0)), metallic=float(rng.uniform(0, 0.3)),
                                   texture=texs[i % len(texs)] if i % 2 == 0 else -1,
                                   uvscale=(float(rng.integers(1, 6)), float(rng.integers(1, 6)))))
    # tessellated column (cylinder with entasis): rings x segments
    seg, rings = 64, 44
    th = np.linspace(0, 2 * np.pi, seg, endpoint=False)
    yy = np.linspace(0, 1, rings + 1)
    rad = 0.45 * (1 - 0.18 * yy[:, None] ** 2) * (1 + 0.04 * np.cos(th[None, :] * 12))
    vx = rad * np.cos(th)[None, :]
    vz = rad * np.sin(th)[None, :]
    cv = np.stack([vx, np.repeat(yy[:, None] * 6.0, seg, 1), vz], -1).reshape(-1, 3).astype(np.float32)
    uv = np.stack([np.repeat((th / (2 * np.pi))[None, :], rings + 1, 0), np.repeat(yy[:, None], seg, 1)], -1).reshape(-1, 2).astype(np.float32)
    ci = []
    for r in range(rings):
        for k in range(seg):
            a, b = r * seg + k, r * seg + (k + 1) % seg
            c, d = a + seg, b + seg
            ci += [(a, c, b), (b, c, d)]
    ci = np.asarray(ci, np.uint32)
    cn = cv.copy()
    cn[:, 1] = 0
    cn /= np.maximum(np.linalg.norm(cn, axis=1, keepdims=True), 1e-9)
    col_meshes = [s.add_mesh(cv, ci, normals=cn, uvs=uv, material=mats[2 + k]) for k in range(4)]
    # floor / walls as tessellated grids
    def grid(nx, nz, origin, ax_u, ax_v, mat_ids, bump=0.0):
        gu, gv = np.meshgrid(np.arange(nx + 1), np.arange(nz + 1), indexing="xy")
        p = (np.asarray(origin, np.float64)[None, None, :] + gu[..., None] / nx * np.asarray(ax_u, np.float64) +
             gv[..., None] / nz * np.asarray(ax_v, np.float64))
        nrm = np.cross(ax_u, ax_v)
        nrm = nrm / np.linalg.norm(nrm)
        if bump:
            p = p + nrm * (bump * (np.sin(gu * 0.9) * np.cos(gv * 0.7))[..., None])
        verts = p.reshape(-1, 3).astype(np.float32)
        uv_ = np.stack([gu / nx * 8.0, gv / nz * 8.0], -1).reshape(-1, 2).astype(np.float32)
        a = (gv[:-1, :-1] * (nx + 1) + gu[:-1, :-1]).ravel()
        idx = np.empty((2 * nx * nz, 3), np.uint32)
        idx[0::2] = np.stack([a, a + 1, a + nx + 1], 1)
        idx[1::2] = np.stack([a + 1, a + nx + 2, a + nx + 1], 1)
        m = np.asarray(mat_ids, np.uint32)[(np.arange(2 * nx * nz) // 2 // max(1, (nx * nz) // len(mat_ids))) % len(mat_ids)]
        return s.add_mesh(verts, idx, uvs=uv_, material=m)

    W_, D_, H_ = 30.0, 14.0, 13.0
    s.add_instance(grid(120, 60, (-W_ / 2, 0, -D_ / 2), (0, 0, D_), (W_, 0, 0), mats[8:12], bump=0.01))       # floor (+y)
    s.add_instance(grid(50, 100, (-W_ / 2, 0, D_ / 2), (0, H_, 0), (W_, 0, 0), mats[12:16], bump=0.02))        # back wall (-z)
    s.add_instance(grid(50, 50, (-W_ / 2, 0, -D_ / 2), (0, H_, 0), (0, 0, D_), mats[16:18], bump=0.02))        # left wall (+x)
    s.add_instance(grid(50, 50, (W_ / 2, 0, D_ / 2), (0, H_, 0), (0, 0, -D_), mats[18:20], bump=0.02))         # right wall (-x)
    s.add_instance(grid(30, 60, (-W_ / 2, 6.3, D_ / 2 - 4.0), (0, 0, 4.0), (W_, 0, 0), mats[20:22]))           # gallery (+y)
    for storey in range(2):
        for k in range(columns):
            x = -W_ / 2 + (k + 0.5) * W_ / columns
            for zrow in (D_ / 2 - 4.0, -D_ / 2 + 2.0):
                m = col_meshes[(k + storey) % 4]
                s.add_instance(m, _translate(x, storey * 6.4, zrow) @ _rot_y(13.0 * k))
    em = s.add_material(color=(18.0, 16.0, 13.0), roughness=1.0)
    lq = np.concatenate([quad((0.0, -1.0, 0.0), (x, H_ - 0.3, 0.0), 2.0, 2.0) for x in (-9.0, -3.0, 3.0, 9.0)])
    s.add_instance(s.add_mesh(lq, None, material=em))
    s.update_area_lights()
    s.add_point_light((0.0, 5.0, -3.0), (40.0, 36.0, 30.0))
    s.set_gradient_sky(1024, 512)
    cam = Camera(aperture=0.0, FOV=55.0, focalDistance=5.0)
    cam.look_at((-12.2, 3.4, -6.2), (6.0, 4.6, 3.0))
    cam.resize(width, height)
    s.camera = cam
    return s


def skinned_tube(frame=0.0, rings=160, seg=96, width=1920, height=1080):
    """BASELINE.json config 5 stand-in: a two-bone skinned tube whose pose depends on `frame`; vertex/triangle counts
    never change, so every frame after the first takes the refit path of set_mesh
    (EmbreeRT/src/Mesh.cpp:33-35)."""
    s = Scene()
    s.name = "skinned_tube"
    m0 = s.add_material(color=(0.75, 0.35, 0.25), roughness=0.5)
    fl = s.add_material(color=(0.6, 0.6, 0.6), roughness=0.9)
    v, idx, vn = skinned_tube_pose(frame, rings, seg)
    s.add_instance(s.add_mesh(v, idx, normals=vn, material=m0))
    fv = np.array([[-12, 0, -12], [12, 0, -12], [12, 0, 12], [-12, 0, 12]], np.float32)
    s.add_instance(s.add_mesh(fv, np.array([[0, 2, 1], [0, 3, 2]], np.uint32), material=fl))
    s.add_point_light((4.0, 9.0, -6.0), (90.0, 85.0, 80.0))
    s.add_area_light_quad((0.0, -1.0, 0.0), (0.0, 12.0, 0.0), 4.0, 4.0, (12.0, 12.0, 12.0))
    s.set_test_sky(256, 128)
    cam = Camera(aperture=0.0, FOV=40.0)
    cam.look_at((0.0, 5.0, -14.0), (0.0, 4.0, 0.0))
    cam.resize(width, height)
    s.camera = cam
    return s


def skinned_tube_pose(frame, rings=160, seg=96):
    """Linear-blend skinning of a vertical tube with two bones (cf. geometry/gltf/mesh.cpp:18-125 which does this on
    the host with TBB): bone 1 bends the upper half by an angle that oscillates with `frame`."""
    th = np.linspace(0, 2 * np.pi, seg, endpoint=False)
    yy = np.linspace(0, 8.0, rings + 1)
    base = np.stack([np.repeat(np.cos(th)[None, :], rings + 1, 0), np.repeat(yy[:, None], seg, 1),
                     np.repeat(np.sin(th)[None, :], rings + 1, 0)], -1).reshape(-1, 3)
    w = np.clip((base[:, 1] - 3.0) / 2.0, 0, 1)[:, None]
    ang = 0.9 * math.sin(frame * 0.35)
    c, s_ = math.cos(ang), math.sin(ang)
    piv = np.array([0.0, 4.0, 0.0])
    rel = base - piv
    rot = np.stack([rel[:, 0] * c - rel[:, 1] * s_, rel[:, 0] * s_ + rel[:, 1] * c, rel[:, 2]], -1) + piv
    pos = (base * (1 - w) + rot * w).astype(np.float32)
    ci = []
    for r in range(rings):
        for k in range(seg):
            a, b = r * seg + k, r * seg + (k + 1) % seg
            ci += [(a, a + seg, b), (b, a + seg, b + seg)]
    idx = np.asarray(ci, np.uint32)
    fn = np.cross(pos[idx[:, 1]] - pos[idx[:, 0]], pos[idx[:, 2]] - pos[idx[:, 0]])
    vn = _accumulate_vertex_normals(len(pos), idx, fn)
    return pos, idx, vn


def skinned_tube_rig(rings=160, seg=96):
    """The tube of skinned_tube as a RIG: bind-pose vertices / indices / vertex normals plus per-vertex joints and
    weights for two joints (0: root, 1: the bone that bends the upper half) — what a glTF skin provides
    (geometry/gltf/mesh.cpp:360-372)."""
    v, idx, vn = skinned_tube_pose(0.0, rings, seg)  # frame 0: the bend angle is 0 = bind pose
    w1 = np.clip((v[:, 1] - 3.0) / 2.0, 0, 1).astype(np.float32)
    joints = np.zeros((len(v), 4), np.uint32)
    joints[:, 1] = 1
    weights = np.zeros((len(v), 4), np.float32)
    weights[:, 0], weights[:, 1] = 1.0 - w1, w1
    return v, idx, vn, joints, weights


def skinned_tube_joint_matrices(frame):
    """(2, 4, 4) joint matrices of the rig at `frame` (row-major, acting on column vectors)."""
    ang = 0.9 * math.sin(frame * 0.35)
    c, s_ = math.cos(ang), math.sin(ang)
    rot = np.array([[c, -s_, 0, 0], [s_, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float64)
    m1 = _translate(0.0, 4.0, 0.0) @ rot @ _translate(0.0, -4.0, 0.0)
    return np.stack([np.eye(4), m1]).astype(np.float32)
