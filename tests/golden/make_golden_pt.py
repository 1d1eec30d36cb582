#!/usr/bin/env python3
"""Generate tests/golden/pt_*.npz — an INDEPENDENT float32 numpy restatement of the reference's WAVEFRONT PATH TRACER,
written from the reference sources only (not from rendering-fw_amd/csrc/rt_core.h or oracle/rfw_oracle.c):

    RFW/system/context/rfw/bsdf/disney.h:18-280      Refract, SchlickFresnel, GTR1/2, SmithGGX, Fr, BSDFPdf/Eval/Sample
    RFW/system/context/rfw/bsdf/tools.h:10-29,86,103-123,163-235   PackNormal, UnpackNormal, SurvivalProbability,
                                                     DiffuseReflection*, SafeOrigin, blueNoiseSampler, clampIntensity,
                                                     createTangentSpace, WangHash, RandomInt/Float
    RFW/system/context/rfw/bsdf/compat.h:47-74       ShadingData parameter unpacking
    RFW/backends/CUDART/src/lights.h:17-265          Potential*Contribution, LightPickProb, RandomBarycentrics,
                                                     RandomPointOnLight, CalculateLightPDF
    RFW/backends/CUDART/src/getShadingData.h:22-217  FetchTexel (bilinear), FetchTexelTrilinear, normals / tangent frame,
                                                     diffuse and normal-map layers, the alpha flag (the `cards` scene)
    RFW/backends/CUDART/src/Kernels.cu:383-426       generatePrimaryRay (the hash-RNG branch)
    RFW/backends/CUDART/src/Kernels.cu:428-499       intersect_rays (closest hit record, shadow connections)
    RFW/backends/CUDART/src/Kernels.cu:571-794       shade_rays
    RFW/backends/CUDART/src/CUDAIntersect.h:11-94    intersect_triangle (+ area-ratio barycentrics)
    RFW/backends/CUDART/src/Context.cpp:65-159       host loop: which waves are launched
    RFW/system/context/rfw/context/Camera.cpp:74-115 get_view (shared with make_golden.py)

Run in the development container:   python tests/golden/make_golden_pt.py

Everything is vectorised over the paths of one wave, in float32, one numpy operation per C operation in the
reference's order of evaluation.  There is NO BVH: every ray is tested against every triangle of every instance (in the
instance's object space, like the reference's two-level traversal, so `t` is shared).

Where the reference's behaviour is undefined, the golden states what it does instead:
  * DeviceTriangle::getLightTriangleIndex reads the material id (device_structs.h:37,40) and indexes an uninitialised
    potential[] entry with it: the light-triangle index (u4.w) is used, which is what the code means;
  * uint(65535 * bary) of a barycentric a rounding error below 0: clamped to [0, 1];
  * acos(D.y) with |D.y| a rounding error above 1: clamped.
These are not reference OUTPUTS (the reference cannot be built here): they are a third, independent reading of the
reference text that pins the two restatements the product is tested with (the C oracle and the HIP kernels).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from __graft_entry__ import load_package  # noqa: E402
from make_golden import camera_view  # noqa: E402  (Camera.cpp:74-88, numpy)
import golden_scenes  # noqa: E402

f32 = np.float32
u32 = np.uint32

INVPI = f32(0.318309886183790671537767526745028724)
PI = f32(3.14159265358979323846264338327950288)
INV2PI = f32(0.159154943091895335768883763372514362)
TWOPI = f32(6.28318530717958647692528676655900576)
MIN_ROUGHNESS = f32(0.01)   # settings.h:4
MAX_PATH_LENGTH = 2         # settings.h:5
GEO_EPS = f32(1e-5)         # geometryEpsilon (CUDART/src/Context.cpp:49)
T_EPSILON = f32(1e-6)       # Kernels.cu:23

np.seterr(all="ignore")


# ----------------------------------------------------------------------------------------------------------------------
# glm-style float32 vector algebra on (n, 3) arrays
# ----------------------------------------------------------------------------------------------------------------------
def V(x):
    return np.asarray(x, dtype=f32)


def dot(a, b):
    return ((a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]).astype(f32)


def cross(a, b):  # glm::cross
    return np.stack([a[..., 1] * b[..., 2] - b[..., 1] * a[..., 2], a[..., 2] * b[..., 0] - b[..., 2] * a[..., 0],
                     a[..., 0] * b[..., 1] - b[..., 0] * a[..., 1]], -1).astype(f32)


def s3(s, v):  # scalar (n,) times vector (n,3)
    return (np.asarray(s, f32)[..., None] * v).astype(f32)


def normalize(v):  # glm::normalize = v * inversesqrt(dot(v, v))
    return s3(f32(1) / np.sqrt(dot(v, v)), v)


def length(v):
    return np.sqrt(dot(v, v)).astype(f32)


def lerp(a, b, t):  # disney_lerp
    return (a + t * (b - a)).astype(f32)


def lerp3(a, b, t):
    return (a + np.asarray(t, f32)[..., None] * (b - a)).astype(f32)


def sqr(x):
    return (x * x).astype(f32)


# ----------------------------------------------------------------------------------------------------------------------
# tools.h
# ----------------------------------------------------------------------------------------------------------------------
def wang_hash(s):  # tools.h:218-225
    s = np.asarray(s, np.uint64) & 0xFFFFFFFF
    s = ((s ^ 61) ^ (s >> 16)) & 0xFFFFFFFF
    s = (s * 9) & 0xFFFFFFFF
    s = (s ^ (s >> 4)) & 0xFFFFFFFF
    s = (s * 0x27d4eb2d) & 0xFFFFFFFF
    s = (s ^ (s >> 15)) & 0xFFFFFFFF
    return s.astype(np.uint64)


def random_int(s):  # tools.h:227-233 (returns the new state = the value)
    s = np.asarray(s, np.uint64)
    s = (s ^ (s << 13)) & 0xFFFFFFFF
    s = (s ^ (s >> 17)) & 0xFFFFFFFF
    s = (s ^ (s << 5)) & 0xFFFFFFFF
    return s


def random_float(s):  # tools.h:235 — returns (value, new state)
    s = random_int(s)
    return (s.astype(f32) * f32(2.3283064365387e-10)).astype(f32), s


def to_uint(x):
    """(uint)float for the non-negative in-range values the reference converts."""
    return np.clip(np.nan_to_num(x, nan=0.0), 0, 4294967040.0).astype(np.uint64).astype(np.uint32)


def pack_normal(N):  # tools.h:10-21
    f = (f32(65535.0) / np.maximum(np.sqrt(f32(8.0) * N[..., 2] + f32(8.0)), f32(0.0001))).astype(f32)
    return (to_uint(N[..., 0] * f + f32(32767.0)) + (to_uint(N[..., 1] * f + f32(32767.0)) << u32(16))).astype(np.uint32)


def unpack_normal(p):  # tools.h:22-29
    p = np.asarray(p, np.uint32)
    x = ((p & u32(65535)).astype(f32) * f32(2.0 / 65535.0)).astype(f32) + f32(-1)
    y = ((p >> u32(16)).astype(f32) * f32(2.0 / 65535.0)).astype(f32) + f32(-1)
    z = np.full_like(x, 1.0)
    w = np.full_like(x, -1.0)
    l = ((x * -x + y * -y) + z * -w).astype(f32)
    nz = l
    l = np.sqrt(l).astype(f32)
    x, y = (x * l).astype(f32), (y * l).astype(f32)
    return np.stack([x * f32(2), y * f32(2), nz * f32(2) + f32(-1)], -1).astype(f32)


def survival_probability(d):  # tools.h:86
    return np.minimum(f32(1), np.maximum(np.maximum(d[..., 0], d[..., 1]), d[..., 2])).astype(f32)


def diffuse_reflection_uniform(r0, r1):  # tools.h:103-109
    term1 = (TWOPI * r0).astype(f32)
    term2 = np.sqrt(f32(1) - r1 * r1).astype(f32)
    s, c = np.sin(term1).astype(f32), np.cos(term1).astype(f32)
    return np.stack([c * term2, s * term2, r1], -1).astype(f32)


def diffuse_reflection_cos_weighted(r0, r1):  # tools.h:111-117 (sqrt(1.0 - r1) is evaluated in double)
    term1 = (TWOPI * r0).astype(f32)
    term2 = np.sqrt(1.0 - r1.astype(np.float64)).astype(f32)
    s, c = np.sin(term1).astype(f32), np.cos(term1).astype(f32)
    return normalize(np.stack([c * term2, s * term2, np.sqrt(r1).astype(f32)], -1).astype(f32))


def clamp_intensity(v, clamp_value):  # tools.h:184-192
    m = np.maximum(v[..., 0], np.maximum(v[..., 1], v[..., 2]))
    scale = np.where(m > clamp_value, f32(clamp_value) / m, f32(1)).astype(f32)
    return np.where((m > clamp_value)[..., None], s3(scale, v), v).astype(f32)


def create_tangent_space(N):  # tools.h:204-211
    s = np.sign(N[..., 2]).astype(f32)
    a = (f32(-1) / (s + N[..., 2])).astype(f32)
    b = (N[..., 0] * N[..., 1] * a).astype(f32)
    T = np.stack([f32(1) + s * N[..., 0] * N[..., 0] * a, s * b, -s * N[..., 0]], -1).astype(f32)
    B = np.stack([b, s + N[..., 1] * N[..., 1] * a, -N[..., 1]], -1).astype(f32)
    return T, B


def blue_noise_sampler(table, x, y, sample_idx, dim):  # tools.h:163-181
    x, y, sample_idx, dim = (np.asarray(v, np.int64) for v in (x, y, sample_idx, dim))
    x, y, sample_idx, dim = x & 127, y & 127, sample_idx & 255, dim & 255
    ranked = sample_idx ^ table[dim + (x + y * 128) * 8 + 65536 * 3].astype(np.int64)
    value = table[dim + ranked * 256].astype(np.int64)
    value = value ^ table[(dim & 7) + (x + y * 128) * 8 + 65536].astype(np.int64)
    return ((f32(0.5) + value.astype(f32)) * f32(1.0 / 256.0)).astype(f32)


# ----------------------------------------------------------------------------------------------------------------------
# compat.h ShadingData: colour, absorption, parameters (uvec4)
# ----------------------------------------------------------------------------------------------------------------------
class SD:
    def __init__(self, color, absorption, params):
        self.color = V(color)
        self.absorption = V(absorption)
        self.p = np.asarray(params, np.uint32)

    def _c(self, word, shift):
        return (((self.p[..., word] >> u32(shift)) & u32(255)).astype(f32) * f32(1.0 / 255.0)).astype(f32)

    METALLIC = property(lambda s: s._c(0, 0))
    SUBSURFACE = property(lambda s: s._c(0, 8))
    SPECULAR = property(lambda s: s._c(0, 16))
    ROUGHNESS = property(lambda s: np.maximum(f32(0.001), s._c(0, 24)).astype(f32))
    SPECTINT = property(lambda s: s._c(1, 0))
    CLEARCOAT = property(lambda s: s._c(2, 0))
    CLEARCOATGLOSS = property(lambda s: s._c(2, 8))
    TRANSMISSION = property(lambda s: s._c(2, 16))
    ETA = property(lambda s: s._c(2, 24))

    def take(self, idx):
        return SD(self.color[idx], self.absorption[idx], self.p[idx])


# ----------------------------------------------------------------------------------------------------------------------
# disney.h
# ----------------------------------------------------------------------------------------------------------------------
def refract(wi, n, eta):  # disney.h:18-28 -> (ok, wt)
    cosThetaI = dot(n, wi)
    sin2ThetaI = np.maximum(f32(0), f32(1) - cosThetaI * cosThetaI).astype(f32)
    sin2ThetaT = (eta * eta * sin2ThetaI).astype(f32)
    ok = ~(sin2ThetaT >= 1)
    cosThetaT = np.sqrt(f32(1) - sin2ThetaT).astype(f32)
    wt = (s3(eta, wi * f32(-1)) + s3(eta * cosThetaI - cosThetaT, n)).astype(f32)
    return ok, wt


def schlick_fresnel(u):  # disney.h:30-34
    m = np.clip(f32(1) - u, f32(0), f32(1)).astype(f32)
    return ((m * m) * (m * m) * m).astype(f32)


def gtr1(NDotH, a):  # disney.h:36-43
    a2 = (a * a).astype(f32)
    t = (f32(1) + (a2 - f32(1)) * NDotH * NDotH).astype(f32)
    r = ((a2 - f32(1)) / (PI * np.log(a2).astype(f32) * t)).astype(f32)
    return np.where(a >= 1, INVPI, r).astype(f32)


def gtr2(NDotH, a):  # disney.h:45-50
    a2 = (a * a).astype(f32)
    t = (f32(1) + (a2 - f32(1)) * NDotH * NDotH).astype(f32)
    return (a2 / (PI * t * t)).astype(f32)


def smith_ggx(NDotv, alphaG):  # disney.h:52-57
    a = (alphaG * alphaG).astype(f32)
    b = (NDotv * NDotv).astype(f32)
    return (f32(1) / (NDotv + np.sqrt(a + b - a * b).astype(f32))).astype(f32)


def fr(VDotN, eio):  # disney.h:59-70
    SinThetaT2 = (sqr(eio) * (f32(1) - VDotN * VDotN)).astype(f32)
    LDotN = np.sqrt(f32(1) - SinThetaT2).astype(f32)
    eta = (f32(1) / eio).astype(f32)
    r1 = ((VDotN - eta * LDotN) / (VDotN + eta * LDotN)).astype(f32)
    r2 = ((LDotN - eta * VDotN) / (LDotN + eta * VDotN)).astype(f32)
    return np.where(SinThetaT2 > 1, f32(1), f32(0.5) * (sqr(r1) + sqr(r2))).astype(f32)


def safe_normalize(a):  # disney.h:72-79
    ls = dot(a, a)
    return np.where((ls > 0)[..., None], s3(f32(1) / np.sqrt(ls), a), f32(0)).astype(f32)


def bsdf_pdf(sd, N, wo, wi):  # disney.h:83-101
    neg = dot(wi, N) <= 0
    brdf_neg = (INV2PI * sd.SUBSURFACE * f32(0.5)).astype(f32)
    F = fr(dot(N, wo), sd.ETA)
    halfway = safe_normalize((wi + wo).astype(f32))
    cosThetaHalf = np.abs(dot(halfway, N))
    pdfHalf = (gtr2(cosThetaHalf, sd.ROUGHNESS) * cosThetaHalf).astype(f32)
    pdfSpec = (f32(0.25) * pdfHalf / np.maximum(f32(1e-6), dot(wi, halfway))).astype(f32)
    pdfDiff = (np.abs(dot(wi, N)) * INVPI * (f32(1) - sd.SUBSURFACE)).astype(f32)
    bsdfPdf = np.where(neg, f32(0), pdfSpec * F).astype(f32)
    brdfPdf = np.where(neg, brdf_neg, lerp(pdfDiff, pdfSpec, f32(0.5))).astype(f32)
    return lerp(brdfPdf, bsdfPdf, sd.TRANSMISSION)


def bsdf_eval(sd, N, wo, wi, t, backfacing):  # disney.h:104-185
    one3 = np.ones_like(sd.color)
    NDotL, NDotV = dot(N, wi), dot(N, wo)
    H = normalize((wi + wo).astype(f32))
    NDotH, LDotH = dot(N, H), dot(wi, H)
    Cdlin = sd.color
    Cdlum = ((f32(.3) * Cdlin[..., 0] + f32(.6) * Cdlin[..., 1]) + f32(.1) * Cdlin[..., 2]).astype(f32)
    Ctint = np.where((Cdlum > 0)[..., None], Cdlin / Cdlum[..., None], one3).astype(f32)
    METALLIC, TRANSMISSION, SUBSURFACE, ROUGHNESS, ETA = sd.METALLIC, sd.TRANSMISSION, sd.SUBSURFACE, sd.ROUGHNESS, sd.ETA
    Cspec0 = lerp3(s3(sd.SPECULAR * f32(.08), lerp3(one3, Ctint, sd.SPECTINT)), Cdlin, METALLIC)
    # --- BSDF part (TRANSMISSION > 0)
    F = fr(NDotV, ETA)
    bsdf_a = (((f32(1) - F) / np.abs(NDotL) * (f32(1) - METALLIC)) * TRANSMISSION).astype(f32)
    a = ROUGHNESS
    Ds = gtr2(NDotH, a)
    FHt = fr(LDotH, ETA)
    Fst = lerp3(Cspec0, one3, FHt)
    Gs = (smith_ggx(NDotV, a) * smith_ggx(NDotL, a)).astype(f32)
    bsdf_b = s3(Gs * Ds, Fst)
    bsdf = np.where((NDotL <= 0)[..., None], bsdf_a[..., None] * one3, bsdf_b).astype(f32)
    bsdf = np.where((TRANSMISSION > 0)[..., None], bsdf, f32(0)).astype(f32)
    # --- BRDF part (TRANSMISSION < 1)
    s = np.sqrt(sd.color).astype(f32)
    FLn, FVn = schlick_fresnel(np.abs(NDotL)), schlick_fresnel(NDotV)
    Fdn = ((f32(1) - f32(0.5) * FLn) * (f32(1) - f32(0.5) * FVn)).astype(f32)
    brdf_a = s3(f32(1) - METALLIC, s3(Fdn, s3(SUBSURFACE, (INVPI * s).astype(f32))))
    brdf_a = np.where((SUBSURFACE > 0)[..., None], brdf_a, f32(0)).astype(f32)
    FH = schlick_fresnel(LDotH)
    Fs = lerp3(Cspec0, one3, FH)
    FL, FV = schlick_fresnel(NDotL), schlick_fresnel(NDotV)
    Fd90 = (f32(0.5) + f32(2.0) * LDotH * LDotH * a).astype(f32)
    Fd = (lerp(f32(1), Fd90, FL) * lerp(f32(1), Fd90, FV)).astype(f32)
    Dr = gtr1(NDotH, lerp(f32(.1), f32(.001), sd.CLEARCOATGLOSS))
    Fc = lerp(f32(.04), f32(1), FH)
    Gr = (smith_ggx(NDotL, f32(.25)) * smith_ggx(NDotV, f32(.25))).astype(f32)
    term1 = s3(f32(1) - SUBSURFACE, s3(f32(1) - METALLIC, s3(INVPI * Fd, Cdlin)))
    term2 = s3(Ds, s3(Gs, Fs))
    term3 = (sd.CLEARCOAT * Gr * Fc * Dr).astype(f32)
    brdf_b = ((term1 + term2) + term3[..., None]).astype(f32)
    brdf = np.where((NDotL <= 0)[..., None], brdf_a, brdf_b).astype(f32)
    brdf = np.where((TRANSMISSION < 1)[..., None], brdf, f32(0)).astype(f32)
    final = lerp3(brdf, bsdf, TRANSMISSION)
    att = np.exp(-sd.absorption * np.asarray(t, f32)[..., None]).astype(f32)
    return np.where(np.asarray(backfacing, bool)[..., None], final * att, final).astype(f32)


def reflect(I, N):  # glm::reflect
    return (I - s3(dot(N, I), N) * f32(2)).astype(f32)


def ggx_halfway(T, B, N, wo, rough, r1, r2):  # the two identical blocks disney.h:199-206 / :247-254
    cosThetaHalf = np.sqrt((f32(1) - r2) / (f32(1) + (sqr(rough) - f32(1)) * r2)).astype(f32)
    sinThetaHalf = np.sqrt(np.maximum(f32(0), f32(1) - sqr(cosThetaHalf))).astype(f32)
    sinPhiHalf = np.sin((r1 * TWOPI).astype(f32)).astype(f32)
    cosPhiHalf = np.cos((r1 * TWOPI).astype(f32)).astype(f32)
    h = ((s3(sinThetaHalf * cosPhiHalf, T) + s3(sinThetaHalf * sinPhiHalf, B)) + s3(cosThetaHalf, N)).astype(f32)
    return np.where((dot(h, wo) <= 0)[..., None], h * f32(-1), h).astype(f32)


def bsdf_sample(sd, T, B, N, wo, r3, r4, pdf_in):  # disney.h:188-262 -> (wi, pdf); pdf_in = the caller's value (0)
    n = len(r3)
    transmission, ROUGHNESS, ETA, subsurface = sd.TRANSMISSION, sd.ROUGHNESS, sd.ETA, sd.SUBSURFACE
    wi = np.zeros((n, 3), f32)
    wi[:, 2] = 1  # the caller's uninitialised R: never used when pdf stays 0
    pdf = np.asarray(pdf_in, f32).copy()
    br_t = r3 < transmission
    # --- sample BSDF
    F = fr(dot(N, wo), ETA)
    refl = br_t & (r4 < F)
    wi_refl = reflect(wo * f32(-1), ggx_halfway(T, B, N, wo, ROUGHNESS, (r3 / transmission).astype(f32), (r4 / F).astype(f32)))
    ok, wt = refract(wo, N, ETA)
    trans = br_t & ~(r4 < F)
    wi = np.where(refl[:, None], wi_refl, wi)
    wi = np.where((trans & ok)[:, None], wt, wi)
    pdf = np.where(trans, np.where(ok, (f32(1) - F) * transmission, f32(0)), pdf).astype(f32)  # reflection: pdf untouched
    # --- sample BRDF
    r1 = ((r3 - transmission) / (f32(1) - transmission)).astype(f32)
    diff = ~br_t & (r4 < f32(0.5))
    r2d = (r4 * f32(2)).astype(f32)
    sub = r2d < subsurface
    d_sub = diffuse_reflection_uniform(r1, (r2d / subsurface).astype(f32))
    d_sub[:, 2] *= f32(-1)
    d_cos = diffuse_reflection_cos_weighted(r1, ((r2d - subsurface) / (f32(1) - subsurface)).astype(f32))
    d = np.where(sub[:, None], d_sub, d_cos).astype(f32)
    wi_diff = ((s3(d[:, 0], T) + s3(d[:, 1], B)) + s3(d[:, 2], N)).astype(f32)
    spec = ~br_t & ~(r4 < f32(0.5))
    wi_spec = reflect(wo * f32(-1), ggx_halfway(T, B, N, wo, ROUGHNESS, r1, ((r4 - f32(0.5)) * f32(2.0)).astype(f32)))
    wi = np.where(diff[:, None], wi_diff, wi)
    wi = np.where(spec[:, None], wi_spec, wi).astype(f32)
    pdf = np.where(~br_t, bsdf_pdf(sd, N, wo, wi), pdf).astype(f32)
    return wi, pdf


# ----------------------------------------------------------------------------------------------------------------------
# lights.h
# ----------------------------------------------------------------------------------------------------------------------
class Lights:
    def __init__(self, area, point, spot, direc):
        self.area, self.point, self.spot, self.dir = area, point, spot, direc

    def count(self):
        return len(self.area) + len(self.point) + len(self.spot) + len(self.dir)


def area_energy(l):  # DeviceAreaLight::getEnergy, device_structs.h:115 — length(radiance), not pos_energy.w
    return length(V(l["radiance"]))


def pot_area(l, O, N, I, bary):  # lights.h:17-36
    LN = V(l["normal"])
    if bary is None:  # bary.x < 0
        L = I
    else:
        L = ((s3(bary[:, 0], V(l["vertex0"])[None]) + s3(bary[:, 1], V(l["vertex1"])[None])) + s3(bary[:, 2], V(l["vertex2"])[None])).astype(f32)
    L = (L - O).astype(f32)
    att = (f32(1) / dot(L, L)).astype(f32)
    L = normalize(L)
    LNdotL = np.maximum(f32(0), -dot(LN[None], L)).astype(f32)
    NdotL = np.maximum(f32(0), dot(N, L)).astype(f32)
    return (f32(l["energy"]) * LNdotL * NdotL * att).astype(f32)  # posEnergy.w


def pot_point(l, I, N):  # lights.h:38-46
    L = (V(l["position"])[None] - I).astype(f32)
    NdotL = np.maximum(f32(0), dot(N, L)).astype(f32)
    att = (f32(1) / dot(L, L)).astype(f32)
    return (f32(l["energy"]) * NdotL * att).astype(f32)


def pot_spot(l, I, N):  # lights.h:48-67
    L = (V(l["position"])[None] - I).astype(f32)
    att = (f32(1) / dot(L, L)).astype(f32)
    L = normalize(L)
    d = ((np.maximum(f32(0), -dot(L, V(l["direction"])[None])) - f32(l["cosOuter"])) / (f32(l["cosInner"]) - f32(l["cosOuter"]))).astype(f32)
    NdotL = np.maximum(f32(0), dot(N, L)).astype(f32)
    LNdotL = np.maximum(f32(0), np.minimum(f32(1), d)).astype(f32)
    return (f32(l["energy"]) * LNdotL * NdotL * att).astype(f32)


def pot_dir(l, N):  # lights.h:69-76
    LNdotL = np.maximum(f32(0), -dot(V(l["direction"])[None], N)).astype(f32)
    return (f32(l["energy"]) * LNdotL).astype(f32)


def light_pick_prob(lt, idx, O, N, I):  # lights.h:83-116 (IS_LIGHTS)
    n = len(O)
    total = np.zeros(n, f32)
    mine = np.zeros(n, f32)
    for i, l in enumerate(lt.area):
        c = pot_area(l, O, N, I, None)
        mine = np.where(idx == i, c, mine)
        total = (total + c).astype(f32)
    for l in lt.point:
        total = (total + pot_point(l, O, N)).astype(f32)
    for l in lt.spot:
        total = (total + pot_spot(l, O, N)).astype(f32)
    for l in lt.dir:
        total = (total + pot_dir(l, N)).astype(f32)
    return np.where(total <= 0, f32(0), mine / total).astype(f32)


def random_barycentrics(r0):  # lights.h:119-157
    uf = to_uint(r0 * f32(4294967295.0)).astype(np.uint64)
    n = len(r0)
    A = np.tile(V([1, 0]), (n, 1))
    B = np.tile(V([0, 1]), (n, 1))
    C = np.tile(V([0, 0]), (n, 1))
    h = f32(0.5)
    for i in range(16):
        d = ((uf >> np.uint64(2 * (15 - i))) & np.uint64(3)).astype(np.int64)[:, None]
        An = np.where(d == 0, (B + C) * h, np.where(d == 1, A, np.where(d == 2, (B + A) * h, (C + A) * h)))
        Bn = np.where(d == 0, (A + C) * h, np.where(d == 1, (A + B) * h, np.where(d == 2, B, (C + B) * h)))
        Cn = np.where(d == 0, (A + B) * h, np.where(d == 1, (A + C) * h, np.where(d == 2, (B + C) * h, C)))
        A, B, C = An.astype(f32), Bn.astype(f32), Cn.astype(f32)
    r = (((A + B) + C) * f32(0.3333333)).astype(f32)
    return np.stack([r[:, 0], r[:, 1], f32(1) - r[:, 0] - r[:, 1]], -1).astype(f32)


def random_point_on_light(lt, r0, r1, I, N):  # lights.h:159-265 -> P, pickProb, lightPdf, lightColor
    n = len(r0)
    bary = random_barycentrics(r0)
    pots = [pot_area(l, I, N, np.zeros_like(I), bary) for l in lt.area]
    pots += [pot_point(l, I, N) for l in lt.point]
    pots += [pot_spot(l, I, N) for l in lt.spot]
    pots += [pot_dir(l, N) for l in lt.dir]
    pots = np.stack(pots, 1).astype(f32)  # (n, lights)
    total = np.zeros(n, f32)
    for k in range(pots.shape[1]):
        total = (total + pots[:, k]).astype(f32)
    none = total <= 0
    r1s = (r1 * total).astype(f32)
    run = np.zeros(n, f32)
    idx = np.zeros(n, np.int64)
    found = np.zeros(n, bool)
    for k in range(pots.shape[1]):
        run = (run + pots[:, k]).astype(f32)
        hit = ~found & (run >= r1s)
        idx = np.where(hit, k, idx)
        found |= hit
    pickProb = (pots[np.arange(n), idx] / total).astype(f32)
    P = np.ones((n, 3), f32)
    lightPdf = np.zeros(n, f32)
    color = np.zeros((n, 3), f32)
    base = 0
    for k, l in enumerate(lt.area):
        sel = idx == base + k
        Pk = ((s3(bary[:, 0], V(l["vertex0"])[None]) + s3(bary[:, 1], V(l["vertex1"])[None])) + s3(bary[:, 2], V(l["vertex2"])[None])).astype(f32)
        L = (I - Pk).astype(f32)
        sqDist = dot(L, L)
        L = normalize(L)
        LNdotL = dot(L, V(l["normal"])[None])
        reci = (sqDist / (f32(l["area"]) * LNdotL)).astype(f32)
        pdf = np.where((LNdotL > 0) & (dot(L, N) < 0), reci * (f32(1) / area_energy(l)), f32(0)).astype(f32)
        P, lightPdf, color = np.where(sel[:, None], Pk, P), np.where(sel, pdf, lightPdf), np.where(sel[:, None], V(l["radiance"])[None], color)
    base += len(lt.area)
    for k, l in enumerate(lt.point):
        sel = idx == base + k
        pos = V(l["position"])[None]
        L = (I - pos).astype(f32)
        sqDist = dot(L, L)
        pdf = np.where(dot(L, N) < 0, sqDist / f32(l["energy"]), f32(0)).astype(f32)
        P, lightPdf, color = np.where(sel[:, None], pos, P), np.where(sel, pdf, lightPdf), np.where(sel[:, None], V(l["radiance"])[None], color)
    base += len(lt.point)
    for k, l in enumerate(lt.spot):
        sel = idx == base + k
        pos = V(l["position"])[None]
        L = (I - pos).astype(f32)
        sqDist = dot(L, L)
        L = normalize(L)
        d = (np.maximum(f32(0), dot(L, V(l["direction"])[None]) - f32(l["cosOuter"])) / (f32(l["cosInner"]) - f32(l["cosOuter"]))).astype(f32)
        LNdotL = np.minimum(f32(1), d).astype(f32)
        pdf = np.where((LNdotL > 0) & (dot(L, N) < 0), sqDist / (LNdotL * f32(l["energy"])), f32(0)).astype(f32)
        P, lightPdf, color = np.where(sel[:, None], pos, P), np.where(sel, pdf, lightPdf), np.where(sel[:, None], V(l["radiance"])[None], color)
    base += len(lt.spot)
    for k, l in enumerate(lt.dir):
        sel = idx == base + k
        Ld = V(l["direction"])[None]
        NdotL = dot(Ld, N)
        pdf = np.where(NdotL < 0, f32(1) * f32(1.0 / np.float64(f32(l["energy"]))), f32(0)).astype(f32)
        Pk = (I - f32(1000.0) * Ld).astype(f32)
        P, lightPdf, color = np.where(sel[:, None], Pk, P), np.where(sel, pdf, lightPdf), np.where(sel[:, None], V(l["radiance"])[None], color)
    # no potential light: lightPdf = 0, returns vec3(1); pickProb stays the caller's uninitialised variable (never used
    # because lightPdf == 0): reported as 0
    pickProb = np.where(none, f32(0), pickProb).astype(f32)
    color = np.where(none[:, None], f32(0), color)
    P = np.where(none[:, None], f32(1), P).astype(f32)
    lightPdf = np.where(none, f32(0), lightPdf).astype(f32)
    return P, pickProb, lightPdf, color.astype(f32)


# ----------------------------------------------------------------------------------------------------------------------
# geometry: brute force over all triangles of all instances (CUDAIntersect.h:11-94, Kernels.cu:226-381)
# ----------------------------------------------------------------------------------------------------------------------
class Geometry:
    def __init__(self, scene):
        self.inst = []
        for ii, inst in enumerate(scene.instances):
            m = scene.meshes[inst["mesh"]]
            M = np.asarray(inst["transform"], np.float64)
            inv = np.linalg.inv(M).astype(f32)
            nrm = np.linalg.inv(M[:3, :3]).T.astype(f32)  # the mat3 handed to set_instance (system.cpp:347)
            v = m["vertices"][:, :3].astype(f32)
            idx = m["indices"] if m["indices"] is not None else np.arange(len(v), dtype=np.uint32).reshape(-1, 3)
            self.inst.append(dict(inv=inv, nrm=nrm, p0=v[idx[:, 0]], p1=v[idx[:, 1]], p2=v[idx[:, 2]], tris=m["triangles"]))

    @staticmethod
    def to_object(inv, O, D):
        # mat4 * vec4(origin, 1), mat4 * vec4(direction, 0): column-major glm product
        o = (((O[:, 0:1] * inv[:3, 0][None] + O[:, 1:2] * inv[:3, 1][None]) + O[:, 2:3] * inv[:3, 2][None]) + inv[:3, 3][None]).astype(f32)
        d = ((D[:, 0:1] * inv[:3, 0][None] + D[:, 1:2] * inv[:3, 1][None]) + D[:, 2:3] * inv[:3, 2][None]).astype(f32)
        return o, d

    @staticmethod
    def _mt(o, d, p0, p1, p2):
        e1, e2 = (p1 - p0).astype(f32), (p2 - p0).astype(f32)
        h = cross(d, e2[None])
        a = dot(e1[None], h)
        ok = ~((a > -T_EPSILON) & (a < T_EPSILON))
        f = (f32(1) / a).astype(f32)
        s = (o - p0[None]).astype(f32)
        u = (f * dot(s, h)).astype(f32)
        ok &= ~((u < 0) | (u > 1))
        q = cross(s, e1[None])
        v = (f * dot(d, q)).astype(f32)
        ok &= ~((v < 0) | (u + v > 1))
        t = (f * dot(e2[None], q)).astype(f32)
        return ok, t, e1, e2

    def closest(self, O, D, tmin):
        n = len(O)
        t = np.full(n, 1e34, f32)
        inst = np.full(n, -1, np.int32)
        prim = np.full(n, -1, np.int32)
        bx = np.zeros(n, f32)
        by = np.zeros(n, f32)
        for ii, g in enumerate(self.inst):
            o, d = self.to_object(g["inv"], O, D)
            for k in range(len(g["p0"])):
                p0, p1, p2 = g["p0"][k], g["p1"][k], g["p2"][k]
                ok, tt, e1, e2 = self._mt(o, d, p0, p1, p2)
                ok &= (tt > tmin) & (t > tt)
                if not ok.any():
                    continue
                # barycentrics by area ratios (CUDAIntersect.h:82-88)
                p = (o + s3(tt, d)).astype(f32)
                cr = cross(e1[None], e2[None])
                Nn = normalize(cr)
                areaABC = dot(Nn, cr)
                areaPBC = dot(Nn, cross((p1[None] - p).astype(f32), (p2[None] - p).astype(f32)))
                areaPCA = dot(Nn, cross((p2[None] - p).astype(f32), (p0[None] - p).astype(f32)))
                t = np.where(ok, tt, t)
                inst = np.where(ok, ii, inst)
                prim = np.where(ok, k, prim)
                bx = np.where(ok, (areaPBC / areaABC).astype(f32), bx)
                by = np.where(ok, (areaPCA / areaABC).astype(f32), by)
        # 16:16 packing of the hit record (Kernels.cu:454-459) and its decode (:619-620)
        qx = to_uint(f32(65535.0) * np.clip(bx, 0, 1)) & u32(65535)
        qy = to_uint(f32(65535.0) * np.clip(by, 0, 1)) & u32(65535)
        bu = (qx.astype(f32) * f32(1.0 / 65535.0)).astype(f32)
        bv = (qy.astype(f32) * f32(1.0 / 65535.0)).astype(f32)
        return t.astype(f32), inst, prim, bu, bv

    def occluded(self, O, D, tmin, tmax):
        occ = np.zeros(len(O), bool)
        for g in self.inst:
            o, d = self.to_object(g["inv"], O, D)
            for k in range(len(g["p0"])):
                ok, tt, _, _ = self._mt(o, d, g["p0"][k], g["p1"][k], g["p2"][k])
                occ |= ok & (tt > tmin) & (tmax > tt)
        return occ


# ----------------------------------------------------------------------------------------------------------------------
# the wavefront loop (CUDART/src/Context.cpp:65-159 + Kernels.cu)
# ----------------------------------------------------------------------------------------------------------------------
def half3(x):
    return np.asarray(x, f32).astype(np.float16).astype(f32)


MIPLEVELCOUNT = 5  # settings / texture.h


def uchar4_to_float4(v):  # getShadingData.h:22-26
    v = np.asarray(v, np.uint32)
    r = f32(1.0 / 256.0)
    return np.stack([(v & u32(255)).astype(f32) * r, ((v >> u32(8)) & u32(255)).astype(f32) * r,
                     ((v >> u32(16)) & u32(255)).astype(f32) * r, (v >> u32(24)).astype(f32) * r], -1).astype(f32)


def fetch_texel(tex, tcx, tcy, o, w, h):  # getShadingData.h:28-61, BILINEAR 1; o, w, h: per-lane arrays
    fx = ((np.maximum(tcx + f32(1000), f32(0)) * w.astype(f32)) - f32(0.5)).astype(f32)
    fy = ((np.maximum(tcy + f32(1000), f32(0)) * h.astype(f32)) - f32(0.5)).astype(f32)
    iu = fx.astype(np.int64) % w
    iv = fy.astype(np.int64) % h
    fu = (fx - np.floor(fx)).astype(f32)
    fv = (fy - np.floor(fy)).astype(f32)
    w0 = ((f32(1) - fu) * (f32(1) - fv)).astype(f32)
    w1 = (fu * (f32(1) - fv)).astype(f32)
    w2 = ((f32(1) - fu) * fv).astype(f32)
    w3 = (f32(1) - ((w0 + w1) + w2)).astype(f32)
    iu1, iv1 = (iu + 1) % w, (iv + 1) % h
    def px(i):
        i = np.minimum(i, len(tex["data"]) - 1)
        return tex["data"][i] if tex["float4"] else uchar4_to_float4(tex["data"][i])
    p0, p1, p2, p3 = px(o + iu + iv * w), px(o + iu1 + iv * w), px(o + iu + iv1 * w), px(o + iu1 + iv1 * w)
    return (((p0 * w0[:, None] + p1 * w1[:, None]) + p2 * w2[:, None]) + p3 * w3[:, None]).astype(f32)


def fetch_trilinear(tex, lam, tcx, tcy, width, height):  # getShadingData.h:63-98
    n = len(lam)
    ilam = np.trunc(lam).astype(np.int64)  # (int)lambda
    level0 = np.minimum(MIPLEVELCOUNT - 1, ilam)
    level1 = np.minimum(MIPLEVELCOUNT - 1, level0 + 1)
    f = (lam - np.floor(lam)).astype(f32)
    def select(level):  # `for (i = 0; i < level; i++)`: a level <= 0 leaves offset, width and height alone
        o = np.zeros(n, np.int64)
        w = np.full(n, width, np.int64)
        h = np.full(n, height, np.int64)
        for i in range(MIPLEVELCOUNT - 1):
            step = level > i
            o = np.where(step, o + w * h, o)
            w = np.where(step, w >> 1, w)
            h = np.where(step, h >> 1, h)
        return o, np.maximum(w, 1), np.maximum(h, 1)
    o0, w0, h0 = select(level0)
    o1, w1, h1 = select(level1)
    p0 = fetch_texel(tex, tcx, tcy, o0, w0, h0)
    p1 = fetch_texel(tex, tcx, tcy, o1, w1, h1)
    return ((f32(1) - f)[:, None] * p0 + f[:, None] * p1).astype(f32)


class PathTracer:
    def __init__(self, pkg, scene, W, H, blue_noise=None):
        self.W, self.H = W, H
        self.scene = scene
        self.blue_noise = blue_noise  # the 5 x 65536 table (BLUENOISE 1, settings.h:12) or None: the RandomFloat branches
        self.geo = Geometry(scene)
        a, p, s, d = scene.light_arrays()
        self.lights = Lights(list(a), list(p), list(s), list(d))
        mats, _ = pkg.scenes.pack_materials(scene.host_materials, scene.textures)
        self.mat_color = np.stack([half3(m["color"]) for m in scene.host_materials])
        self.mat_absorption = np.stack([half3(m["absorption"]) for m in scene.host_materials])
        self.mat_params = np.stack([np.asarray(m["parameters"], np.uint32) for m in mats])
        self.mat_flags = np.asarray([int(m["flags"]) for m in mats], np.uint32)
        self.mat_maps = [m["map"] for m in mats]  # 10 map descriptors per material (structs.h:98-115); addr = texture index
        self.textures = []
        for t in scene.textures:
            f4t = int(t["type"]) == 0  # TexelStorage: RGBA128 (float4) = 0, RGBA32 = 1
            self.textures.append(dict(float4=f4t, data=(np.asarray(t["data"], f32).reshape(-1, 4) if f4t else np.asarray(t["data"], np.uint32))))
        # Camera.cpp:80: spreadAngle = (FOV * pi / 180) / pixelCount.y
        self.spread_angle = f32(f32(scene.camera.FOV) * f32(np.pi) / f32(180)) / f32(H)
        pos, p1, p2, p3 = camera_view(scene.camera)
        self.pos, self.p1 = pos, p1
        self.right, self.up = (p2 - p1).astype(f32), (p3 - p1).astype(f32)
        self.aperture = f32(scene.camera.aperture)
        self.clamp = f32(scene.camera.clampValue)
        self.sky, self.sky_w, self.sky_h = scene.sky
        self.counts = []

    def primary(self, sample_index):  # Kernels.cu:383-426 (RandomFloat branch)
        W, H = self.W, self.H
        pid = np.arange(W * H, dtype=np.uint64)
        seed = wang_hash((pid * 16789 + sample_index * 1791) & 0xFFFFFFFF)
        sx, sy = (pid % W).astype(np.int64), (pid // W).astype(np.int64)
        if self.blue_noise is not None:  # Kernels.cu:391-394
            r0, r1, r2, r3 = (blue_noise_sampler(self.blue_noise, sx, sy, sample_index, k) for k in range(4))
        else:  # Kernels.cu:396-399
            r0, seed = random_float(seed)
            r1, seed = random_float(seed)
            r2, seed = random_float(seed)
            r3, seed = random_float(seed)
        blade = (r0 * f32(9)).astype(np.int32).astype(f32)
        r2 = ((r2 - blade * f32(1.0 / 9.0)) * f32(9.0)).astype(f32)
        po = f32(3.14159265359) / f32(4.5)
        x1, y1 = np.sin(blade * po).astype(f32), np.cos(blade * po).astype(f32)  # __sincosf(a, &x1, &y1)
        x2, y2 = np.sin((blade + f32(1)) * po).astype(f32), np.cos((blade + f32(1)) * po).astype(f32)
        flip = (r2 + r3) > 1
        r2, r3 = np.where(flip, f32(1) - r2, r2).astype(f32), np.where(flip, f32(1) - r3, r3).astype(f32)
        xr, yr = (x1 * r2 + x2 * r3).astype(f32), (y1 * r2 + y2 * r3).astype(f32)
        O = (self.pos[None] + self.aperture * (s3(xr, self.right[None]) + s3(yr, self.up[None]))).astype(f32)
        u = ((sx.astype(f32) + r0) * (f32(1) / f32(W))).astype(f32)
        v = ((sy.astype(f32) + r1) * (f32(1) / f32(H))).astype(f32)
        pix = ((self.p1[None] + s3(u, self.right[None])) + s3(v, self.up[None])).astype(f32)
        D = normalize((pix - O).astype(f32))
        return O, D, pid.astype(np.uint32)

    def sky_sample(self, D):  # Kernels.cu:593-600
        u = to_uint(f32(self.sky_w) * f32(0.5) * (f32(1) + np.arctan2(D[:, 0], -D[:, 2]).astype(f32) * INVPI))
        v = to_uint(f32(self.sky_h) * np.arccos(np.clip(D[:, 1], -1, 1)).astype(f32) * INVPI)
        idx = u.astype(np.int64) + v.astype(np.int64) * self.sky_w
        ok = idx < self.sky_w * self.sky_h
        out = np.zeros((len(D), 3), f32)
        out[ok] = self.sky[idx[ok]]
        return out

    def shading_data(self, D, bu, bv, inst, prim, t):  # getShadingData.h:100-217
        n = len(D)
        texu = np.zeros((n, 3), f32)
        texv = np.zeros((n, 3), f32)
        lod = np.zeros(n, f32)
        N = np.zeros((n, 3), f32)
        iN = np.zeros((n, 3), f32)
        matid = np.zeros(n, np.int64)
        area = np.zeros(n, f32)
        ltri = np.full(n, -1, np.int64)
        w = (f32(1) - bu - bv).astype(f32)
        for ii, g in enumerate(self.geo.inst):
            sel = inst == ii
            if not sel.any():
                continue
            tr = g["tris"][prim[sel]]
            Ng = np.stack([tr["Nx"], tr["Ny"], tr["Nz"]], -1).astype(f32)
            mid = tr["material"].astype(np.int64)
            smooth = ((self.mat_flags[mid] >> u32(11)) & u32(1)).astype(bool)
            iNs = normalize(((s3(bu[sel], tr["vN0"]) + s3(bv[sel], tr["vN1"])) + s3(w[sel], tr["vN2"])).astype(f32))
            iNl = np.where(smooth[:, None], iNs, Ng).astype(f32)
            M = g["nrm"]  # mat3 * vec3, column-major
            mul = lambda x: ((x[:, 0:1] * M[:, 0][None] + x[:, 1:2] * M[:, 1][None]) + x[:, 2:3] * M[:, 2][None]).astype(f32)  # noqa: E731
            N[sel] = normalize(mul(Ng))
            iN[sel] = normalize(mul(iNl))
            matid[sel] = mid
            area[sel] = tr["area"]
            ltri[sel] = tr["lightTriIdx"]
            texu[sel], texv[sel], lod[sel] = tr["u"], tr["v"], tr["LOD"]
        sd = SD(self.mat_color[matid], self.mat_absorption[matid], self.mat_params[matid])
        T, B = create_tangent_space(iN)
        alpha = np.zeros(n, bool)
        flag = lambda bit: ((self.mat_flags[matid] >> u32(bit)) & u32(1)).astype(bool)  # noqa: E731  structs.h:67-83
        has_diffuse = flag(2)
        if has_diffuse.any():
            color = sd.color.copy()
            tu = ((bu * texu[:, 0] + bv * texu[:, 1]) + w * texu[:, 2]).astype(f32)
            tv = ((bu * texv[:, 0] + bv * texv[:, 1]) + w * texv[:, 2]).astype(f32)
            coneWidth = (self.spread_angle * t).astype(f32)
            lam = (lod + np.log2(coneWidth * (f32(1) / np.abs(dot(-D, N)))).astype(f32)).astype(f32)  # eq. 26
            for mi in np.unique(matid[has_diffuse]):
                sel = np.nonzero(has_diffuse & (matid == mi))[0]
                maps = self.mat_maps[mi]
                def uv(k):
                    m = maps[k]
                    us, vs, uo, vo = (f32(m[x]) for x in ("uscale", "vscale", "uoffs", "voffs"))
                    return (us * (uo + tu[sel])).astype(f32), (vs * (vo + tv[sel])).astype(f32)
                def layer(k):  # FetchTexelTrilinear of map slot k
                    m = maps[k]
                    x, y = uv(k)
                    return fetch_trilinear(self.textures[int(m["addr"])], lam[sel], x, y, int(m["width"]), int(m["height"]))
                def nlayer(k):  # (FetchTexel(level 0) - 0.5) * 2
                    m = maps[k]
                    x, y = uv(k)
                    z = np.zeros(len(sel), np.int64)
                    tx = fetch_texel(self.textures[int(m["addr"])], x, y, z, z + int(m["width"]), z + int(m["height"]))
                    return ((tx[:, :3] - f32(0.5)) * f32(2.0)).astype(f32)
                fl = int(self.mat_flags[mi])
                texel = layer(0)
                a = np.zeros(len(sel), bool)
                if (fl >> 12) & 1:  # HasAlpha: the rest of the surface is not evaluated
                    a = texel[:, 3] < f32(0.5)
                alpha[sel] = a
                c = (color[sel] * texel[:, :3]).astype(f32)
                if (fl >> 9) & 1:   # Has2ndDiffuseMap: additive
                    c = (c + layer(1)[:, :3]).astype(f32)
                if (fl >> 10) & 1:  # Has3rdDiffuseMap
                    c = (c + layer(2)[:, :3]).astype(f32)
                if (fl >> 3) & 1:   # HasNormalMap (slots 3..5 of the map array)
                    sn = nlayer(3)
                    if (fl >> 7) & 1:
                        sn = (sn + nlayer(4)).astype(f32)
                    if (fl >> 8) & 1:  # the third layer reads the SECOND layer's descriptor (getShadingData.h:189-196)
                        sn = (sn + nlayer(4)).astype(f32)
                    sn = normalize(sn)
                    wn = normalize(((T[sel] * sn[:, 0:1] + B[sel] * sn[:, 1:2]) + iN[sel] * sn[:, 2:3]).astype(f32))  # tangentToWorld
                    iN[sel] = np.where(a[:, None], iN[sel], wn)
                c = (c * texel[:, :3]).astype(f32)  # :206, the second multiplication by the texel
                color[sel] = np.where(a[:, None], color[sel], c)
            sd.color = color
        return sd, N, iN, T, B, area, ltri, alpha

    def render_sample(self, acc, sample_index, record=None):
        """One call of CUDAContext::render_frame.  acc: (W*H, 4) accumulator.  samplesTaken == sample_index."""
        W, H = self.W, self.H
        O, D, pid = self.primary(sample_index)
        n = len(O)
        st = dict(O=O, D=D, T=np.ones((n, 3), f32), pdf=np.ones(n, f32), flags=np.ones(n, np.uint32), pid=pid,
                  lastN=np.zeros(n, np.uint32))
        counts = dict(ext=[n], shadow_emitted=[], shadow_traced=[])
        pathLength = 0
        pending_shadow = None
        while True:
            t, inst, prim, bu, bv = self.geo.closest(st["O"], st["D"], f32(1e-5))
            nxt, shadow = self.shade(acc, st, t, inst, prim, bu, bv, pathLength, sample_index)
            counts["shadow_emitted"].append(len(shadow["O"]))
            active = len(nxt["O"])
            if record is not None and pathLength == 0:
                record.update(t0=t, inst0=inst, prim0=prim)
            # Context.cpp:109: while (activePaths > 0 && pathLength < MAX_PATH_LENGTH) { shadow; extend; shade }
            if not (active > 0 and pathLength < MAX_PATH_LENGTH):
                break  # the connections emitted by the LAST shade call are never traced
            pathLength += 1
            if len(shadow["O"]):
                occ = self.geo.occluded(shadow["O"], shadow["D"], GEO_EPS, shadow["tmax"])
                counts["shadow_traced"].append(len(shadow["O"]))
                np.add.at(acc, shadow["pid"][~occ], np.concatenate([shadow["E"][~occ], np.ones((int((~occ).sum()), 1), f32)], 1))
            else:
                counts["shadow_traced"].append(0)
            counts["ext"].append(active)
            st = nxt
        self.counts.append(counts)
        return counts

    def shade(self, acc, st, t, inst, prim, bu, bv, pathLength, samplesTaken):  # Kernels.cu:571-794
        O, D, T, bsdfPdf, flags, pid, lastNp = st["O"], st["D"], st["T"], st["pdf"], st["flags"], st["pid"], st["lastN"]
        n = len(O)
        empty = dict(O=np.zeros((0, 3), f32), D=np.zeros((0, 3), f32), T=np.zeros((0, 3), f32), pdf=np.zeros(0, f32),
                     flags=np.zeros(0, np.uint32), pid=np.zeros(0, np.uint32), lastN=np.zeros(0, np.uint32))
        # ---- miss: sky
        miss = prim < 0
        if miss.any():
            c = (s3(f32(1) / bsdfPdf[miss], T[miss]) * self.sky_sample(D[miss])).astype(f32)
            ok = ~np.isnan(c).any(1)
            c = clamp_intensity(c, self.clamp)
            np.add.at(acc, pid[miss][ok], np.concatenate([c[ok], np.zeros((int(ok.sum()), 1), f32)], 1))
        h = ~miss
        if not h.any():
            return empty, dict(O=np.zeros((0, 3), f32), D=np.zeros((0, 3), f32), tmax=np.zeros(0, f32), E=np.zeros((0, 3), f32), pid=np.zeros(0, np.uint32))
        O, D, T, bsdfPdf, flags, pid, lastNp = O[h], D[h], T[h], bsdfPdf[h], flags[h], pid[h], lastNp[h]
        t, inst, prim, bu, bv = t[h], inst[h], prim[h], bu[h], bv[h]
        I = (O + s3(t, D)).astype(f32)
        sd, N, iN, Tg, Bt, area, ltri, alpha = self.shading_data(D, bu, bv, inst, prim, t)
        # ---- alpha cut-out (Kernels.cu:633-647): the path goes on behind the surface with its state untouched.  (The
        # reference stores that state into the wrong buffers — its own TODO says the branch is broken; what it means is kept.)
        through = None
        if alpha.any():
            a = alpha & ~np.isnan(T).any(1)
            if pathLength < MAX_PATH_LENGTH and a.any():
                through = dict(O=(I[a] + D[a] * GEO_EPS).astype(f32), D=D[a], T=T[a], pdf=bsdfPdf[a], flags=flags[a], pid=pid[a], lastN=lastNp[a])
            k = ~alpha
            O, D, T, bsdfPdf, flags, pid, lastNp = O[k], D[k], T[k], bsdfPdf[k], flags[k], pid[k], lastNp[k]
            t, inst, prim, bu, bv, I = t[k], inst[k], prim[k], bu[k], bv[k], I[k]
            N, iN, Tg, Bt, area, ltri = N[k], iN[k], Tg[k], Bt[k], area[k], ltri[k]
            sd = sd.take(k)
        # ---- emissive: terminate (Kernels.cu:650-692)
        em = (sd.color > 1).any(1)
        if em.any():
            e = em
            DdotNL = -dot(D[e], N[e])
            col = sd.color[e]
            if pathLength == 0:
                c = col.copy()
                drop = np.zeros(len(col), bool)
            else:
                spec = (flags[e] & u32(1)).astype(bool)
                c_spec = (T[e] * col * (f32(1) / bsdfPdf[e])[:, None]).astype(f32)
                lastN = unpack_normal(lastNp[e])
                lightPdf = ((t[e] * t[e]) / (-dot(D[e], N[e]) * area[e])).astype(f32)  # CalculateLightPDF
                pick = light_pick_prob(self.lights, ltri[e], O[e], lastN, I[e])
                den = (bsdfPdf[e] + lightPdf * pick).astype(f32)
                drop = ~spec & (den <= 0) & (DdotNL > 0)
                c_mis = (T[e] * col * (f32(1) / den)[:, None]).astype(f32)
                c = np.where(spec[:, None], c_spec, c_mis).astype(f32)
            c = np.where((DdotNL > 0)[:, None], c, f32(0)).astype(f32)
            c = np.where(np.isnan(c).any(1)[:, None], f32(0), c).astype(f32)
            c = clamp_intensity(c, self.clamp)
            keep = ~drop
            np.add.at(acc, pid[e][keep], np.concatenate([c[keep], np.zeros((int(keep.sum()), 1), f32)], 1))
        s = ~em
        O, D, T, bsdfPdf, flags, pid = O[s], D[s], T[s], bsdfPdf[s], flags[s], pid[s]
        t, I, N, iN, Tg, Bt = t[s], I[s], N[s], iN[s], Tg[s], Bt[s]
        sd = sd.take(s)
        m = len(O)
        flags = np.where(sd.ROUGHNESS < MIN_ROUGHNESS, flags | u32(1), flags & ~u32(1)).astype(np.uint32)
        seed = wang_hash((pid.astype(np.uint64) * 16789 + samplesTaken * 1791 + pathLength * 720898027) & 0xFFFFFFFF)
        flip = np.where(dot(D, N) > 0, f32(-1), f32(1)).astype(f32)
        N = s3(flip, N)
        iN = s3(flip, iN)
        T = (T * (f32(1) / bsdfPdf)[:, None]).astype(f32)
        wo = (D * f32(-1)).astype(f32)
        shadow = dict(O=np.zeros((0, 3), f32), D=np.zeros((0, 3), f32), tmax=np.zeros(0, f32), E=np.zeros((0, 3), f32), pid=np.zeros(0, np.uint32))
        nee = (flags & u32(1)) == 0
        if self.lights.count() > 0 and nee.any():
            r0 = np.zeros(m, f32)
            r1 = np.zeros(m, f32)
            if self.blue_noise is not None and samplesTaken < 256:  # Kernels.cu:712-719: the seed is NOT advanced
                px, py = (pid % np.uint32(self.W)).astype(np.int64), (pid // np.uint32(self.W)).astype(np.int64)
                r0 = blue_noise_sampler(self.blue_noise, px, py, samplesTaken, 4)
                r1 = blue_noise_sampler(self.blue_noise, px, py, samplesTaken, 5)
            else:
                a, seed_n = random_float(seed[nee])
                b, seed_n = random_float(seed_n)
                r0[nee], r1[nee] = a, b
                seed = seed.copy()
                seed[nee] = seed_n
            P, pickProb, lightPdf, lightColor = random_point_on_light(self.lights, r0, r1, I, iN)
            L = (P - I).astype(f32)
            dist = length(L)
            L = s3(f32(1) / dist, L)
            NdotL = dot(L, iN)
            sampled = bsdf_eval(sd, iN, wo, L, np.zeros(m, f32), np.zeros(m, bool))
            shadowPdf = bsdf_pdf(sd, iN, wo, L)
            contribution = (T * sampled * lightColor * (NdotL / (shadowPdf + lightPdf * pickProb))[:, None]).astype(f32)
            contribution = clamp_intensity(contribution, self.clamp)
            emit = nee & (NdotL > 0) & (lightPdf > 0) & (shadowPdf > 0) & ~np.isnan(contribution).any(1)
            shadow = dict(O=(I[emit] + N[emit] * GEO_EPS).astype(f32), D=L[emit], tmax=(dist[emit] - f32(2.0) * GEO_EPS).astype(f32),
                          E=contribution[emit], pid=pid[emit])
        if pathLength >= MAX_PATH_LENGTH:
            return empty, shadow
        r3, seed = random_float(seed)
        r4, seed = random_float(seed)
        R, newPdf = bsdf_sample(sd, Tg, Bt, iN, wo, r3, r4, np.zeros(m, f32))
        bsdf = bsdf_eval(sd, iN, wo, R, t, flip < 0)
        # throughput * 1.0f / SurvivalProbability(throughput) * bsdf * abs(dot(iN, R))
        T = (((T * f32(1.0)) / survival_probability(T)[:, None]) * bsdf * np.abs(dot(iN, R))[:, None]).astype(f32)
        ok = ~((newPdf < f32(1e-6)) | np.isnan(newPdf) | (T < 0).any(1))
        nxt = dict(O=(I[ok] + N[ok] * GEO_EPS).astype(f32), D=R[ok], T=T[ok], pdf=newPdf[ok], flags=flags[ok], pid=pid[ok],
                   lastN=pack_normal(iN[ok]))
        if through is not None:
            nxt = {k: np.concatenate([nxt[k], through[k]]) for k in nxt}
        return nxt, shadow


# ----------------------------------------------------------------------------------------------------------------------
# known-answer tables
# ----------------------------------------------------------------------------------------------------------------------
def unit_vectors(rng, n):
    v = rng.normal(size=(n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(f32)


def make_kat(pkg):
    rng = np.random.default_rng(20260928)
    n = 512
    out = {}
    # materials: a spread over all parameters, packed exactly like rfw::system does (8-bit)
    def pack4(a, b, c, d):
        q = lambda x: (np.asarray(x, f32) * f32(255.0)).astype(np.uint32)  # noqa: E731  TOCHAR
        return (q(a) + (q(b) << u32(8)) + (q(c) << u32(16)) + (q(d) << u32(24))).astype(np.uint32)
    U = lambda: rng.uniform(0, 1, n).astype(f32)  # noqa: E731
    Z = lambda p: np.where(rng.uniform(0, 1, n) < p, 0.0, rng.uniform(0, 1, n)).astype(f32)  # noqa: E731  often exactly 0
    metallic, subsurface, specular, roughness = Z(0.4), Z(0.5), U(), U()
    spectint, clearcoat, ccgloss, transmission, eta = U(), Z(0.5), U(), Z(0.5), rng.uniform(0.5, 1.0, n).astype(f32)
    params = np.stack([pack4(metallic, subsurface, specular, roughness), pack4(spectint, U(), U(), U()),
                       pack4(clearcoat, ccgloss, transmission, eta * f32(0.5)), np.zeros(n, np.uint32)], 1)
    color = rng.uniform(0.02, 0.98, (n, 3)).astype(np.float16).astype(f32)
    absorption = rng.uniform(0.0, 0.8, (n, 3)).astype(np.float16).astype(f32)
    sd = SD(color, absorption, params)
    N = unit_vectors(rng, n)
    # wo in the upper hemisphere of N; wi anywhere (both hemispheres are branches of the BSDF)
    wo = unit_vectors(rng, n)
    wo = np.where((dot(wo, N) < 0)[:, None], -wo, wo).astype(f32)
    wi = unit_vectors(rng, n)
    t = rng.uniform(0.1, 10.0, n).astype(f32)
    back = rng.uniform(0, 1, n) < 0.3
    out.update(bsdf_color=color, bsdf_absorption=absorption, bsdf_params=params, bsdf_N=N, bsdf_wo=wo, bsdf_wi=wi, bsdf_t=t,
               bsdf_backfacing=back.astype(np.uint32))
    out["bsdf_eval"] = bsdf_eval(sd, N, wo, wi, t, back)
    out["bsdf_pdf"] = bsdf_pdf(sd, N, wo, wi)
    r3, r4 = U(), U()
    T, B = create_tangent_space(N)
    swi, spdf = bsdf_sample(sd, T, B, N, wo, r3, r4, np.zeros(n, f32))
    out.update(sample_r3=r3, sample_r4=r4, sample_wi=swi, sample_pdf=spdf, tangent_T=T, tangent_B=B)
    # tools
    out["pack_in"] = N
    out["pack_out"] = pack_normal(N)
    out["unpack_out"] = unpack_normal(out["pack_out"])
    rb = np.concatenate([np.array([0.0, 0.25, 0.5, 0.75, 0.999999], f32), U()[:251]])
    out["bary_r0"] = rb
    out["bary_out"] = random_barycentrics(rb)
    seeds = rng.integers(0, 2 ** 32, 64, dtype=np.uint64)
    out["hash_in"] = seeds.astype(np.uint32)
    out["wang_hash"] = wang_hash(seeds).astype(np.uint32)
    fl, st = random_float(wang_hash(seeds))
    out["random_float"] = fl
    out["random_state"] = st.astype(np.uint32)
    # lights: the "lights" golden scene's light set (2 area triangles + point + spot + directional)
    scene = golden_scenes.cornell_lights(pkg, 96, 64)
    a, p, s_, d = scene.light_arrays()
    lt = Lights(list(a), list(p), list(s_), list(d))
    I = rng.uniform(-4.5, 4.5, (n, 3)).astype(f32)
    I[:, 1] = rng.uniform(0.1, 9.0, n).astype(f32)
    Nl = unit_vectors(rng, n)
    lr0, lr1 = U(), U()
    P, pick, lpdf, lcol = random_point_on_light(lt, lr0, lr1, I, Nl)
    out.update(light_I=I, light_N=Nl, light_r0=lr0, light_r1=lr1, light_P=P, light_pick=pick, light_pdf=lpdf, light_color=lcol)
    Oq = rng.uniform(-4.5, 4.5, (n, 3)).astype(f32)
    Oq[:, 1] = rng.uniform(0.1, 9.0, n).astype(f32)
    lidx = rng.integers(0, len(a), n)
    out.update(pickprob_idx=lidx.astype(np.int32), pickprob_O=Oq, pickprob=light_pick_prob(lt, lidx, Oq, Nl, I))
    return out


def reference_blue_noise_table():
    """The reference's createBlueNoiseBuffer() output through oracle/_ref/libbluenoise.so (a build of blue_noise.h itself,
    oracle/Makefile target `ref`), or None when that library was not built."""
    import ctypes
    lib_path = os.path.join(ROOT, "oracle", "_ref", "libbluenoise.so")
    if not os.path.exists(lib_path):
        return None
    lib = ctypes.CDLL(lib_path)
    lib.rfw_ref_blue_noise_table.restype = ctypes.POINTER(ctypes.c_uint32)
    return np.ctypeslib.as_array(lib.rfw_ref_blue_noise_table(), shape=(5 * 65536,)).copy()


def blue_noise_kat(out):
    """Sampler known answers on the REAL table.  Only outputs are stored, never the table."""
    import zlib
    table = reference_blue_noise_table()
    if table is None:
        raise SystemExit("oracle/_ref/libbluenoise.so not built: run `make -C oracle ref` first")
    # crc32 of the three byte tables
    crc = [zlib.crc32(table[a:b].astype(np.uint8).tobytes()) & 0xFFFFFFFF for a, b in ((0, 65536), (65536, 65536 + 131072), (3 * 65536, 3 * 65536 + 131072))]
    assert crc == [0xd87313bd, 0x12b18559, 0x24c59e1f], [hex(c) for c in crc]
    rng = np.random.default_rng(99)
    n = 4096
    x, y = rng.integers(0, 1920, n), rng.integers(0, 1080, n)
    si, dim = rng.integers(0, 300, n), rng.integers(0, 8, n)
    out.update(bn_x=x.astype(np.int32), bn_y=y.astype(np.int32), bn_sample=si.astype(np.int32), bn_dim=dim.astype(np.int32),
               bn_value=blue_noise_sampler(table, x, y, si, dim), bn_crc32=np.asarray(crc, np.uint32),
               bn_table_sum=np.uint64(table.astype(np.uint64).sum()))


def main():
    """python make_golden_pt.py [name ...]: (re)generate the named fixtures only; without arguments, all of them."""
    pkg = load_package()
    only = set(sys.argv[1:])
    if not only or "pt_kat" in only:
        kat = make_kat(pkg)
        blue_noise_kat(kat)
        np.savez_compressed(os.path.join(HERE, "pt_kat.npz"), **kat)
        print("pt_kat.npz:", len(kat), "arrays")
    W, H, SPP = 96, 64, 4
    table = reference_blue_noise_table() if (not only or "pt_cornell96x64_bluenoise" in only) else None
    for name, scene, bn in (("pt_cornell96x64", golden_scenes.cornell_pt(pkg, W, H), None),
                            ("pt_lights96x64", golden_scenes.cornell_lights(pkg, W, H), None),
                            ("pt_cornell96x64_bluenoise", golden_scenes.cornell_pt(pkg, W, H), table),
                            ("pt_terrain96x64", golden_scenes.terrain_small(pkg, W, H), None),
                            ("pt_cards96x64", golden_scenes.cards_pt(pkg, W, H), None),
                            ("pt_lens96x64", golden_scenes.cornell_lens(pkg, W, H), None)):
        if only and name not in only:
            continue
        pt = PathTracer(pkg, scene, W, H, blue_noise=bn)
        acc = np.zeros((W * H, 4), f32)
        per_sample = []
        rec = {}
        for s in range(SPP):
            before = acc.copy()
            c = pt.render_sample(acc, s, rec if s == 0 else None)
            per_sample.append((acc - before)[:, :3].reshape(H, W, 3))
            print(name, "sample", s, c)
        img = (acc[:, :3] * (f32(1) / f32(SPP))).reshape(H, W, 3)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), image=img, sample0=per_sample[0],
                            ext=np.asarray([c["ext"] + [0] * (3 - len(c["ext"])) for c in pt.counts], np.int64),
                            shadow_emitted=np.asarray([c["shadow_emitted"] + [0] * (3 - len(c["shadow_emitted"])) for c in pt.counts], np.int64),
                            shadow_traced=np.asarray([c["shadow_traced"] + [0] * (2 - len(c["shadow_traced"])) for c in pt.counts], np.int64),
                            t0=rec["t0"].reshape(H, W), prim0=rec["prim0"].reshape(H, W), inst0=rec["inst0"].reshape(H, W), spp=SPP)
        print(name, "mean", float(img.mean()))


if __name__ == "__main__":
    main()
