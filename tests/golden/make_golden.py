#!/usr/bin/env python3
"""Generate tests/golden/*.npz — an INDEPENDENT float32 numpy brute-force rendering of the Cornell configuration
(BASELINE.json config 1), used to pin the C oracle (oracle/rfw_oracle.c).  Run in the development container:

    python tests/golden/make_golden.py

Independent means: no BVH (every ray is tested against every triangle), instancing by transforming the VERTICES to
world space (the oracle transforms the RAY into object space), shading and the xor128 jitter stream re-derived here
from the behavioural spec in SURVEY.md §9.3 (EmbreeRT/src/Context.cpp:104-300, Ray.cpp:16-47, Camera.cpp:74-88,
utils/xor128.h:20-27).  The reference itself cannot be built or imported here, so these are not
reference outputs: the oracle stays "parity unpinned" with respect to the reference, and pinned with respect to this
second implementation.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

f32 = np.float32


def xor128_stream(n, state=(123456789, 362436069, 521288629, 88675123)):
    x, y, z, w = state
    out = np.empty(n, np.uint32)
    M = 0xFFFFFFFF
    for i in range(n):
        t = (x ^ (x << 11)) & M
        x, y, z = y, z, w
        w = (w ^ (w >> 19) ^ (t ^ (t >> 8))) & M
        out[i] = w
    return out, (x, y, z, w)


def camera_view(cam):
    pos = np.asarray(cam.position, f32)
    d = np.asarray(cam.direction, f32)
    right = np.cross(d, np.array([0, 1, 0], f32)).astype(f32)
    right = (right * (f32(1) / np.sqrt(np.dot(right, right), dtype=f32))).astype(f32)
    up = np.cross(right, d).astype(f32)
    s = f32(np.tan(f32(cam.FOV) / f32(2) / (f32(180) / f32(np.pi)), dtype=f32))
    fd, asp = f32(cam.focalDistance), f32(cam.aspectRatio)
    c = pos + fd * d
    h = ((s * right) * fd) * asp
    v = (s * fd) * up
    return pos, (c - h + v).astype(f32), (c + h + v).astype(f32), (c - h - v).astype(f32)


def world_triangles(scene):
    tris = []
    for ii, inst in enumerate(scene.instances):
        m = scene.meshes[inst["mesh"]]
        M = inst["transform"]
        Nm = np.linalg.inv(M[:3, :3]).T
        v = m["vertices"][:, :3].astype(np.float64) @ M[:3, :3].T + M[:3, 3]
        idx = m["indices"] if m["indices"] is not None else np.arange(len(v)).reshape(-1, 3)
        for pi, (a, b, c) in enumerate(idx):
            t = m["triangles"][pi]
            vn = [(Nm @ t[k].astype(np.float64)) for k in ("vN0", "vN1", "vN2")]
            tris.append(dict(p=[v[a].astype(f32), v[b].astype(f32), v[c].astype(f32)], inst=ii, prim=pi, vn=vn,
                             material=int(t["material"]), tu=np.asarray(t["u"], f32), tv=np.asarray(t["v"], f32)))
    return tris


def intersect_all(O, D, tris, tmin, tmax0):
    """Closest hit of rays (N,3) against all triangles, float32 Möller–Trumbore with the reference's rejections."""
    n = len(O)
    t = np.full(n, tmax0, f32)
    prim = np.full(n, -1, np.int32)
    which = np.full(n, -1, np.int32)
    uu = np.zeros(n, f32)
    vv = np.zeros(n, f32)
    for k, tr in enumerate(tris):
        p0, p1, p2 = tr["p"]
        e1, e2 = (p1 - p0).astype(f32), (p2 - p0).astype(f32)
        h = np.cross(D, e2).astype(f32)
        a = (h @ e1).astype(f32)
        ok = ~((a > f32(-1e-6)) & (a < f32(1e-6)))
        with np.errstate(divide="ignore", invalid="ignore"):
            f = (f32(1) / a).astype(f32)
            s = (O - p0).astype(f32)
            u = (f * np.einsum("ij,ij->i", s, h)).astype(f32)
            q = np.cross(s, e1).astype(f32)
            v = (f * np.einsum("ij,ij->i", D, q)).astype(f32)
            tt = (f * (q @ e2)).astype(f32)
        ok &= ~((u < 0) | (u > 1)) & ~((v < 0) | (u + v > 1)) & (tt > tmin) & (t > tt)
        t[ok], prim[ok], which[ok], uu[ok], vv[ok] = tt[ok], tr["prim"], k, u[ok], v[ok]
    return t, which, uu, vv


def occluded(O, D, tris, tmin, tmax):
    occ = np.zeros(len(O), bool)
    for tr in tris:
        p0, p1, p2 = tr["p"]
        e1, e2 = (p1 - p0).astype(f32), (p2 - p0).astype(f32)
        h = np.cross(D, e2).astype(f32)
        a = (h @ e1).astype(f32)
        ok = ~((a > f32(-1e-6)) & (a < f32(1e-6)))
        with np.errstate(divide="ignore", invalid="ignore"):
            f = (f32(1) / a).astype(f32)
            s = (O - p0).astype(f32)
            u = (f * np.einsum("ij,ij->i", s, h)).astype(f32)
            q = np.cross(s, e1).astype(f32)
            v = (f * np.einsum("ij,ij->i", D, q)).astype(f32)
            tt = (f * (q @ e2)).astype(f32)
        ok &= ~((u < 0) | (u > 1)) & ~((v < 0) | (u + v > 1)) & (tt > tmin) & (tmax > tt)
        occ |= ok
    return occ


def half(x):
    return f32(np.float16(x))


def diffuse_map(scene, tr, color, b0, b1, b2):
    """retrieve_material's texture lookup (EmbreeRT/src/Context.cpp:432-472): nearest texel at t * (size - 1), wrap by fmod,
    and the FLOAT4 case falling through into the UINT case (no break), which reads the same texel index out of the float
    data reinterpreted as 32-bit words."""
    m = scene.host_materials[tr["material"]]
    n = len(b0)
    out = np.broadcast_to(color, (n, 3)).astype(f32)
    if m.get("texture", -1) < 0:
        return out
    tex = scene.textures[m["texture"]]
    tu = ((b0 * tr["tu"][0] + b1 * tr["tu"][1]) + b2 * tr["tu"][2]).astype(f32)
    tv = ((b0 * tr["tv"][0] + b1 * tr["tv"][1]) + b2 * tr["tv"][2]).astype(f32)
    u = ((tu + half(m["uvoffset"][0])) * half(m["uvscale"][0])).astype(f32)
    v = ((tv + half(m["uvoffset"][1])) * half(m["uvscale"][1])).astype(f32)
    tx, ty = np.fmod(u, f32(1)).astype(f32), np.fmod(v, f32(1)).astype(f32)
    tx = np.where(tx < 0, f32(1) + tx, tx).astype(f32)
    ty = np.where(ty < 0, f32(1) + ty, ty).astype(f32)
    w, h = int(tex["width"]), int(tex["height"])
    ix = (tx * f32(w - 1)).astype(np.uint32).astype(np.int64)
    iy = (ty * f32(h - 1)).astype(np.uint32).astype(np.int64)
    tid = iy * w + ix
    if int(tex["type"]) == 0:  # TextureData::FLOAT4
        data = np.asarray(tex["data"], f32).reshape(-1, 4)
        out = (out * data[tid][:, :3]).astype(f32)
        words = np.asarray(tex["data"], f32).reshape(-1).view(np.uint32)  # ... and no break: falls into case UINT
    else:
        words = np.asarray(tex["data"], np.uint32)
    tc = words[tid]
    rgb = np.stack([tc & np.uint32(255), (tc >> np.uint32(8)) & np.uint32(255), (tc >> np.uint32(16)) & np.uint32(255)], -1).astype(f32)
    return ((out * f32(1.0 / 256.0)) * rgb).astype(f32)


def render(scene, W, H, jitter):
    cam = scene.camera
    pos, p1, p2, p3 = camera_view(cam)
    right, up = (p2 - p1).astype(f32), (p3 - p1).astype(f32)
    r0 = np.full((H, W), 0.5, f32)
    r1 = np.full((H, W), 0.5, f32)
    if jitter:
        npx, npy = W // 4, H // 2
        draws, _ = xor128_stream(npx * npy * 32)
        rnd = (draws.astype(f32) * f32(2.3283064365387e-10)).reshape(npy, npx, 4, 8)
        for j in range(8):
            r0[(j >> 2)::2, (j & 3)::4] = rnd[:, :, 0, j]
            r1[(j >> 2)::2, (j & 3)::4] = rnd[:, :, 1, j]
    ys, xs = np.mgrid[0:H, 0:W]
    u = ((xs.astype(f32) + r0) * (f32(1) / f32(W))).astype(f32)
    v = ((ys.astype(f32) + r1) * (f32(1) / f32(H))).astype(f32)
    pix = p1 + (u[..., None] * right + v[..., None] * up)
    org = np.broadcast_to(pos, pix.shape).astype(f32)
    if f32(cam.aperture) != 0:
        # Ray::generateFromView (Ray.cpp:16-47), the scalar form: nine-blade aperture, r2 / r3 = draws 16..23 / 24..31 of the packet
        r2 = np.full((H, W), 0.5, f32)
        r3 = np.full((H, W), 0.5, f32)
        if jitter:
            for j in range(8):
                r2[(j >> 2)::2, (j & 3)::4] = rnd[:, :, 2, j]
                r3[(j >> 2)::2, (j & 3)::4] = rnd[:, :, 3, j]
        blade = (r0 * f32(9)).astype(np.int32).astype(f32)
        r2 = ((r2 - blade * f32(1.0 / 9.0)) * f32(9.0)).astype(f32)
        po = f32(3.14159265359) / f32(4.5)
        x1, y1 = np.cos(blade * po).astype(f32), np.sin(blade * po).astype(f32)
        x2, y2 = np.cos((blade + f32(1)) * po).astype(f32), np.sin((blade + f32(1)) * po).astype(f32)
        flip = (r2 + r3) > 1
        r2, r3 = np.where(flip, f32(1) - r2, r2).astype(f32), np.where(flip, f32(1) - r3, r3).astype(f32)
        xr, yr = (x1 * r2 + x2 * r3).astype(f32), (y1 * r2 + y2 * r3).astype(f32)
        org = (pos + f32(cam.aperture) * (xr[..., None] * right + yr[..., None] * up)).astype(f32)
    d = (pix - org).astype(f32).reshape(-1, 3)
    l2 = (d[:, 0] * d[:, 0]).astype(f32)
    l2 = (d[:, 1] * d[:, 1] + l2).astype(f32)
    l2 = (d[:, 2] * d[:, 2] + l2).astype(f32)
    D = (d * (f32(1) / np.sqrt(l2, dtype=f32))[:, None]).astype(f32)
    O = org.reshape(-1, 3).astype(f32)
    tris = world_triangles(scene)
    t, which, bu, bv = intersect_all(O, D, tris, f32(1e-5), f32(1e34))
    img = np.zeros((H * W, 4), f32)
    miss = which < 0
    # sky (Context.cpp:187-196)
    pixs, sw, sh = scene.sky
    ux = (f32(0.5) * (f32(1) + np.arctan2(D[:, 0], -D[:, 2]).astype(f32) * f32(1 / np.pi))).astype(f32)
    uy = (np.arccos(np.clip(D[:, 1], -1, 1)).astype(f32) * f32(1 / np.pi)).astype(f32)
    px = np.minimum((ux * f32(sw - 1)).astype(np.uint32), sw - 1)
    py = np.minimum((uy * f32(sh - 1)).astype(np.uint32), sh - 1)
    img[miss, :3] = pixs[py[miss] * sw + px[miss]]
    hit = ~miss
    P = (O + D * t[:, None]).astype(f32)
    colors = np.array([np.asarray(m["color"], f32).astype(np.float16).astype(f32) for m in scene.host_materials])
    iN = np.zeros_like(P)
    col = np.zeros_like(P)
    for k, tr in enumerate(tris):
        sel = which == k
        if not sel.any():
            continue
        b0 = (f32(1) - bu[sel] - bv[sel]).astype(f32)
        n = b0[:, None] * tr["vn"][0] + bu[sel][:, None] * tr["vn"][1] + bv[sel][:, None] * tr["vn"][2]
        iN[sel] = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(f32)
        col[sel] = diffuse_map(scene, tr, colors[tr["material"]], b0, bu[sel], bv[sel])
    contrib = np.full_like(P, 0.1)
    area, point, _, _ = scene.light_arrays()
    for l in area:
        L = (l["position"] - P).astype(f32)
        sq = np.einsum("ij,ij->i", L, L).astype(f32)
        dist = np.sqrt(sq, dtype=f32)
        L = (L / dist[:, None]).astype(f32)
        ndl = np.einsum("ij,ij->i", iN, L).astype(f32)
        lndl = (-(L @ l["normal"])).astype(f32)
        cand = hit & (ndl > 0) & (lndl > 0)
        occ = np.ones(len(P), bool)
        occ[cand] = occluded(P[cand], L[cand], tris, f32(1e-4), dist[cand])
        lit = cand & ~occ
        contrib[lit] += ((l["radiance"] * l["area"])[None, :] / sq[lit, None] * ndl[lit, None] * lndl[lit, None]).astype(f32)
    for l in point:
        L = (l["position"] - P).astype(f32)
        sq = np.einsum("ij,ij->i", L, L).astype(f32)
        dist = np.sqrt(sq, dtype=f32)
        L = (L / dist[:, None]).astype(f32)
        ndl = np.einsum("ij,ij->i", iN, L).astype(f32)
        cand = hit & (ndl > 0)
        occ = np.ones(len(P), bool)
        occ[cand] = occluded(P[cand], L[cand], tris, f32(1e-4), dist[cand])
        lit = cand & ~occ
        contrib[lit] += (l["radiance"][None, :] / sq[lit, None] * ndl[lit, None]).astype(f32)
    shaded = (col * contrib).astype(f32)
    emissive = (col > 1).any(1)  # Context.cpp:217-221: a colour above 1 is written as it is
    shaded[emissive] = col[emissive]
    img[hit, :3] = shaded[hit]
    img[hit, 3] = 1.0
    prim = np.array([tris[k]["prim"] if k >= 0 else -1 for k in which], np.int32)
    inst = np.array([tris[k]["inst"] if k >= 0 else -1 for k in which], np.int32)
    shp = (H, W)
    return dict(image=img.reshape(H, W, 4), t=t.reshape(shp), prim=prim.reshape(shp), inst=inst.reshape(shp),
                u=bu.reshape(shp), v=bv.reshape(shp))


def main():
    pkg = load_package()
    out_dir = os.path.dirname(os.path.abspath(__file__))
    W, H = 96, 64  # wider than the box so the fixture also covers sky lookups
    scene = pkg.scenes.cornell(W, H)
    sys.path.insert(0, out_dir)
    import golden_scenes
    for name, jitter in (("cornell96x64_center", False), ("cornell96x64_xor128", True), ("cards96x64_center", False),
                         ("lens96x64_xor128", True)):
        if sys.argv[1:] and name not in sys.argv[1:]:
            continue
        if name.startswith("cards"):
            scene = golden_scenes.cards_parity(pkg, W, H)
        if name.startswith("lens"):
            scene = golden_scenes.cornell_lens_parity(pkg, W, H)
        r = render(scene, W, H, jitter)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **r)
        print(name, "mean", float(r["image"][..., :3].mean()), "hit fraction", float((r["prim"] >= 0).mean()))
    if sys.argv[1:] and "rng_kat" not in sys.argv[1:]:
        return
    # known answers of the integer generators (hand-derivable from xor128.h:20-27 / tools.h:218-235)
    draws, state = xor128_stream(1000)
    np.savez_compressed(os.path.join(out_dir, "rng_kat.npz"), xor128_first8=draws[:8], xor128_state_after_1000=np.array(state, np.uint32),
                        xor128_draw_1000=draws[999])


if __name__ == "__main__":
    main()
