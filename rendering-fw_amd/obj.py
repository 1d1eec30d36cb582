"""Wavefront OBJ / MTL front end for the rendercore.

The reference reads OBJ through assimp — a third-party library, unpinned — with the post-process steps
`aiProcess_GenSmoothNormals | aiProcess_JoinIdenticalVertices | aiProcess_Triangulate | ...`
(RFW/system/src/rfw/geometry/assimp/object.cpp:88-91, :379-382) and hands one indexed `rfw::Mesh` per assimp mesh to the
core (object.cpp:1105-1123): assimp splits an object wherever the material changes.  This reader restates those semantics:

  * one indexed mesh per material (`usemtl`), faces triangulated as fans, negative (relative) indices resolved;
  * vertices joined per unique (position, texcoord, normal) index triple (JoinIdenticalVertices);
  * normals from the file when every corner of the mesh has one, else smooth normals generated per POSITION: the sum of the
    incident face normals, area-weighted (GenSmoothNormals; assimp's exact weighting depends on its version);
  * materials from the .mtl: Kd -> colour, Ns -> roughness = sqrt(2 / (Ns + 2)), d / Tr -> ignored, map_Kd is recorded by name
    only (no image decoder in this environment); a face before any `usemtl` gets a default material.

Smoothing groups (`s`) do not change the result: with vn records the file's normals are used as they are, without them
GenSmoothNormals smooths over every shared position regardless of group.
"""
import math
import os

import numpy as np

from . import scenes


def parse_mtl(path):
    mats, cur = {}, None
    if not os.path.exists(path):
        return mats
    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "newmtl":
                cur = mats.setdefault(" ".join(t[1:]), {"Kd": (0.6, 0.6, 0.6), "Ns": None, "map_Kd": None})
            elif cur is not None and t[0] == "Kd" and len(t) >= 4:
                cur["Kd"] = tuple(float(x) for x in t[1:4])
            elif cur is not None and t[0] == "Ns" and len(t) >= 2:
                cur["Ns"] = float(t[1])
            elif cur is not None and t[0] == "map_Kd":
                cur["map_Kd"] = " ".join(t[1:])
    return mats


def load_obj(path):
    """-> (materials: list of (name, dict), meshes: list of dict(vertices (V,3), indices (F,3), normals (V,3), uvs (V,2),
    material index)), one mesh per material in order of first use."""
    v, vt, vn = [], [], []
    mtl = {}
    groups = {}   # material name -> list of corner triples per triangle
    order = []
    cur = None
    base = os.path.dirname(os.path.abspath(path))
    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            k = t[0]
            if k == "v":
                v.append((float(t[1]), float(t[2]), float(t[3])))
            elif k == "vt":
                vt.append((float(t[1]), float(t[2]) if len(t) > 2 else 0.0))
            elif k == "vn":
                vn.append((float(t[1]), float(t[2]), float(t[3])))
            elif k == "mtllib":
                mtl.update(parse_mtl(os.path.join(base, " ".join(t[1:]))))
            elif k == "usemtl":
                cur = " ".join(t[1:])
            elif k == "f":
                corners = []
                for c in t[1:]:
                    p = (c.split("/") + ["", ""])[:3]
                    iv = int(p[0])
                    it = int(p[1]) if p[1] else 0
                    inn = int(p[2]) if p[2] else 0
                    corners.append((iv - 1 if iv > 0 else len(v) + iv, (it - 1 if it > 0 else len(vt) + it) if it else -1,
                                    (inn - 1 if inn > 0 else len(vn) + inn) if inn else -1))
                name = cur if cur is not None else "__default__"
                if name not in groups:
                    groups[name] = []
                    order.append(name)
                for i in range(1, len(corners) - 1):  # aiProcess_Triangulate: a fan
                    groups[name].append((corners[0], corners[i], corners[i + 1]))
    V = np.asarray(v, np.float32).reshape(-1, 3)
    VT = np.asarray(vt, np.float32).reshape(-1, 2)
    VN = np.asarray(vn, np.float32).reshape(-1, 3)
    materials, meshes = [], []
    for name in order:
        tris = groups[name]
        if not tris:
            continue
        m = mtl.get(name, {"Kd": (0.6, 0.6, 0.6), "Ns": None, "map_Kd": None})
        materials.append((name, m))
        corner = np.asarray(tris, np.int64).reshape(-1, 3)           # (3F, 3): position, texcoord, normal index
        uniq, inv = np.unique(corner, axis=0, return_inverse=True)   # JoinIdenticalVertices
        idx = inv.reshape(-1, 3).astype(np.uint32)
        pos = V[uniq[:, 0]]
        uvs = VT[uniq[:, 1]] if (len(VT) and (uniq[:, 1] >= 0).all()) else np.zeros((len(uniq), 2), np.float32)
        if len(VN) and (uniq[:, 2] >= 0).all():
            nrm = VN[uniq[:, 2]].copy()
            ln = np.linalg.norm(nrm, axis=1, keepdims=True)
            nrm = (nrm / np.maximum(ln, 1e-30)).astype(np.float32)
        else:
            # GenSmoothNormals: per POSITION (not per joined vertex), area-weighted sum of the face normals
            p0, p1, p2 = (V[corner[:, 0]].reshape(-1, 3, 3)[:, k] for k in range(3))
            fn = np.cross(p1 - p0, p2 - p0).astype(np.float64)
            acc = np.zeros((len(V), 3), np.float64)
            pi = corner[:, 0].reshape(-1, 3)
            for k in range(3):
                for a in range(3):
                    acc[:, a] += np.bincount(pi[:, k], weights=fn[:, a], minlength=len(V))
            acc /= np.maximum(np.linalg.norm(acc, axis=1, keepdims=True), 1e-30)
            nrm = acc[uniq[:, 0]].astype(np.float32)
        meshes.append({"vertices": pos, "indices": idx, "normals": nrm, "uvs": uvs, "material": len(materials) - 1})
    return materials, meshes


def load_scene(path, width=480, height=270):
    """A Scene with one mesh + identity instance per material group of the file."""
    materials, meshes = load_obj(path)
    s = scenes.Scene()
    s.name = os.path.basename(path)
    ids = []
    for name, m in materials:
        rough = 1.0 if m["Ns"] is None else max(0.05, min(1.0, math.sqrt(2.0 / (m["Ns"] + 2.0))))
        ids.append(s.add_material(color=tuple(m["Kd"]), roughness=rough))
    for me in meshes:
        s.add_instance(s.add_mesh(me["vertices"], me["indices"], normals=me["normals"], uvs=me["uvs"], material=ids[me["material"]]))
    return s, materials, meshes
