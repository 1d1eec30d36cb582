#!/usr/bin/env python3
"""Generate tests/golden/asset_*.npz — DATA derived from the assets BASELINE config 5 and SURVEY §8 f4 name, read with the
product's own front ends in the development container (the assets live in the reference checkout, /root/reference/assets, and
do not travel to the GPU box):

  asset_cesiumman.npz   assets/models/CesiumMan/CesiumMan.gltf: bind-pose vertices / normals / indices / joints / weights of the
                        skinned primitive, the POSITION accessor's min / max as the FILE states them, and for three animation times
                        the joint matrices and the posed vertices / normals — posed HERE by a float64 numpy restatement of
                        SceneMesh::set_pose (geometry/gltf/mesh.cpp:31-45), independent of the C oracle and of the device kernel
  asset_morphcube.npz   assets/models/AnimatedMorphCube.glb (.glb container, two morph targets, a "weights" animation channel):
                        base / target arrays, the weights at four times and the morphed vertices (mesh.cpp:127-147)
  asset_obj.npz         assets/models/legocar.obj + .mtl and sphere.obj: triangle counts per material, bounds, surface area,
                        Kd colours — summaries only

    python tests/golden/make_golden_assets.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

ASSETS = "/root/reference/assets/models"
TIMES = (0.0, 0.45, 1.3)


def skin_f64(base_v, base_n, joints, weights, mats):
    """vertex = sum_k w_k M[j_k] * base; normal = normalize(base_normal * inverse(that matrix)) (row vector), in float64."""
    m = np.einsum("vk,vkij->vij", weights.astype(np.float64), mats.astype(np.float64)[joints])
    v4 = np.concatenate([base_v.astype(np.float64), np.ones((len(base_v), 1))], 1)
    pv = np.einsum("vij,vj->vi", m, v4)[:, :3]
    inv = np.linalg.inv(m)
    n4 = np.concatenate([base_n.astype(np.float64), np.zeros((len(base_n), 1))], 1)
    pn = np.einsum("vj,vji->vi", n4, inv)[:, :3]
    pn /= np.linalg.norm(pn, axis=1, keepdims=True)
    return pv.astype(np.float32), pn.astype(np.float32)


def main():
    pkg = load_package()
    # ---- CesiumMan ---------------------------------------------------------------------------------------------------
    path = os.path.join(ASSETS, "CesiumMan", "CesiumMan.gltf")
    g = pkg.gltf.Gltf(path)
    ni = [i for i in g.mesh_nodes() if "skin" in g.nodes[i]][0]
    pr = g.primitive(g.nodes[ni]["mesh"])
    doc = json.load(open(path))
    acc = doc["accessors"][doc["meshes"][g.nodes[ni]["mesh"]]["primitives"][0]["attributes"]["POSITION"]]
    out = dict(positions=pr["positions"], normals=pr["normals"], indices=pr["indices"], joints=pr["joints"], weights=pr["weights"],
               file_min=np.asarray(acc["min"], np.float32), file_max=np.asarray(acc["max"], np.float32), times=np.asarray(TIMES, np.float32),
               node_transform=g.combined(ni).astype(np.float32))
    jm, pv, pn = [], [], []
    for t in TIMES:
        g.set_time(t)
        m = g.joint_matrices(ni)
        v, n = skin_f64(pr["positions"], pr["normals"], pr["joints"], pr["weights"], m)
        jm.append(m), pv.append(v), pn.append(n)
    out.update(joint_matrices=np.stack(jm), posed_positions=np.stack(pv), posed_normals=np.stack(pn))
    np.savez_compressed(os.path.join(HERE, "asset_cesiumman.npz"), **out)
    print("CesiumMan:", len(pr["positions"]), "vertices,", len(pr["indices"]), "triangles,", len(jm[0]), "joints; posed bounds",
          pv[1].min(0), pv[1].max(0))
    # ---- AnimatedMorphCube.glb ---------------------------------------------------------------------------------------------
    g = pkg.gltf.Gltf(os.path.join(ASSETS, "AnimatedMorphCube.glb"))
    ni = g.mesh_nodes()[0]
    pr = g.primitive(g.nodes[ni]["mesh"])
    times = (0.0, 0.5, 1.0, 1.7)
    ws, mv, mn = [], [], []
    for t in times:
        g.set_time(t)
        w = g.W[ni].astype(np.float32)
        p = pr["positions"].astype(np.float64) + sum(float(w[j]) * pr["targets"][j][0].astype(np.float64) for j in range(len(w)))
        n = pr["normals"].astype(np.float64) + sum(float(w[j]) * pr["targets"][j][1].astype(np.float64) for j in range(len(w)))
        ws.append(w), mv.append(p.astype(np.float32)), mn.append(n.astype(np.float32))
    np.savez_compressed(os.path.join(HERE, "asset_morphcube.npz"), positions=pr["positions"], normals=pr["normals"], indices=pr["indices"],
                        target_positions=np.stack([t[0] for t in pr["targets"]]), target_normals=np.stack([t[1] for t in pr["targets"]]),
                        times=np.asarray(times, np.float32), weights=np.stack(ws), morphed_positions=np.stack(mv), morphed_normals=np.stack(mn),
                        node_transform=g.combined(ni).astype(np.float32))
    print("AnimatedMorphCube:", len(pr["positions"]), "vertices,", len(pr["targets"]), "targets, weights", [list(np.round(w, 4)) for w in ws])
    # ---- OBJ ---------------------------------------------------------------------------------------------------------------
    o = {}
    for name in ("legocar", "sphere"):
        mats, meshes = pkg.obj.load_obj(os.path.join(ASSETS, name + ".obj"))
        area = 0.0
        lo, hi = np.full(3, 1e30), np.full(3, -1e30)
        for me in meshes:
            p = me["vertices"][me["indices"]].astype(np.float64)
            area += 0.5 * np.linalg.norm(np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), axis=1).sum()
            lo, hi = np.minimum(lo, me["vertices"].min(0)), np.maximum(hi, me["vertices"].max(0))
        o[name + "_tris"] = np.asarray([len(me["indices"]) for me in meshes], np.int64)
        o[name + "_verts"] = np.asarray([len(me["vertices"]) for me in meshes], np.int64)
        o[name + "_kd"] = np.asarray([m[1]["Kd"] for m in mats], np.float32)
        o[name + "_bounds"] = np.stack([lo, hi]).astype(np.float32)
        o[name + "_area"] = np.float64(area)
        print(name, "triangles per material", list(o[name + "_tris"]), "area %.6g" % area)
    np.savez_compressed(os.path.join(HERE, "asset_obj.npz"), **o)


if __name__ == "__main__":
    main()
