"""glTF 2.0 front end for the rendercore: geometry, skins and node animations of a .gltf file become
`Scene` meshes (the reference's `Mesh` / `Triangle` records), a rig for `rfwhip_set_mesh_skin` and, per time value, the
joint matrices for `rfwhip_pose_mesh`.

What it restates from the reference (which parses with tinygltf, a third-party library, and keeps the result in
`geometry/gltf/*`):
  * node transform  localTransform = T * R * S * matrix            (node.cpp:108-116)
  * hierarchy       combined = parent.combined * local             (node.cpp:54-66)
  * joint matrices  inverse(meshNode.combined) * joint.combined * inverseBind[j]     (node.cpp:90-98)
  * samplers        LINEAR: lerp for translation / scale, component-wise lerp + normalise for rotations (not slerp),
                    STEP, CUBICSPLINE (animation.cpp:230-320); time wraps with fmod(duration) (animation.cpp:378-390)
  * morph targets   pose = base + sum_j w_j * target_j for positions AND normals, normals not renormalised
                    (mesh.cpp:127-147; poses built at object.cpp:463-497); weights from the mesh / node defaults and from
                    "weights" animation channels (animation.cpp:201,231-257,365-367: sampleFloat over `count` targets)
  * containers      .gltf (json + external / data-uri buffers) and binary .glb (12-byte header, JSON chunk, BIN chunk)
Not covered: cameras, KHR extensions, sparse accessors, image decoding (no image library in this environment: textures are
referenced by index only); materials map baseColorFactor / metallic / roughness only.  File parsing itself is plain
json + base64 / external buffers + numpy.
"""
import base64
import json
import math
import os
import struct

import numpy as np

from . import scenes

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_WIDTH = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _quat_to_mat(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 0],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w), 0],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y), 0],
                     [0, 0, 0, 1]], np.float64)


class Gltf:
    def __init__(self, path):
        self.dir = os.path.dirname(os.path.abspath(path))
        self.glb_bin = None
        with open(path, "rb") as f:
            raw = f.read()
        if raw[:4] == b"glTF":
            # binary container: header (magic, version, length), then chunks (length, type, payload); chunk 0 = JSON,
            # an optional chunk 1 = BIN, the buffer without a uri
            magic, version, length = struct.unpack_from("<4sII", raw, 0)
            if version != 2:
                raise ValueError("%s: glb version %d (only 2 is defined)" % (path, version))
            off = 12
            self.doc = None
            while off + 8 <= min(length, len(raw)):
                clen, ctype = struct.unpack_from("<II", raw, off)
                payload = raw[off + 8:off + 8 + clen]
                if ctype == 0x4E4F534A:  # "JSON"
                    self.doc = json.loads(payload.decode("utf-8"))
                elif ctype == 0x004E4942 and self.glb_bin is None:  # "BIN\0"
                    self.glb_bin = payload
                off += 8 + clen + ((4 - clen % 4) % 4)
            if self.doc is None:
                raise ValueError("%s: glb without a JSON chunk" % path)
        else:
            self.doc = json.loads(raw.decode("utf-8"))
        self.buffers = [self._load_buffer(b) for b in self.doc.get("buffers", [])]
        self.nodes = self.doc.get("nodes", [])
        self.parents = [-1] * len(self.nodes)
        for i, n in enumerate(self.nodes):
            for c in n.get("children", []):
                self.parents[c] = i
        # animated TRS state per node, initialised from the file
        self.T = [np.asarray(n.get("translation", (0, 0, 0)), np.float64) for n in self.nodes]
        self.R = [np.asarray(n.get("rotation", (0, 0, 0, 1)), np.float64) for n in self.nodes]
        self.S = [np.asarray(n.get("scale", (1, 1, 1)), np.float64) for n in self.nodes]
        self.M = [np.asarray(n["matrix"], np.float64).reshape(4, 4).T if "matrix" in n else np.eye(4) for n in self.nodes]
        # morph-target weights per node: the node's own, else its mesh's defaults, else zeros (node.cpp:44-51)
        self.W = []
        for n in self.nodes:
            w = None
            if "mesh" in n:
                mesh = self.doc["meshes"][n["mesh"]]
                nt = len(mesh["primitives"][0].get("targets", []))
                w = np.asarray(n.get("weights", mesh.get("weights", [0.0] * nt)), np.float64)
            self.W.append(w)
        self.animations = [self._load_animation(a) for a in self.doc.get("animations", [])]

    # ---- buffers / accessors ---------------------------------------------------------------------------------------
    def _load_buffer(self, b):
        if "uri" not in b:
            if self.glb_bin is None:
                raise ValueError("buffer without a uri outside a .glb container")
            return self.glb_bin
        uri = b["uri"]
        if uri.startswith("data:"):
            return base64.b64decode(uri.split(",", 1)[1])
        with open(os.path.join(self.dir, uri), "rb") as f:
            return f.read()

    def accessor(self, idx):
        a = self.doc["accessors"][idx]
        v = self.doc["bufferViews"][a["bufferView"]]
        dt = np.dtype(_COMPONENT[a["componentType"]])
        w = _WIDTH[a["type"]]
        off = v.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = v.get("byteStride", 0) or dt.itemsize * w
        raw = np.frombuffer(self.buffers[v["buffer"]], dtype=np.uint8, count=stride * (a["count"] - 1) + dt.itemsize * w,
                            offset=off)
        rows = np.lib.stride_tricks.as_strided(raw, shape=(a["count"], dt.itemsize * w), strides=(stride, 1))
        out = np.ascontiguousarray(rows).view(dt).reshape(a["count"], w)
        if a.get("normalized", False) and dt != np.float32:
            # animation.cpp:62-110: signed types scale by 127 / 32767 (clamped at -1), unsigned by 255 / 65535
            scale = {np.int8: 127.0, np.uint8: 255.0, np.int16: 32767.0, np.uint16: 65535.0}[dt.type]
            out = np.maximum(out.astype(np.float32) / np.float32(scale), -1.0)
        return out

    # ---- hierarchy -------------------------------------------------------------------------------------------------
    def local(self, i):
        t = np.eye(4)
        t[:3, 3] = self.T[i]
        s = np.diag(list(self.S[i]) + [1.0])
        return t @ _quat_to_mat(self.R[i]) @ s @ self.M[i]

    def combined(self, i):
        m = self.local(i)
        p = self.parents[i]
        while p >= 0:
            m = self.local(p) @ m
            p = self.parents[p]
        return m

    # ---- animation -------------------------------------------------------------------------------------------------
    def _load_animation(self, a):
        samplers = []
        for s in a["samplers"]:
            samplers.append({"method": s.get("interpolation", "LINEAR"), "times": self.accessor(s["input"])[:, 0].astype(np.float64),
                             "keys": self.accessor(s["output"]).astype(np.float64)})
        chans = [{"sampler": c["sampler"], "node": c["target"]["node"], "path": c["target"]["path"]} for c in a["channels"]]
        return {"samplers": samplers, "channels": chans}

    @staticmethod
    def _sample(s, time):
        times, keys, method = s["times"], s["keys"], s["method"]
        duration = times[-1]
        if time > duration and duration > 0:
            time = math.fmod(time, duration)
        k = 0
        while k + 2 < len(times) and time > times[k + 1]:
            k += 1
        if len(times) == 1:
            return keys[0 if method != "CUBICSPLINE" else 1]
        t0, t1 = times[k], times[k + 1]
        f = (time - t0) / (t1 - t0)
        if f <= 0:
            return keys[0 if method != "CUBICSPLINE" else 1]
        if method == "STEP":
            return keys[k]
        if method == "CUBICSPLINE":
            t, t2, t3 = f, f * f, f * f * f
            p0, m0 = keys[k * 3 + 1], (t1 - t0) * keys[k * 3 + 2]
            p1, m1 = keys[(k + 1) * 3 + 1], (t1 - t0) * keys[(k + 1) * 3]
            return m0 * (t3 - 2 * t2 + t) + p0 * (2 * t3 - 3 * t2 + 1) + p1 * (-2 * t3 + 3 * t2) + m1 * (t3 - t2)
        return (1 - f) * keys[k] + f * keys[k + 1]

    def set_time(self, time, animation=0):
        """SceneAnimation::setTime: every channel writes its node's translation / rotation / scale."""
        if not self.animations:
            return
        a = self.animations[animation]
        for c in a["channels"]:
            smp = a["samplers"][c["sampler"]]
            if c["path"] == "weights":
                # one key = `count` consecutive scalars (x3 for CUBICSPLINE): sampleFloat(time, k, i, count)
                count = len(self.W[c["node"]])
                per_key = count * (3 if smp["method"] == "CUBICSPLINE" else 1)
                keyed = dict(smp, keys=smp["keys"].reshape(-1, per_key))
                if smp["method"] == "CUBICSPLINE":  # (in-tangent, value, out-tangent) per target -> three rows per key
                    keyed["keys"] = smp["keys"].reshape(-1, count, 3).transpose(0, 2, 1).reshape(-1, count)
                self.W[c["node"]] = np.asarray(self._sample(keyed, float(time)), np.float64).reshape(-1)[:count]
                continue
            v = self._sample(smp, float(time))
            if c["path"] == "translation":
                self.T[c["node"]] = v[:3]
            elif c["path"] == "scale":
                self.S[c["node"]] = v[:3]
            elif c["path"] == "rotation":
                self.R[c["node"]] = v[:4] / np.linalg.norm(v[:4])  # sampleQuat normalises

    # ---- meshes / skins --------------------------------------------------------------------------------------------
    def mesh_nodes(self):
        return [i for i, n in enumerate(self.nodes) if "mesh" in n]

    def primitive(self, mesh_index, prim=0):
        p = self.doc["meshes"][mesh_index]["primitives"][prim]
        at = p["attributes"]
        pos = self.accessor(at["POSITION"]).astype(np.float32)
        idx = self.accessor(p["indices"]).astype(np.uint32).reshape(-1, 3) if "indices" in p else None
        nrm = self.accessor(at["NORMAL"]).astype(np.float32) if "NORMAL" in at else None
        uv = self.accessor(at["TEXCOORD_0"]).astype(np.float32) if "TEXCOORD_0" in at else None
        joints = self.accessor(at["JOINTS_0"]).astype(np.uint32) if "JOINTS_0" in at else None
        weights = self.accessor(at["WEIGHTS_0"]).astype(np.float32) if "WEIGHTS_0" in at else None
        if nrm is None:
            tri = idx if idx is not None else np.arange(len(pos), dtype=np.uint32).reshape(-1, 3)
            fn = np.cross(pos[tri[:, 1]] - pos[tri[:, 0]], pos[tri[:, 2]] - pos[tri[:, 0]])
            nrm = scenes._accumulate_vertex_normals(len(pos), tri, fn)
        # morph targets: displacement sets for positions and normals (object.cpp:476-497); a target without NORMAL adds 0
        targets = []
        for t in p.get("targets", []):
            tp = self.accessor(t["POSITION"]).astype(np.float32) if "POSITION" in t else np.zeros_like(pos)
            tn = self.accessor(t["NORMAL"]).astype(np.float32) if "NORMAL" in t else np.zeros_like(nrm)
            targets.append((tp, tn))
        return {"positions": pos, "indices": idx, "normals": nrm, "uvs": uv, "joints": joints, "weights": weights,
                "material": p.get("material", -1), "targets": targets}

    def morphed(self, node_index, prim=0):
        """SceneMesh::set_pose(weights) (mesh.cpp:127-147) for the mesh of `node_index` with the node's current weights:
        positions and normals = base + sum_j w_j * target_j (normals are not renormalised there)."""
        pr = self.primitive(self.nodes[node_index]["mesh"], prim)
        pos, nrm = pr["positions"].copy(), pr["normals"].copy()
        for w, (tp, tn) in zip(self.W[node_index], pr["targets"]):
            pos += np.float32(w) * tp
            nrm += np.float32(w) * tn
        return pos, nrm

    def joint_matrices(self, node_index):
        """Joint matrices of the skin attached to mesh node `node_index` in the current pose, (J, 4, 4) row-major."""
        skin = self.doc["skins"][self.nodes[node_index]["skin"]]
        ibm = self.accessor(skin["inverseBindMatrices"]).astype(np.float64).reshape(-1, 4, 4).transpose(0, 2, 1)
        inv = np.linalg.inv(self.combined(node_index))
        return np.stack([inv @ self.combined(j) @ ibm[k] for k, j in enumerate(skin["joints"])]).astype(np.float32)

    def material(self, idx):
        if idx < 0:
            return scenes.host_material()
        pbr = self.doc["materials"][idx].get("pbrMetallicRoughness", {})
        c = pbr.get("baseColorFactor", (1, 1, 1, 1))
        return scenes.host_material(color=tuple(float(v) for v in c[:3]), metallic=float(pbr.get("metallicFactor", 1.0)),
                                    roughness=float(pbr.get("roughnessFactor", 1.0)))


def load_scene(path, width=480, height=270):
    """A Scene with one mesh + instance per mesh node (first primitive), the node transforms as instance transforms, and
    per mesh node the rig (joints, weights, bind normals) when it is skinned.  Returns (scene, gltf, rigs) with
    rigs = {scene mesh index: (node index, joints, weights, normals)}."""
    g = Gltf(path)
    s = scenes.Scene()
    s.name = os.path.basename(path)
    rigs = {}
    mats = {}
    for ni in g.mesh_nodes():
        prim = g.primitive(g.nodes[ni]["mesh"])
        if prim["material"] not in mats:
            mats[prim["material"]] = s.add_material(**g.material(prim["material"]))
        m = s.add_mesh(prim["positions"], prim["indices"], normals=prim["normals"], uvs=prim["uvs"], material=mats[prim["material"]])
        # skinned meshes are posed in mesh space by the joint matrices; the node transform places the instance
        s.add_instance(m, g.combined(ni).astype(np.float32))
        if prim["joints"] is not None and "skin" in g.nodes[ni]:
            rigs[m] = (ni, prim["joints"], prim["weights"], prim["normals"])
    return s, g, rigs
