import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from __graft_entry__ import load_package, load_oracle
import test_bvh as T
pkg = load_package(); orc = load_oracle()
kind, builder = "comb", "host"
rng = np.random.default_rng(sum(map(ord, kind + builder)))
verts = T._soup(kind, rng, 600)
s = pkg.scenes.Scene(); s.add_material(color=(0.8, 0.8, 0.8)); s.add_instance(s.add_mesh(verts, None)); s.set_test_sky(16, 8)
cam = pkg.Camera(aperture=0.0); lo, hi = verts.min(0), verts.max(0)
cam.look_at(tuple(lo - (hi - lo)), tuple((lo + hi) / 2)); cam.resize(16, 16); s.camera = cam
ref = orc.OracleContext(pkg); ref.set_setting("bvh", 0); ref.init(16, 16); s.upload(ref)
n = 4000
tri_c = verts.reshape(-1, 3, 3).mean(1)
org = (tri_c[rng.integers(0, len(tri_c), n)] + rng.normal(0, 1.0, (n, 3)) * (hi - lo) * 0.7).astype(np.float32)
tgt = tri_c[rng.integers(0, len(tri_c), n)] + rng.normal(0, 0.05, (n, 3))
d = tgt - org; d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
b = ref.trace_rays(org, d)
for lds in (-1, 0):
    core = pkg.RenderContext(device=0); core.set_setting("builder", builder); core.set_setting("lds_nodes", lds)
    core.init(16, 16); s.upload(core)
    a = core.trace_rays(org, d)
    ha, hb = a["prim"] >= 0, b["prim"] >= 0
    mm = np.nonzero(ha != hb)[0]
    print("lds_nodes", lds, "mismatches", len(mm), "gpu-hit-only", int((ha & ~hb).sum()), "oracle-hit-only", int((~ha & hb).sum()))
    for i in mm[:10]:
        p = b["prim"][i] if hb[i] else a["prim"][i]
        print("  ray", i, "gpu", a["prim"][i], a["t"][i], "oracle", b["prim"][i], b["t"][i], "u,v", (b["u"][i], b["v"][i]) if hb[i] else (a["u"][i], a["v"][i]), "tri size", np.ptp(verts.reshape(-1,3,3)[p], axis=0).max())
    both = ha & hb
    print("  prim mismatch among both-hit:", int((a["prim"][both] != b["prim"][both]).sum()), "max |dt|/t", float((np.abs(a["t"][both]-b["t"][both])/np.abs(b["t"][both])).max()))
    scale = float(np.abs(verts).max())
    same = both & (a["prim"] == b["prim"])
    err = np.abs(a["t"] - b["t"]) - (1e-6 * scale + 2e-5 * np.abs(b["t"]))
    bad = np.nonzero(same & (err > 0))[0]
    print("  t-tolerance violations among same-prim hits:", len(bad), "scale", scale)
    for i in bad[:10]:
        print("   ray", i, "prim", a["prim"][i], "t gpu", a["t"][i], "t oracle", b["t"][i], "uv gpu", a["u"][i], a["v"][i], "uv oracle", b["u"][i], b["v"][i])
