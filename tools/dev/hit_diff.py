"""GPU probe: how far are the HIP hit records from the oracle's on the 1 M-triangle terrain, now that both state the triangle
test in the same fixed shape (v_rcp_f32 is the one operation that differs)?  Sets the bounds of tests/test_fullsize_gpu.py."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
from __graft_entry__ import load_package, load_oracle
from test_trace_rays import _rays

pkg, orc = load_package(), load_oracle()
terrain = pkg.scenes.terrain(n=708, width=64, height_px=64)
core, ref = pkg.RenderContext(device=0), orc.OracleContext(pkg)
for c in (core, ref):
    c.init(64, 64)
    terrain.upload(c)
o, d = _rays(np.random.default_rng(5), 300000, 48.0)
a, b = core.trace_rays(o, d), ref.trace_rays(o, d)
same = (a["prim"] == b["prim"]) & (a["prim"] >= 0)
print("prim differ %.3g inst differ %.3g same-hit share %.3f" % ((a["prim"] != b["prim"]).mean(), (a["inst"] != b["inst"]).mean(), same.mean()))
for k in ("t", "u", "v"):
    dd = np.abs(a[k][same] - b[k][same])
    rel = dd / np.maximum(np.abs(b[k][same]), 1e-30)
    print(k, "max abs %.3g  max rel %.3g  bit-equal share %.4f  abs 99.9%% %.3g" % (dd.max(), rel[np.abs(b[k][same]) > 1e-3].max(), (dd == 0).mean(), np.quantile(dd, 0.999)))
