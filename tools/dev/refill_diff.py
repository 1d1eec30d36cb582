"""Development probe: image / count differences between the persistent-lane kernels (refill=7) and the one-ray-per-lane kernels (refill=0)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
out = {}
for refill in (7, 0, 3):
    c = pkg.RenderContext(0); c.init(W, H); scene.upload(c)
    for k, v in (("integrator", "pt"), ("spp", 4), ("max_depth", 3), ("refill", refill)): c.set_setting(k, v)
    c.render_frame(scene.camera, pkg.RESET)
    st = c.get_stats()
    out[refill] = (c.framebuffer().copy(), (st.primaryCount, st.secondaryCount, st.deepCount, st.shadowCount))
    c.destroy()
a = out[7][0]
for r in (0, 3):
    b = out[r][0]
    d = np.abs(a - b)[..., :3].max(-1)
    print("refill 7 vs", r, "counts", out[7][1], out[r][1], "pixels differing", int((d > 0).sum()), "max diff", float(d.max()), "pixels > 1e-3", int((d > 1e-3).sum()))
