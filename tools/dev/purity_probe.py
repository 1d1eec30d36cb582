"""Development probe: what is the packet form of the depth-0 connection wave worth when every ray of a run goes to ONE light?
terrain with all its lights / one area light only / one point light only; serialised shadowTime per 64-spp frame, packets off / on."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
def variant(kind):
    s = pkg.scenes.terrain(n=708, width=W, height_px=H)
    if kind == "one_area":
        s.area_lights = s.area_lights[:1]; s.point_lights = []
    elif kind == "one_point":
        s.area_lights = s.area_lights[:0]; s.point_lights = s.point_lights[:1]
    return s
for kind in ("all", "one_area", "one_point"):
    s = variant(kind)
    res = []
    for sp in (0, 1):
        c = pkg.RenderContext(0); c.init(W, H); s.upload(c)
        for k, v in (("integrator", "pt"), ("spp", 64), ("streams", 1), ("fuse", 0), ("overlap", 0), ("shadow_side", 0), ("shadow_packets", sp), ("stage_timing", 1), ("max_depth", 1)):
            c.set_setting(k, v)
        c.render_frame(s.camera, pkg.RESET)
        acc = 0.0
        for k in range(3):
            c.render_frame(s.camera, pkg.CONVERGE)
            st = c.get_stats()
            acc += st.shadowTime / 3
        res.append((acc, st.shadowCount, c.get_setting("shadow_bins_per_run")))
        c.destroy()
    print(kind, "shadow rays %d: per-lane %.3f ms, packets %.3f ms (bins per run %s)" % (res[0][1], res[0][0], res[1][0], res[1][2]), flush=True)
