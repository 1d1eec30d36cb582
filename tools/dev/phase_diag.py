"""Development probe (build with RFWHIP_EXTRA_FLAGS=-DRT_DIAG_PHASES): lane utilisation of the node loop and of the
triangle loop of the traversal kernels.  With the flag, lds_* counts wave-level node-loop iterations and tris_* wave-level
triangle-loop iterations; inner_* stays per lane."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
ctx = pkg.RenderContext(0); ctx.init(W, H); scene.upload(ctx)
ctx.set_setting("integrator", "pt"); ctx.set_setting("spp", 32); ctx.set_setting("streams", 1); ctx.set_setting("count_traversal", 1)
ctx.render_frame(scene.camera, pkg.RESET)
ctx.get_counters(reset=True)
ctx.render_frame(scene.camera, pkg.RESET)
c = ctx.get_counters(reset=True)
print(c)
for k in ("extend", "shadow"):
    rays, inner, wnode, wtri = c["rays_" + k], c["inner_" + k], c["lds_" + k], c["tris_" + k]
    print(k, "rays %.1fM" % (rays / 1e6), "node steps per ray %.2f" % (inner / rays), "node-loop lanes active %.1f of 64" % (inner / max(1, wnode)),
          "wave node iterations per ray %.3f" % (wnode / rays), "wave triangle iterations per ray %.3f" % (wtri / rays))
