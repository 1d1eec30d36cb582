"""Development probe: 200 pipelined 1080p pt frames at the given spp (default 1) — run under rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
ctx = pkg.RenderContext(0); ctx.init(W, H); scene.upload(ctx)
ctx.set_setting("integrator", "pt"); ctx.set_setting("spp", spp)
for kv in sys.argv[2:]:
    k, _, v = kv.partition("="); ctx.set_setting(k, v)
for k in range(20): ctx.render_async(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
ctx.wait()
t = time.perf_counter()
for k in range(200): ctx.render_async(scene.camera, pkg.CONVERGE)
ctx.wait()
print("spp %d: %.3f ms/frame pipelined" % (spp, (time.perf_counter() - t) / 200 * 1e3), flush=True)
