"""Development probe: ms per 1-spp 1080p pt frame of the bench scene for the refill masks given on the command line."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
for spp in (1, 8):
    for refill in [int(x) for x in sys.argv[1:]] or [3, 7]:
        ctx = pkg.RenderContext(0); ctx.init(W, H); scene.upload(ctx)
        ctx.set_setting("integrator", "pt"); ctx.set_setting("spp", spp); ctx.set_setting("refill", refill)
        for k in range(20): ctx.render_async(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
        ctx.wait()
        t = time.perf_counter()
        for k in range(100): ctx.render_async(scene.camera, pkg.CONVERGE)
        ctx.wait()
        dt = (time.perf_counter() - t) * 10
        ctx.set_setting("stage_timing", 1); ctx.set_setting("streams", 1)
        ctx.render_frame(scene.camera, pkg.RESET); ctx.render_frame(scene.camera, pkg.RESET)
        st = ctx.get_stats().as_dict()
        print("spp", spp, "refill", refill, "ms/frame %.3f" % dt, {k: round(st[k], 3) for k in ("primaryTime", "secondaryTime", "deepTime", "shadowTime", "shadeTime")}, flush=True)
        ctx.destroy()
