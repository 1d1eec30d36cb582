import os, sys, time
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
for world, spp in ((8, 8), (8, 16), (1, 1), (1, 2)):
    c = pkg.RenderContext(0, 0, world); c.init(W, H); scene.upload(c)
    c.set_setting("integrator", "pt"); c.set_setting("spp", spp); c.set_setting("stage_timing", 1)
    for k in range(3): c.render_frame(scene.camera, pkg.RESET)
    acc = {}
    n = 10
    t = time.perf_counter()
    for k in range(n):
        c.render_frame(scene.camera, pkg.RESET)
        st = c.get_stats().as_dict()
        for key in ("primaryTime", "secondaryTime", "deepTime", "shadowTime", "shadeTime", "finalizeTime"):
            acc[key] = acc.get(key, 0) + st[key] / n
    wall = (time.perf_counter() - t) / n * 1e3
    print("world", world, "spp", spp, {k: round(v, 3) for k, v in acc.items()}, "sum %.3f wall %.3f" % (sum(acc.values()), wall), {k: st[k] for k in ("primaryCount", "secondaryCount", "deepCount", "shadowCount")}, flush=True)
    c.destroy()
