"""Development probe: enqueue-only 1-spp frames through a plain context and through a group of one."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
def run(t, name):
    t.init(W, H); scene.upload(t); t.set_setting("integrator", "pt"); t.set_setting("spp", 1)
    for k in range(20): t.render_async(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
    t.wait()
    for rep in range(3):
        s = time.perf_counter()
        for k in range(200): t.render_async(scene.camera, pkg.CONVERGE)
        t.wait()
        print(name, "%.3f ms/frame" % ((time.perf_counter() - s) / 200 * 1e3), flush=True)
run(pkg.RenderContext(0), "context")
run(pkg.render_group([0], "peer"), "group  ")
run(pkg.RenderContext(0), "context")
