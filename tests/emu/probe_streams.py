"""Development probe: do two half-batches on two HIP streams (two contexts) beat one full batch on one stream?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
def mk(spp, rank=0, world=1):
    c = pkg.RenderContext(0, rank, world); c.init(W, H); scene.upload(c)
    c.set_setting("integrator", "pt"); c.set_setting("spp", spp)
    return c
def timeit(ctxs, steps=10):
    for k in range(3):
        for c in ctxs: c.render_async(scene.camera, pkg.RESET)
        for c in ctxs: c.wait()
    t = time.perf_counter()
    for k in range(steps):
        for c in ctxs: c.render_async(scene.camera, pkg.CONVERGE)
    for c in ctxs: c.wait()
    return (time.perf_counter() - t) / steps * 1e3
for total, world in ((8, 1), (8, 8), (16, 8)):
    one = timeit([mk(total, 0, world)])
    two = timeit([mk(total // 2, 0, world), mk(total // 2, 0, world)])
    four = timeit([mk(total // 4, 0, world) for _ in range(4)])
    print("world %d spp %d: 1 stream %.3f ms | 2 streams %.3f ms | 4 streams %.3f ms" % (world, total, one, two, four), flush=True)
