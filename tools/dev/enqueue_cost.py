"""Development probe: what does the HOST spend per 1-spp frame (time inside render_async, device idle-free: a ring of three keeps
it ahead) against what the device needs per frame — i.e. would capturing a frame's launch chain as a hipGraph buy anything?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
ctx = pkg.RenderContext(0, 0, 1); ctx.init(W, H); scene.upload(ctx)
ctx.set_setting("integrator", "pt")
for spp in (1, 4):
    ctx.set_setting("spp", spp)
    for k in range(20): ctx.render_async(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
    ctx.wait()
    N = 200
    host = 0.0
    t0 = time.perf_counter()
    for k in range(N):
        t = time.perf_counter()
        ctx.render_async(scene.camera, pkg.CONVERGE)
        host += time.perf_counter() - t
    ctx.wait()
    total = time.perf_counter() - t0
    print("spp %d: host inside render_async %.3f ms per frame; frame rate-limited at %.3f ms per frame" % (spp, host / N * 1e3, total / N * 1e3), flush=True)
