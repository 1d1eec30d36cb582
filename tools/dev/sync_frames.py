"""Development probe: ms per 1080p pt frame of the bench scene — (a) render + wait per frame (the reference's render_frame
contract), (b) frames in flight through rfwhip_group_present_async / _wait (frame k - 1 handed out while k renders),
(c) enqueue-only (the ceiling)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
g = pkg.render_group([0], "peer"); g.init(W, H); scene.upload(g)
g.set_setting("integrator", "pt")
for kv in sys.argv[1:]:   # e.g. ring=4
    k, _, v = kv.partition("="); g.set_setting(k, v)
N = 100
for spp in (1, 8):
    g.set_setting("spp", spp)
    for k in range(10): g.render_frame(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
    t = time.perf_counter()
    for k in range(N): g.render_frame(scene.camera, pkg.CONVERGE)
    a = (time.perf_counter() - t) / N * 1e3
    t = time.perf_counter()
    for k in range(N):
        g.render_frame(scene.camera, pkg.CONVERGE); img = g.framebuffer()
    a2 = (time.perf_counter() - t) / N * 1e3
    bs = []
    for n in (2, 4):
        g.wait()
        t = time.perf_counter()
        for k in range(N):
            g.render_async(scene.camera, pkg.CONVERGE); g.present_async(k % n)
            if k >= n - 1: img = g.present_wait((k + 1) % n)
        g.wait()
        bs.append((time.perf_counter() - t) / N * 1e3)
    b = bs[0]
    g.wait()
    t = time.perf_counter()
    for k in range(N): g.render_async(scene.camera, pkg.CONVERGE)
    g.wait()
    c = (time.perf_counter() - t) / N * 1e3
    print("spp %d: render+wait %.3f ms  render+wait+readback %.3f ms  2 / 4 frames in flight (image on the host every frame) %.3f / %.3f ms  enqueue only %.3f ms" % (spp, a, a2, bs[0], bs[1], c), flush=True)
