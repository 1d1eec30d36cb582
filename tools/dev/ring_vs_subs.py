"""Development probe: one sub-batch per call on the ring of buffer sets vs four sub-batches per call, by batch size."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
spp, force_ring = int(sys.argv[1]), int(sys.argv[2])
ctx = pkg.RenderContext(0); ctx.init(W, H); scene.upload(ctx)
ctx.set_setting("integrator", "pt"); ctx.set_setting("spp", spp)
if force_ring:
    ctx.set_setting("sub_batch_paths", 1 << 40)
else:
    ctx.set_setting("sub_batch_paths", 1000000)
for k in range(6): ctx.render_async(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
ctx.wait()
n = max(8, 256 // spp)
t = time.perf_counter()
for k in range(n): ctx.render_async(scene.camera, pkg.CONVERGE)
ctx.wait()
dt = (time.perf_counter() - t) / n
print("spp", spp, "ring" if force_ring else "4 sub-batches", "ms/step %.3f" % (dt * 1e3), "Msamples/s %.1f" % (W * H * spp / dt / 1e6), flush=True)
