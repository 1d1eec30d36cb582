"""A few 64-spp pt frames of the bench scene, serialised (one launch at a time), for profilers that attribute samples to kernels.
usage: pc_frame.py [frames] [spp] [scene]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from __graft_entry__ import load_package

pkg = load_package()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W, H = 1920, 1080
scene = pkg.scenes.atrium(W, H) if len(sys.argv) > 3 and sys.argv[3] == "atrium" else pkg.scenes.terrain(n=708, width=W, height_px=H)
ctx = pkg.RenderContext(device=0)
ctx.init(W, H)
scene.upload(ctx)
ctx.set_setting("integrator", "pt")
ctx.set_setting("spp", spp)
ctx.set_setting("max_depth", 2)
ctx.set_setting("fuse", int(os.environ.get("PC_FUSE", "0")))
ctx.set_setting("streams", 1)
for f in range(frames):
    ctx.render_frame(scene.camera, pkg.RESET if f == 0 else 0)
print("frames", frames, "spp", spp, flush=True)
ctx.destroy()
