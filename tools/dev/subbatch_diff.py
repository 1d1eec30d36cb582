"""GPU probe: 16 spp of the bench scene cut 4/4/4/4 (streams=4) and 5/5/6 (streams=3): which pixels differ, by how much."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from __graft_entry__ import load_package

pkg = load_package()
W, H = 1920, 1080
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)


def run(streams):
    ctx = pkg.RenderContext(device=0)
    ctx.init(W, H)
    scene.upload(ctx)
    for k, v in (("integrator", "pt"), ("spp", 16), ("max_depth", 2), ("streams", streams), ("sub_batch_paths", 1000000)):
        ctx.set_setting(k, v)
    ctx.render_frame(scene.camera, pkg.RESET)
    img = ctx.framebuffer().copy()
    ctx.destroy()
    return img


a, b = run(4), run(3)
d = np.abs(a - b).max(-1)
ys, xs = np.nonzero(d)
print("differing pixels:", len(ys), "max diff", float(d.max()))
for y, x in list(zip(ys, xs))[:10]:
    print(y, x, a[y, x], b[y, x])
