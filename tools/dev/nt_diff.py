"""Development probe: the bench frame (one 64-spp call) rendered by the tree's library and by variant libraries, compared pixel by
pixel — how many pixels differ, where, by how much, and whether a variant reproduces itself."""
import os, sys, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
def render(lib):
    ctx = pkg.RenderContext(0) if lib is None else pkg._binding.CoreBinding(ctypes.CDLL(lib), "rfwhip_", 0, 0, 1)
    ctx.init(W, H); scene.upload(ctx)
    for k, v in {"integrator": "pt", "spp": 64, "max_depth": 2}.items():
        ctx.set_setting(k, v)
    ctx.render_frame(scene.camera, pkg.RESET)
    img = ctx.framebuffer().copy()
    return img
base = render(None)
base2 = render(None)
print("base vs base:", int((base != base2).any(-1).sum()), "pixels differ")
for name in sys.argv[1:]:
    a = render(os.path.join("tools", "dev", "variants", name + ".so"))
    b = render(os.path.join("tools", "dev", "variants", name + ".so"))
    d = (a != base).any(-1)
    ys, xs = np.nonzero(d)
    mag = np.abs(a - base)[d].max(-1) if d.any() else np.zeros(0)
    print(name, "vs base:", int(d.sum()), "pixels differ; self:", int((a != b).any(-1).sum()),
          "; max |d|", float(mag.max()) if len(mag) else 0.0, "; rows", (int(ys.min()), int(ys.max())) if len(ys) else None,
          "cols", (int(xs.min()), int(xs.max())) if len(xs) else None, "first", list(zip(ys[:8].tolist(), xs[:8].tolist())))
