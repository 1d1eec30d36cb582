import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from __graft_entry__ import load_package, load_oracle
pkg = load_package(); orc = load_oracle()
W, H = 480, 270
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
out = []
for ctx in (pkg.RenderContext(device=0), orc.OracleContext(pkg)):
    ctx.init(W, H); scene.upload(ctx)
    for k, v in {"integrator": "pt", "spp": 8, "max_depth": 2}.items():
        ctx.set_setting(k, v)
    ctx.render_frame(scene.camera, pkg.RESET)
    out.append(ctx)
hip, ref = out
a, b = hip.framebuffer()[..., :3].astype(np.float64), ref.framebuffer()[..., :3].astype(np.float64)
d = np.sqrt(((a - b) ** 2).sum(-1))
for tol in (1e-3, 1e-2, 3e-2, 1e-1, 0.5):
    print("frac >", tol, (d > tol).mean())
print("rmse", np.sqrt((d ** 2).mean()), "mean rel", abs(a.mean() - b.mean()) / b.mean())
# block means 8x8 (crop)
def blocks(x, k=8):
    h, w = (x.shape[0] // k) * k, (x.shape[1] // k) * k
    return x[:h, :w].reshape(h // k, k, w // k, k, 3).mean((1, 3))
ba, bb = blocks(a), blocks(b)
rel = np.abs(ba - bb).max(-1) / np.maximum(bb.mean(-1), 1e-3)
print("block 8x8: max rel", rel.max(), "mean rel", rel.mean(), "frac > 2%", (rel > 0.02).mean(), "frac>5%", (rel > 0.05).mean())
sa, sb = hip.get_stats(), ref.get_stats()
for name in ("primaryCount", "secondaryCount", "deepCount", "shadowCount"):
    x, y = getattr(sa, name), getattr(sb, name)
    print(name, x, y, (x - y) / max(1, y))
