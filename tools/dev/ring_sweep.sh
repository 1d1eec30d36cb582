#!/bin/bash
# development probe: Msamples/s for spp x ring (single-sub-batch calls rotating through `ring` buffer sets)
for s in ${SPPS:-1 8 16 32}; do for r in ${RINGS:-2 3 4}; do
  python - "$s" "$r" <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
pkg = load_package()
spp, ring = int(sys.argv[1]), int(sys.argv[2])
W, H = 1920, 1080
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
ctx = pkg.RenderContext(0); ctx.init(W, H); scene.upload(ctx)
ctx.set_setting("integrator", "pt"); ctx.set_setting("spp", spp); ctx.set_setting("ring", ring); ctx.set_setting("overlap", int(os.environ.get("OVERLAP", "-1")))
for k in range(12): ctx.render_async(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
ctx.wait()
n = max(10, 200 // spp)
t = time.perf_counter()
for k in range(n): ctx.render_async(scene.camera, pkg.CONVERGE)
ctx.wait()
dt = (time.perf_counter() - t) / n
print("spp", spp, "ring", ring, "ms/step %.3f" % (dt * 1e3), "Msamples/s %.1f" % (W * H * spp / dt / 1e6), flush=True)
PY
done; done
