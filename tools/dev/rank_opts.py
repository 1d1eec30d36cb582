"""Development probe: rank 3 of an 8-GPU strip split alone on cuda:0, 128 spp per step, 16 timed steps, settings A/B."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
pkg = load_package()
W, H, spp = 1920, 1080, 128
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
for extra in ({}, {"sample_group": 64}, {"sample_group": 16}, {"ring": 3}, {"ring": 2}, {"overlap": 1}, {"sample_group": 64, "overlap": 1}):
    ctx = pkg.RenderContext(0, 3, 8); ctx.init(W, H); scene.upload(ctx)
    ctx.set_setting("integrator", "pt"); ctx.set_setting("spp", spp)
    for k, v in extra.items(): ctx.set_setting(k, v)
    for k in range(3): ctx.render_async(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
    ctx.wait(); torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(16): ctx.render_async(scene.camera, pkg.CONVERGE)
    ctx.wait(); torch.cuda.synchronize()
    print(extra, "%.3f ms/step" % ((time.perf_counter() - t) / 16 * 1e3), flush=True)
    ctx.destroy()
