"""Development probe: one rank of an 8-GPU strip split alone on cuda:0 — ms/step over K timed steps for several K (pipeline
ramp-up / drain of the ring of buffer sets) and sub-batch settings."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
pkg = load_package()
W, H, spp = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 128
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
for extra in ({}, {"sub_batch_paths": 4000000}, {"sub_batch_paths": 8000000, "streams": 2}):
    ctx = pkg.RenderContext(0, 3, 8); ctx.init(W, H); scene.upload(ctx)
    ctx.set_setting("integrator", "pt"); ctx.set_setting("spp", spp)
    for k, v in extra.items(): ctx.set_setting(k, v)
    out = []
    for K in (4, 8, 16, 40):
        for k in range(2): ctx.render_async(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
        ctx.wait(); torch.cuda.synchronize()
        t = time.perf_counter()
        for k in range(K): ctx.render_async(scene.camera, pkg.CONVERGE)
        ctx.wait(); torch.cuda.synchronize()
        out.append("K=%d: %.3f" % (K, (time.perf_counter() - t) / K * 1e3))
    print(spp, extra, " ".join(out), flush=True)
    ctx.destroy()
