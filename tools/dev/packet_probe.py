"""GPU probe: the packet (wave-uniform) form of the pt primary wave against the per-lane kernels — primary hit records
compared bit for bit, images compared, and the serialised primary stage timed.  usage: packet_probe.py [grid] [spp]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from __graft_entry__ import load_package

pkg = load_package()
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 708
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W, H = 1920, 1080


def run(scene, refill, spp, sample_group=64, reps=3, w=W, h=H):
    ctx = pkg.RenderContext(device=0)
    ctx.init(w, h)
    scene.upload(ctx)
    ctx.set_setting("integrator", "pt")
    ctx.set_setting("spp", spp)
    ctx.set_setting("max_depth", 2)
    ctx.set_setting("refill", refill)
    ctx.set_setting("sample_group", sample_group)
    ctx.set_setting("stage_timing", 1)
    ctx.render_frame(scene.camera, pkg.RESET)
    hits = ctx.primary_hits()
    img = ctx.framebuffer().copy()
    st0 = ctx.get_stats()
    t = []
    for _ in range(reps):
        t0 = time.time()
        ctx.render_frame(scene.camera, pkg.RESET)
        t.append(time.time() - t0)
    st = ctx.get_stats()
    ctx.destroy()
    return hits, img, st, min(t)


for name, scene, s, g in (("terrain%d" % grid, pkg.scenes.terrain(n=grid, width=W, height_px=H), spp, 64),
                          ("terrain%d-1spp" % grid, None, 1, 1),
                          ("atrium", pkg.scenes.atrium(W, H), 16, 16),
                          ("cornell", pkg.scenes.cornell(W, H, geometric_emitter=True), 16, 16)):
    if scene is None:
        scene = last
    last = scene
    res = {}
    for refill in (0, 7, 15):
        res[refill] = run(scene, refill, s, g)
    print(name, "one-ray-per-lane kernels vs persistent lanes: image max diff %.3g, differing pixels %d, stats %s / %s" % (
        float(np.abs(res[0][1] - res[7][1]).max()), int((np.abs(res[0][1] - res[7][1]).max(-1) > 0).sum()),
        [res[0][2].secondaryCount, res[0][2].deepCount, res[0][2].shadowCount], [res[7][2].secondaryCount, res[7][2].deepCount, res[7][2].shadowCount]), flush=True)
    (h7, i7, s7, t7), (h15, i15, s15, t15) = res[7], res[15]
    same = {k: bool(np.array_equal(h7[k], h15[k])) for k in h7} if isinstance(h7, dict) else None
    if isinstance(h7, dict):
        diff = {k: int((h7[k] != h15[k]).sum()) for k in h7}
    else:
        same = bool(np.array_equal(np.asarray(h7), np.asarray(h15)))
        diff = int((np.asarray(h7) != np.asarray(h15)).sum())
    print(name, "spp", s, "hits equal:", same, "differing:", diff, "image max diff %.3g" % float(np.abs(i7 - i15).max()),
          "primary ms per-lane %.3f packet %.3f" % (s7.primaryTime, s15.primaryTime), "frame s %.4f / %.4f" % (t7, t15), flush=True)
