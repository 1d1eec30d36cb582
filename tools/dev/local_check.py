"""Development probe: k_frame_local (setting local_frames) against the multi-launch path — images bit for bit, ray counts, ms per pipelined frame."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
def scene_of(name):
    if name.startswith("terrain") and len(name) > 7:  # terrainN: N cells per side
        return pkg.scenes.terrain(n=int(name[7:]), width=W, height_px=H)
    return {"terrain": lambda: pkg.scenes.terrain(n=708, width=W, height_px=H), "atrium": lambda: pkg.scenes.atrium(W, H),
            "cornell": lambda: pkg.scenes.cornell(W, H, geometric_emitter=True)}[name]()
for name in sys.argv[1].split(","):
    scene = scene_of(name)
    for spp in [int(x) for x in os.environ.get('LOCAL_SPP', '1,2,4').split(',')]:
        out = []
        for local in (0, 1 << 26):
            ctx = pkg.RenderContext(0); ctx.init(W, H); scene.upload(ctx)
            ctx.set_setting("integrator", "pt"); ctx.set_setting("spp", spp); ctx.set_setting("local_frames", local)
            ctx.render_frame(scene.camera, pkg.RESET)            # (the first frame after an update is always multi-launch)
            for k in range(3):
                ctx.render_frame(scene.camera, pkg.CONVERGE)
            st = ctx.get_stats()
            img = ctx.framebuffer()
            for k in range(20): ctx.render_async(scene.camera, pkg.CONVERGE)
            ctx.wait()
            t = time.perf_counter()
            for k in range(100): ctx.render_async(scene.camera, pkg.CONVERGE)
            ctx.wait()
            ms = (time.perf_counter() - t) / 100 * 1e3
            out.append((img, (st.primaryCount, st.secondaryCount, st.deepCount, st.shadowCount), ms, ctx.get_setting("local_mispredicted")))
            ctx.destroy()
        same = np.array_equal(out[0][0], out[1][0])
        d = np.abs(out[0][0] - out[1][0])
        print("%s spp %d: multi-launch %.3f ms  local %.3f ms  image %s (max diff %.3g, %d px)  counts %s %s  mispredicted %s" % (
            name, spp, out[0][2], out[1][2], "BIT-EQUAL" if same else "DIFFERS", d.max(), int((d.max(-1) > 0).sum()), out[0][1], "==" if out[0][1] == out[1][1] else "!= %s" % (out[1][1],), out[1][3]), flush=True)
