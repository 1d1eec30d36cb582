import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
def run(rank, world, spp, chain, timing=0, steps=8, warm=3, h=H):
    ctx = pkg.RenderContext(0, rank, world); ctx.init(W, h); scene.upload(ctx)
    ctx.set_setting("integrator", "pt"); ctx.set_setting("spp", spp); ctx.set_setting("streams", 4); ctx.set_setting("stage_timing", timing)
    rows = ctx.local_rows()
    local = torch.empty((rows, W, 4), dtype=torch.float32, device="cuda:0")
    ts = torch.cuda.current_stream().cuda_stream
    def step(first):
        ctx.render_async(scene.camera, pkg.RESET if first else pkg.CONVERGE)
        if chain: ctx.read_local_framebuffer_stream(local.data_ptr(), ts)
    for k in range(warm): step(k == 0)
    ctx.wait(); torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(steps): step(k == 0)
    ctx.wait(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps * 1e3
    st = ctx.get_stats()
    ctx.destroy()
    return dt, st.primaryCount, st.secondaryCount, st.shadowCount
for (rank, world, spp, chain, timing, steps) in ((0, 1, 64, 0, 0, 8), (0, 1, 64, 0, 1, 8), (0, 1, 64, 0, 0, 16), (0, 1, 64, 0, 1, 16), (0, 2, 128, 0, 1, 16), (0, 2, 128, 0, 0, 16), (0, 8, 128, 0, 1, 32), (0, 8, 128, 0, 0, 32)):
    dt, pc, sc, sh = run(rank, world, spp, chain, timing, steps)
    print("timing", timing, "steps", steps, end=" ")
    print("rank %d/%d spp %d chain %d: %.2f ms/step, %d primaries -> %.1f Mprimaries/s; secondary/primary %.3f shadow/primary %.3f" % (rank, world, spp, chain, dt, pc, pc / dt / 1e3, sc / pc, sh / pc), flush=True)
