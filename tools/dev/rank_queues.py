"""Development probe: ms/step of (world, rank 3 or 0, spp) configurations, each as the FIRST context of a fresh process, for the
hardware-queue count the HIP runtime is started with (GPU_MAX_HW_QUEUES)."""
import os, subprocess, sys
child = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
world, rank, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
ctx = pkg.RenderContext(0, rank, world); ctx.init(W, H); scene.upload(ctx)
ctx.set_setting("integrator", "pt"); ctx.set_setting("spp", spp)
for k in range(3): ctx.render_async(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
ctx.wait(); torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t = time.perf_counter()
    for k in range(16): ctx.render_async(scene.camera, pkg.CONVERGE)
    ctx.wait(); torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t) / 16 * 1e3)
print("%.3f" % best)
'''
cfgs = ((1, 0, 128), (1, 0, 32), (1, 0, 8), (1, 0, 1), (8, 3, 128), (8, 3, 32), (8, 3, 8))
print("queues  " + "  ".join("w%d/spp%d" % (w, s) for w, r, s in cfgs))
for q in ("4", "8", "16"):
    env = dict(os.environ, GPU_MAX_HW_QUEUES=q)
    row = []
    for w, r, s in cfgs:
        out = subprocess.run([sys.executable, "-c", child, str(w), str(r), str(s)], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        row.append(out.strip().splitlines()[-1])
    print("%-7s " % q + "  ".join("%9s" % v for v in row), flush=True)
