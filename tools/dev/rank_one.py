"""Development probe: rank R of an N-GPU strip split alone on cuda:0 in a FRESH process, one settings variant.
usage: rank_one.py RANK WORLD SPP key=value ...   -> ms/step (16 timed steps, pipelined, local present per step)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
pkg = load_package()
rank, world, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
W, H = 1920, 1080
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
ctx = pkg.RenderContext(0, rank, world); ctx.init(W, H); scene.upload(ctx)
ctx.set_setting("integrator", "pt"); ctx.set_setting("spp", spp)
for kv in sys.argv[4:]:
    k, _, v = kv.partition("="); ctx.set_setting(k, v)
ctx.update()
local = torch.empty((ctx.local_rows(), W, 4), dtype=torch.float32, device="cuda:0")
side = torch.cuda.Stream(); torch.cuda.set_stream(side)
ts = torch.cuda.current_stream().cuda_stream
def step(first):
    ctx.render_async(scene.camera, pkg.RESET if first else pkg.CONVERGE)
    ctx.read_local_framebuffer_stream(local.data_ptr(), ts)
for k in range(4): step(k == 0)
ctx.wait(); torch.cuda.synchronize()
t = time.perf_counter()
for k in range(16): step(False)
ctx.wait(); torch.cuda.synchronize()
print(" ".join(sys.argv[1:]), "%.3f ms/step" % ((time.perf_counter() - t) / 16 * 1e3), flush=True)
