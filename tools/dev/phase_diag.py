"""Development probe (build with RFWHIP_EXTRA_FLAGS=-DRT_DIAG_PHASES): lane utilisation of the node loop and of the
triangle loop of the traversal kernels, primary wave and bounce waves apart.  With the flag, lds_* counts wave-level
node-loop iterations and tris_* wave-level triangle-loop iterations; inner_* stays per lane."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
scene = pkg.scenes.terrain(n=708, width=W, height_px=H)
ctx = pkg.RenderContext(0); ctx.init(W, H); scene.upload(ctx)
ctx.set_setting("integrator", "pt"); ctx.set_setting("spp", 32); ctx.set_setting("streams", 1); ctx.set_setting("count_traversal", 1)
for kv in sys.argv[1:]:
    k, _, v = kv.partition("="); ctx.set_setting(k, v)
def run(depth):
    ctx.set_setting("max_depth", depth)
    ctx.render_frame(scene.camera, pkg.RESET)
    ctx.get_counters(reset=True)
    ctx.render_frame(scene.camera, pkg.RESET)
    return ctx.get_counters(reset=True)
def show(name, rays, inner, wnode, wtri):
    print("%-8s rays %.1fM  node steps/ray %.2f  node-loop lanes %.1f  wave node iters/ray %.3f  wave tri iters/ray %.3f" % (
        name, rays / 1e6, inner / rays, inner / max(1, wnode), wnode / rays, wtri / rays))
c0, c2 = run(0), run(2)
show("primary", c0["rays_extend"], c0["inner_extend"], c0["lds_extend"], c0["tris_extend"])
show("bounce", c2["rays_extend"] - c0["rays_extend"], c2["inner_extend"] - c0["inner_extend"], c2["lds_extend"] - c0["lds_extend"], c2["tris_extend"] - c0["tris_extend"])
show("shadow", c2["rays_shadow"], c2["inner_shadow"], c2["lds_shadow"], c2["tris_shadow"])
