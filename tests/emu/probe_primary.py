"""Development probe (GPU box): time the wavefront stages per depth for a few camera / spp variants."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
def run(tag, scene, spp, depth=2, frames=6, integ="pt"):
    ctx = pkg.RenderContext(0); ctx.init(1920,1080); scene.upload(ctx)
    ctx.set_setting("integrator",integ); ctx.set_setting("spp",spp); ctx.set_setting("max_depth",depth); ctx.set_setting("stage_timing",1)
    for k in range(2): ctx.render_frame(scene.camera, pkg.RESET)
    acc={}
    for k in range(frames):
        ctx.render_frame(scene.camera, pkg.RESET)
        st=ctx.get_stats().as_dict()
        for key in ("primaryTime","secondaryTime","deepTime","shadowTime","shadeTime","renderTime"): acc[key]=acc.get(key,0)+st[key]/frames
    print(tag, {k: round(v,3) for k,v in acc.items()}, {k: st[k] for k in ("primaryCount","secondaryCount","deepCount","shadowCount")}, flush=True)
    ctx.destroy()
sc = pkg.scenes.terrain(n=708)
run("default spp8", sc, 8)
run("default spp1", sc, 1)
run("default spp8 depth0", sc, 8, depth=0)
sc.camera.look_at((0.0, 60.0, -5.0), (0.0, 0.0, 0.0))
run("topdown spp8", sc, 8)
sc.camera.look_at((3.0, 18.0, -62.0), (1.0, 3.6, 0.0))
run("offaxis spp8", sc, 8)
run("parity spp8", sc, 8, integ="parity")
