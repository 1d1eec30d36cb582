import os, sys
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
pkg = load_package()
sc = pkg.scenes.terrain(n=708)
for name, o, t in (("default",(0.0,18.0,-62.0),(0.0,3.6,0.0)), ("offaxis",(3.0,18.0,-62.0),(1.0,3.6,0.0))):
    sc.camera.look_at(o,t)
    ctx = pkg.RenderContext(0); ctx.init(1920,1080); sc.upload(ctx)
    ctx.set_setting("integrator","pt"); ctx.set_setting("spp",8); ctx.set_setting("stage_timing",1)
    ctx.render_frame(sc.camera, pkg.RESET)
    print("==", name, flush=True); sys.stderr.flush()
    os.environ["RFWHIP_DBG"]="1"
    ctx.render_frame(sc.camera, pkg.RESET)
    del os.environ["RFWHIP_DBG"]
    print(ctx.get_stats().as_dict()["primaryTime"], flush=True)
    ctx.destroy()
