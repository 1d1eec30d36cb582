import sys, time, json
sys.path.insert(0, "/root/repo")
import torch
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
scene = pkg.scenes.atrium(W, H)
for lds in (0, -1):
    ctx = pkg.RenderContext(device=0)
    ctx.init(W, H); scene.upload(ctx)
    for k, v in {"integrator": "pt", "spp": 32, "max_depth": 2, "stage_timing": 1, "lds_nodes": lds}.items():
        ctx.set_setting(k, v)
    for k in range(2):
        ctx.render_async(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
    ctx.wait()
    for name in ctx.KERNELS: ctx.get_kernel_time(name, reset=True)
    t0 = time.perf_counter()
    for k in range(4): ctx.render_async(scene.camera, pkg.CONVERGE)
    ctx.wait()
    el = time.perf_counter() - t0
    print("lds_nodes", lds, round(W*H*32*4/el/1e6, 1), "Msamples/s", {n: round(ctx.get_kernel_time(n)[0]/4, 2) for n in ctx.KERNELS}, flush=True)
    ctx.set_setting("count_traversal", 1); ctx.get_counters(reset=True)
    ctx.render_frame(scene.camera, pkg.RESET)
    c = ctx.get_counters(reset=True)
    print({k: round(c[k]/max(1,c["rays_extend"]),2) for k in ("inner_extend","tris_extend")}, {k: round(c[k]/max(1,c["rays_shadow"]),2) for k in ("inner_shadow","tris_shadow")})
    ctx.destroy()
