"""Development probe (1-GPU box): what would one rank of an N-GPU strip split cost per step?

Runs rank r of world N alone on cuda:0 (same scene, same camera, its interleaved strips only), including everything
bench.py does per step on a rank except the RCCL gather itself (wait, local present into a torch tensor, and on
"rank 0" the de-interleave of a full-size gathered buffer).  Prints the slowest rank's ms/step and the projected
efficiency  t1 / (N * tN)  — a projection, NOT a measurement of the 8-GPU run (the driver does that)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pipeline = int(sys.argv[3]) if len(sys.argv) > 3 else 1
# PROJ_WORKLOAD=atrium: BASELINE config 4 (the config the reference shards over 8 GPUs) instead of the bench terrain
scene = pkg.scenes.atrium(W, H) if os.environ.get("PROJ_WORKLOAD") == "atrium" else pkg.scenes.terrain(n=708, width=W, height_px=H)

def run(rank, world, steps=10, warm=3):
    ctx = pkg.RenderContext(0, rank, world); ctx.init(W, H); scene.upload(ctx)
    ctx.set_setting("integrator", "pt"); ctx.set_setting("spp", spp); ctx.set_setting("streams", streams)
    for kv in os.environ.get("PROJ_SET", "").split():   # e.g. PROJ_SET="ring=4"
        k, _, v = kv.partition("="); ctx.set_setting(k, v)
    rows = ctx.local_rows()
    local = torch.empty((rows, W, 4), dtype=torch.float32, device="cuda:0")
    flat = torch.empty((world, rows, W, 4), dtype=torch.float32, device="cuda:0")
    full = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
    side = None if os.environ.get("NULL_STREAM") else torch.cuda.Stream()   # (bench.py: the chain has its own stream)
    if side is not None:
        torch.cuda.set_stream(side)
    ts = torch.cuda.current_stream().cuda_stream
    mode = os.environ.get("CHAIN", "full")
    def step(first):
        ctx.render_async(scene.camera, pkg.RESET if first else pkg.CONVERGE)
        if os.environ.get("NO_CHAIN"):
            pass
        elif world > 1 and pipeline:
            ctx.read_local_framebuffer_stream(local.data_ptr(), ts)
            if mode != "present":
                flat[rank].copy_(local, non_blocking=True)   # stands in for the gather's landing copy
            if rank == 0 and mode == "full":
                ctx.deinterleave_stream(flat.data_ptr(), full.data_ptr(), ts)
        elif world > 1:
            ctx.wait()
            ctx.read_local_framebuffer_device(local.data_ptr())
            flat[rank].copy_(local)                      # stands in for the gather's landing copy
            if rank == 0:
                torch.cuda.synchronize()
                ctx.deinterleave_device(flat.data_ptr(), full.data_ptr())
    for k in range(warm): step(k == 0)
    ctx.wait(); torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(steps): step(k == 0)
    ctx.wait(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps * 1e3
    ctx.destroy()
    return dt

import json, subprocess
if len(sys.argv) > 5 and sys.argv[5] == "--one":
    # child: one rank, one process
    r, w = int(sys.argv[6]), int(sys.argv[7])
    print("RANKMS %.6f" % run(r, w), flush=True)
    sys.exit(0)

def run(rank, world):  # noqa: F811 — every rank in its own process
    out = subprocess.run([sys.executable, __file__, str(spp), str(streams), str(pipeline), "0", "--one", str(rank), str(world)],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    return float([l for l in out.splitlines() if l.startswith("RANKMS")][0].split()[1])

t1 = run(0, 1)
print("spp", spp, "streams", streams, "pipeline", pipeline, "world 1: %.3f ms/step" % t1, flush=True)
worlds = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [2, 4, 8]
rec = {"what": "single-GPU PROJECTION of the strip split (every rank of world N run alone on one MI355X with bench.py's per-step "
               "chain, a stream-ordered copy standing in for the RCCL gather; each rank in a FRESH process like a real launch: a second "
               "context in one process runs ~10 % slower than the first) — not a measurement of an N-GPU run",
       "workload": scene.name, "spp_per_step": spp, "streams": streams, "pipeline": pipeline, "ms_per_step_world1": round(t1, 4),
       "msamples_per_s_world1": round(W * H * spp / t1 / 1e3, 1), "worlds": {}}
for world in worlds:
    ts = [run(r, world) for r in range(world)]
    tn = max(ts)
    print("world %d: slowest rank %.3f ms/step (min %.3f)  projected efficiency %.3f  ranks %s" % (world, tn, min(ts), t1 / (world * tn), " ".join("%.2f" % t for t in ts)), flush=True)
    rec["worlds"][str(world)] = {"slowest_rank_ms": round(tn, 4), "fastest_rank_ms": round(min(ts), 4),
                                 "projected_efficiency": round(t1 / (world * tn), 4),
                                 "projected_msamples_per_s": round(W * H * spp / tn / 1e3, 1)}
print("JSON " + json.dumps(rec), flush=True)
