#!/usr/bin/env python3
"""bench.py — Msamples/s of the HIP wavefront rendercore on BASELINE.json's headline workload.

  python bench.py --gpus N --steps K --warmup W           (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[2]): synthetic ~1 M-triangle displaced grid (1 002 528 triangles, seed 0x5EED) +
synthetic HDR sky + 8 emissive light triangles + 2 point lights, 1920x1080, wavefront path-tracing integrator
(primary generate -> BVH2 traverse + Moller-Trumbore -> shade with next-event estimation -> compaction -> connect),
MAX_PATH_LENGTH 2 like the reference (settings.h:5).  One *step* = one frame of `--spp` samples per pixel enqueued as
one wavefront batch, plus — for N > 1 — the RCCL gather of the rank-local strips and the de-interleave on rank 0.
Inputs (scene, BVH, path buffers) are resident in HBM before the timed region starts.

The JSON line carries, besides the driver contract fields:
  roofline      dominant kernel (extend = closest-hit traversal): algorithmic bytes per SURVEY §8(d)
                (ray 32 B in [+32 B out for generated primaries], 64 B per popped inner node, 52 B per triangle test,
                16 B hit record out) from an instrumented replay of the same frames, divided by the kernel's mean
                duration measured with hipEvents on the render stream inside the timed region; peak = 8 TB/s HBM3E.
  cpu_baseline  the CPU oracle's restatement of the same integrator ("port") on the host cores, bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy peak)
L2_PEAK_GBS = 34500.0  # same guide, §L2: 8 XCDs x 4 MiB, ~34.5 TB/s aggregate
# a CU's vector L1 serves ONE divergent 16-byte lane-load per clock (profiles/micro/gather_micro.hip: 75 G 128-byte
# records/s with 8 loads each, whatever the table size) — the ceiling the traversal kernels actually run into
LANE_LOADS_PEAK = 600.0e9
# the traversal node: rt::Node4c, 64 bytes = four 16-byte rows (csrc/rt_types.h) — also SURVEY §8(d)'s "64 B per popped inner node"
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4  # wave64 instructions per second, chip: 256 CUs x 4 SIMDs, one 4-clock instruction at a time
NODE_BYTES = 64.0
NODE_ROWS = 4.0


def usable_cores():
    """Host cores this process may really use: min(affinity mask, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def probe_embree():
    """is an Embree the CPU baseline could call discoverable on this box?  (ldconfig cache, the usual library
    directories, CMake package files.)  Returns a short description or None."""
    import ctypes.util
    import glob
    for name in ("embree3", "embree4", "embree"):
        found = ctypes.util.find_library(name)
        if found:
            return found
    for pat in ("/usr/lib*/**/libembree*.so*", "/usr/local/lib*/**/libembree*.so*", "/opt/**/libembree*.so*",
                "/usr/lib*/cmake/embree*", "/usr/local/lib*/cmake/embree*"):
        hit = glob.glob(pat, recursive=True)
        if hit:
            return hit[0]
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp", type=int, default=128,
                    help="samples per pixel per step (one wavefront batch; config 3: >= 64 spp; 128 keeps the launches of an "
                         "8-GPU strip split large: path state = 53 GB of the 288 GB per GPU at N = 1)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--grid", type=int, default=708, help="terrain cells per side (708 -> 1 002 528 triangles)")
    ap.add_argument("--workload", default="terrain", choices=["terrain", "atrium"],
                    help="terrain = BASELINE config 3 (the contract workload); atrium = config 4 (263 k instanced, textured "
                         "triangles) under the same protocol, e.g. for its 8-GPU strip split")
    ap.add_argument("--integrator", default="pt", choices=["pt", "parity"])
    ap.add_argument("--max-depth", type=int, default=2)
    ap.add_argument("--refill", type=int, default=7, help="persistent-lane traversal: bit 0 bounce waves, bit 1 shadow waves, bit 2 primary wave")
    ap.add_argument("--streams", type=int, default=4, help="concurrent sub-batches (HIP streams) per render call")
    ap.add_argument("--overlap", type=int, default=-1, help="connection waves on a second stream per sub-batch: 0 / 1 / -1 = by launch size")
    ap.add_argument("--lds-nodes", type=int, default=-1,
                    help="top-of-tree 4-wide nodes kept in LDS by the traversal kernels (-1: kernel capacity, 0: off)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="N > 1: 1 = stream-ordered present/gather/de-interleave overlapping the next step's kernels; "
                         "0 = host-synchronous gather per step (reports gather_ms_per_step)")
    ap.add_argument("--stage-rates", action="store_true",
                    help="also render one serialised frame (streams = 1) and report Mrays/s per stage; off by default so "
                         "that a kernel trace of the default command holds the timed region's launches only")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="extra rendercore setting(s) for A/B runs, e.g. --set sample_group=1 (recorded in config.settings)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_extend.json"),
                    help="optional PMC-derived HBM bytes per extend launch (see profiles/README.md)")
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch
    from __graft_entry__ import load_package

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    one_device = os.environ.get("RFWHIP_BENCH_ONE_DEVICE") == "1"
    if one_device:
        # development aid for a 1-GPU box: all ranks on GPU 0 and — RCCL refuses two ranks on one device — gloo with host
        # staging for the gather.  Walks through the N > 1 control flow only; its numbers mean nothing.
        local_rank = 0
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run "
                             "--nnodes=1 --nproc-per-node %d ... bench.py --gpus %d" % (args.gpus, args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the rendercore has no CPU path (the CPU oracle is only the baseline)")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    pkg = load_package()
    t0 = time.time()
    if args.workload == "atrium":
        scene = pkg.scenes.atrium(args.width, args.height)
    else:
        scene = pkg.scenes.terrain(n=args.grid, width=args.width, height_px=args.height)
    t_scene = time.time() - t0
    ctx = pkg.RenderContext(device=local_rank, rank=rank, world=world)
    ctx.init(args.width, args.height)
    t0 = time.time()
    scene.upload(ctx)
    t_upload = time.time() - t0
    ctx.set_setting("integrator", args.integrator)
    ctx.set_setting("spp", args.spp)
    ctx.set_setting("max_depth", args.max_depth)
    ctx.set_setting("stage_timing", 1)
    ctx.set_setting("count_traversal", 0)
    ctx.set_setting("refill", args.refill)
    ctx.set_setting("streams", args.streams)
    ctx.set_setting("lds_nodes", args.lds_nodes)
    ctx.set_setting("overlap", args.overlap)
    for kv in args.set:
        k, _, v = kv.partition("=")
        ctx.set_setting(k, v)

    W, H = args.width, args.height
    local_rows = ctx.local_rows()
    local_fb = torch.empty((local_rows, W, 4), dtype=torch.float32, device=dev)
    # the gather lands directly in the [world][local_rows][W] staging image the de-interleave kernel reads
    gathered_flat = torch.empty((world, local_rows, W, 4), dtype=torch.float32, device=dev) if (world > 1 and rank == 0) else None
    gathered = list(gathered_flat.unbind(0)) if gathered_flat is not None else None
    full_fb = torch.empty((H, W, 4), dtype=torch.float32, device=dev) if rank == 0 else None
    gather_ms = []

    # The per-step present -> gather -> de-interleave chain runs on a stream of its own: on torch's default (legacy null)
    # stream the same three small operations cost an 8-GPU rank 0.75 ms of its 14 ms step (measured on one MI355X with
    # tools/project_scaling.py), although the core's streams are non-blocking.
    chain_stream = torch.cuda.Stream(device=dev) if world > 1 else None
    if chain_stream is not None:
        torch.cuda.set_stream(chain_stream)

    def step(k, first):
        # render_frame(camera, status): RESET on the first step of a series, CONVERGE afterwards (context.h:19-23)
        ctx.render_async(scene.camera, pkg.RESET if first else pkg.CONVERGE)
        if world > 1 and one_device:
            ctx.wait()
            ctx.read_local_framebuffer_device(local_fb.data_ptr())
            host = local_fb.cpu()
            parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
            dist.gather(host, parts, dst=0)
            if rank == 0:
                gathered_flat.copy_(torch.stack(parts))
                torch.cuda.synchronize()
                ctx.deinterleave_device(gathered_flat.data_ptr(), full_fb.data_ptr())
        elif world > 1 and args.pipeline:
            # everything stream-ordered, nothing blocks the host: present on torch's current stream (ordered behind the
            # frame's kernels by an event), RCCL gather behind it, de-interleave on the root behind the gather; the
            # next step's kernels run on the core's own streams meanwhile (its accumulate waits for this present).
            ts = torch.cuda.current_stream().cuda_stream
            ctx.read_local_framebuffer_stream(local_fb.data_ptr(), ts)
            dist.gather(local_fb, gathered, dst=0)
            if rank == 0:
                ctx.deinterleave_stream(gathered_flat.data_ptr(), full_fb.data_ptr(), ts)
        elif world > 1:
            ctx.wait()
            t = time.perf_counter()
            ctx.read_local_framebuffer_device(local_fb.data_ptr())
            dist.gather(local_fb, gathered, dst=0)
            if rank == 0:
                torch.cuda.synchronize()
                ctx.deinterleave_device(gathered_flat.data_ptr(), full_fb.data_ptr())
            gather_ms.append((time.perf_counter() - t) * 1e3)

    def fence():
        ctx.wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # no protocol fallback: a failing stream-ordered step fails the run (--pipeline 0 selects the host-synchronous step
    # explicitly, and config.pipeline records which one ran)
    for k in range(args.warmup):
        step(k, k == 0)
    fence()
    for name in ctx.KERNELS:
        ctx.get_kernel_time(name, reset=True)
    ctx.get_counters(reset=True)  # also re-arms the device-side clock of the extend stage
    del gather_ms[:]
    t_start = time.perf_counter()
    for k in range(args.steps):
        step(k, k == 0)
    fence()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_device else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if world == 1:
        ctx.read_framebuffer_device(full_fb.data_ptr())
    kernel_times = {name: ctx.get_kernel_time(name) for name in ctx.KERNELS}
    clock = ctx.get_counters(reset=False)  # extend stage on the device's own clock: first workgroup in .. last workgroup out
    stats = ctx.get_stats().as_dict()

    samples = float(W) * H * args.spp * args.steps
    value = samples / elapsed / 1e6

    # ---- roofline of the dominant kernel (extend) -------------------------------------------------------------------------
    roofline = None
    if not args.no_roofline and rank == 0:
        ext_ms, ext_launches = kernel_times["extend"]
        ctx.set_setting("count_traversal", 1)
        ctx.set_setting("stage_timing", 0)
        ctx.get_counters(reset=True)
        replay = min(args.steps, 4)
        for k in range(replay):  # same sample indices as the first `replay` timed steps => identical rays
            ctx.render_async(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
        ctx.wait()
        cnt = ctx.get_counters(reset=True)
        ctx.set_setting("count_traversal", 0)
        primaries = float(W) * H * args.spp * replay / world
        algo_bytes = (cnt["rays_extend"] * (32 + 16) + primaries * 32 + NODE_BYTES * cnt["inner_extend"] + 52.0 * cnt["tris_extend"])
        # one render call launches the extend kernel (max_depth + 1) x sub-batches times; the sub-batches run on their
        # own HIP streams, so launches of different sub-batches overlap and each launch's duration is stretched by the
        # share of the chip it gets.  Reported: bytes and duration of the average launch as it ran (what rocprofv3
        # shows), and the mean number of kernels in flight (sum of all kernel durations / wall time) beside it.
        launches_per_step = ext_launches / max(1, args.steps)
        bytes_per_launch = algo_bytes / replay / max(1.0, launches_per_step)
        # Duration of the average extend launch.  Two clocks: HIP events recorded on the launch's stream around it (they
        # include the time a launch waits for CU slots while other streams' persistent kernels hold them), and the kernels'
        # own first-workgroup-in / last-workgroup-out timestamps (100 MHz device counter) — the quantity a rocprofv3 kernel
        # trace reports, and the one `achieved` is computed from; both are in the line.
        ms_events = ext_ms / max(1, ext_launches)
        ms_device = clock["extend_ticks"] * 1e-5 / max(1, clock["extend_launches_timed"])
        ms_per_launch = ms_device if clock["extend_launches_timed"] else ms_events
        achieved = bytes_per_launch / (ms_per_launch * 1e-3) / 1e9 if ms_per_launch > 0 else 0.0
        busy_ms = sum(kernel_times[name][0] for name in ctx.KERNELS)
        concurrency = busy_ms / (elapsed * 1e3) if elapsed > 0 else 1.0
        # the same launches alone on the chip: one sub-batch's worth of samples on one stream (what the PMC passes see)
        sub_spp = max(1, args.spp // max(1, min(args.streams, args.spp)))
        ctx.set_setting("streams", 1)
        ctx.set_setting("spp", sub_spp)
        ctx.set_setting("stage_timing", 1)
        ctx.render_frame(scene.camera, pkg.RESET)
        for name in ctx.KERNELS:
            ctx.get_kernel_time(name, reset=True)
        ser_frames = 3
        for k in range(ser_frames):
            ctx.render_frame(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
        ser_ms, ser_launches = ctx.get_kernel_time("extend")
        ser_ms_per_launch = ser_ms / max(1, ser_launches)
        ctx.set_setting("streams", args.streams)
        ctx.set_setting("spp", args.spp)
        # vector-L1 lane-loads of the extend stage: 4 rows per 64-byte 4-wide node fetched from global memory (visits served
        # by the LDS top-of-tree cache cost none), 3 per triangle test, 2 to read the ray
        lane_loads = (NODE_ROWS * (cnt["inner_extend"] - cnt.get("lds_extend", 0)) + 3.0 * cnt["tris_extend"] + 2.0 * cnt["rays_extend"])
        lane_loads_per_launch = lane_loads / replay / max(1.0, launches_per_step)
        pm = {}
        if os.path.exists(args.traffic_json):
            try:
                tj = json.load(open(args.traffic_json))
                if tj.get("spp") == args.spp and tj.get("workload") == scene.name and tj.get("streams") == args.streams:
                    pm = tj
            except Exception:
                pm = {}
        traffic = pm.get("hbm_bytes_per_extend_launch")
        l2_bytes = pm.get("l2_bytes_per_extend_launch")
        valu_insts = pm.get("sq_insts_valu_per_extend_launch")
        ser_s = ser_ms_per_launch * 1e-3
        roofline = {
            "bound": "hbm", "kernel": "k_extend", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
            "algorithmic_bytes_per_launch": bytes_per_launch, "ms_per_launch": ms_per_launch,
            "ms_per_launch_clock": "device (first workgroup in .. last workgroup out)" if clock["extend_launches_timed"] else "hip events",
            "ms_per_launch_hip_events": ms_events, "achieved_hip_events": round(bytes_per_launch / (ms_events * 1e-3) / 1e9, 2) if ms_events > 0 else None,
            "launches_timed": ext_launches, "launches_timed_device_clock": clock["extend_launches_timed"], "launches_per_step": launches_per_step,
            "kernels_in_flight": round(concurrency, 3),
            "achieved_x_kernels_in_flight": round(achieved * max(1.0, concurrency), 2),
            "per_ray": {"inner_nodes": cnt["inner_extend"] / max(1, cnt["rays_extend"]),
                        "inner_nodes_from_lds": cnt.get("lds_extend", 0) / max(1, cnt["rays_extend"]),
                        "triangle_tests": cnt["tris_extend"] / max(1, cnt["rays_extend"]),
                        "rays_per_sample": cnt["rays_extend"] / max(1.0, primaries),
                        "shadow_rays_per_sample": cnt["rays_shadow"] / max(1.0, primaries)},
            "frac_of_measured_copy_peak_6290": round(achieved / 6290.0, 5),
            # The algorithmic byte rate is NOT an HBM rate: the BVH (42 MB of 4-wide nodes + 48 MB of vertices) is served by
            # the vector L1s, the L2s and the Infinity Cache, so it exceeds the HBM peak once a launch has the chip to itself.
            # What the same launches do when serialised, and the ceilings they actually sit under:
            "serialised": {
                "ms_per_launch": round(ser_ms_per_launch, 4), "launches": ser_launches, "spp_per_launch": sub_spp,
                "achieved": round(bytes_per_launch / ser_s / 1e9, 2) if ser_s > 0 else None,
                "frac": round(bytes_per_launch / ser_s / 1e9 / HBM_PEAK_GBS, 5) if ser_s > 0 else None,
                # PMC-measured HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE, profiles/) over the serialised duration
                "hbm_gbs": round(traffic / ser_s / 1e9, 2) if (traffic and ser_s > 0) else None,
                "hbm_frac": round(traffic / ser_s / 1e9 / HBM_PEAK_GBS, 5) if (traffic and ser_s > 0) else None,
                # L2 requests x 128 B (TCC_REQ, profiles/) over the serialised duration, against ~34.5 TB/s
                "l2_gbs": round(l2_bytes / ser_s / 1e9, 2) if (l2_bytes and ser_s > 0) else None,
                "l2_frac": round(l2_bytes / ser_s / 1e9 / L2_PEAK_GBS, 5) if (l2_bytes and ser_s > 0) else None,
                # the binding ceiling: divergent 16-byte lane-loads through the CUs' vector L1s, one per clock per CU
                "l1_lane_loads_per_launch": lane_loads_per_launch,
                "l1_lane_load_rate": round(lane_loads_per_launch / ser_s / 1e9, 2) if ser_s > 0 else None,
                "l1_lane_load_peak": LANE_LOADS_PEAK / 1e9, "l1_lane_load_unit": "G lane-loads/s",
                "l1_lane_load_frac": round(lane_loads_per_launch / ser_s / LANE_LOADS_PEAK, 5) if ser_s > 0 else None,
                # VALU issue (PMC: SQ_INSTS_VALU per launch, profiles/) against one 4-clock wave64 instruction per SIMD and clock
                # group (1024 SIMDs x 2.4 GHz / 4; 2-clock instructions such as v_mov issue faster: profiles/micro/valu_micro.hip)
                "valu_wave_insts_per_launch": valu_insts,
                "valu_issue_rate": round(valu_insts / ser_s / 1e9, 2) if (valu_insts and ser_s > 0) else None,
                "valu_issue_peak": VALU_ISSUE_PEAK / 1e9, "valu_issue_unit": "G wave-instructions/s",
                "valu_issue_frac": round(valu_insts / ser_s / VALU_ISSUE_PEAK, 5) if (valu_insts and ser_s > 0) else None,
                "valu_lanes_active_of_64": pm.get("valu_lanes_active_extend"),
                "binding_ceiling": "VALU issue at the lane utilisation above (divergent traversal); the vector-L1 lane-load rate is the second ceiling",
            },
            "hbm_traffic_gbs": round(traffic / (ms_per_launch * 1e-3) / 1e9, 2) if (traffic and ms_per_launch > 0) else None,
        }

    # ---- rays per second per stage, the way the reference's stats window shows them (imgui_app/main.cpp:279-286): one
    # extra frame with the launches serialised (streams = 1), so every stage's time is its own -------------------------------
    stage_rates = None
    if args.stage_rates and rank == 0 and world == 1:
        ctx.set_setting("streams", 1)
        ctx.set_setting("stage_timing", 1)
        ctx.set_setting("spp", max(1, args.spp // 4))
        ctx.render_frame(scene.camera, pkg.RESET)
        ctx.render_frame(scene.camera, pkg.RESET)
        st1 = ctx.get_stats().as_dict()
        def rate(count, ms):
            return round(st1[count] / (st1[ms] * 1e-3) / 1e6, 1) if st1[ms] > 0 else None
        stage_rates = {"spp": max(1, args.spp // 4), "streams": 1,
                       "primary": rate("primaryCount", "primaryTime"), "secondary": rate("secondaryCount", "secondaryTime"),
                       "deep": rate("deepCount", "deepTime"), "shadow": rate("shadowCount", "shadowTime"),
                       "stage_ms": {k: round(st1[k], 3) for k in ("primaryTime", "secondaryTime", "deepTime", "shadowTime", "shadeTime")}}
        ctx.set_setting("streams", args.streams)
        ctx.set_setting("spp", args.spp)

    # ---- CPU baseline: the oracle (a port, not the reference build) on this box's host cores ----------------------------
    cpu_baseline, parity = None, None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        from __graft_entry__ import load_oracle
        orc = load_oracle()
        cores = usable_cores()
        ref = orc.OracleContext(pkg)
        ref.init(W, H)
        t0 = time.time()
        scene.upload(ref)  # includes the oracle's own (reference-style) BVH build; not timed
        t_build = time.time() - t0
        ref.set_setting("integrator", args.integrator)
        ref.set_setting("max_depth", args.max_depth)
        ref.set_setting("spp", 1)
        ref.set_setting("threads", cores)
        done, spent = 0, 0.0
        while spent < args.cpu_seconds and done < 64:
            t0 = time.perf_counter()
            ref.render_frame(scene.camera, pkg.RESET if done == 0 else pkg.CONVERGE)
            spent += time.perf_counter() - t0
            done += 1
        cpu_value = float(W) * H * done / spent / 1e6
        embree = probe_embree()
        cpu_baseline = {"value": round(cpu_value, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
                        # an Embree harness would be the first choice; none is installed on the box, so the
                        # oracle port is the only CPU line
                        "embree": embree if embree else "not found (ldconfig, /usr, /usr/local, /opt searched)",
                        "sample": "%d full %dx%d frame(s) at 1 spp of the same scene/camera/integrator (%s, depth %d), "
                                  "oracle/rfw_oracle.c with OpenMP, %.1f s; oracle BVH build %.1f s not timed"
                                  % (done, W, H, args.integrator, args.max_depth, spent, t_build)}
        # cross-check while both are here (outside every timed region; the oracle is the checker, not the thing measured):
        # the oracle has just accumulated sample indices 0..done-1 of this very scene — render the same indices on the GPU
        ref_img = ref.framebuffer()[..., :3]
        ref.destroy()
        ctx.set_setting("spp", done)
        ctx.set_setting("stage_timing", 0)
        ctx.render_frame(scene.camera, pkg.RESET)
        hip_img = ctx.framebuffer()[..., :3]
        ctx.set_setting("spp", args.spp)
        dist = np.sqrt(((hip_img.astype(np.float64) - ref_img) ** 2).sum(-1))
        parity = {"samples_per_pixel": done, "tolerance": 3e-2, "frac_gt_3e-2": round(float((dist > 3e-2).mean()), 6),
                  "rmse": round(float(np.sqrt((dist ** 2).mean())), 6),
                  "mean_rel": round(float(abs(hip_img.mean() - ref_img.mean()) / ref_img.mean()), 7),
                  "hip_mean": float(hip_img.mean()), "oracle_mean": float(ref_img.mean())}

    if rank == 0:
        out = {
            "metric": ("Msamples/sec at 1920x1080, 1M-tri scene; 1/2/4/8-GPU tile scaling" if args.workload == "terrain" else
                       "Msamples/sec at 1920x1080, Sponza-scale instanced textured scene (BASELINE config 4); tile scaling"),
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d triangles, %dx%d, %s integrator depth %d, %d spp per step, %d area-light "
                                   "triangles + %d point lights, synthetic 2048x1024 HDR sky"
                                   % (scene.name, scene.triangle_count(), W, H, args.integrator, args.max_depth, args.spp,
                                      len(scene.area_lights), len(scene.point_lights)),
                       "parallelism": ("single GPU, no collective" if world == 1 else
                                       "image strips of 8 rows interleaved over %d ranks, one RCCL gather per step%s"
                                       % (world, " (stream-ordered, overlapping the next step)" if args.pipeline else " (host-synchronous)")),
                       "pipeline": int(args.pipeline) if world > 1 else None,
                       "spp_per_step": args.spp, "streams": args.streams},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
            # per-pixel RGB L2 between the GPU image and the oracle image of the same sample indices (None when the CPU leg is off)
            "parity_vs_cpu_baseline": parity,
            "stage_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in kernel_times.items()},
            "last_frame_counts": {k: stats[k] for k in ("primaryCount", "secondaryCount", "deepCount", "shadowCount")},
            # rays of one step (this rank's strips) over the step time: closest-hit (primary + extension) and any-hit
            "grays_per_s": {"closest_hit": round((stats["primaryCount"] + stats["secondaryCount"] + stats["deepCount"])
                                                 / (elapsed / args.steps) / 1e9, 3),
                            "shadow": round(stats["shadowCount"] / (elapsed / args.steps) / 1e9, 3)},
            "mrays_per_s_per_stage_serialised": stage_rates,
            "gather_ms_per_step": (round(sum(gather_ms) / len(gather_ms), 4) if gather_ms else
                                   (None if (world > 1 and args.pipeline) else 0.0)),
            "setup_s": {"scene": round(t_scene, 2), "upload_and_bvh": round(t_upload, 2)},
            "image_mean": float(full_fb[..., :3].mean().item()),
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
