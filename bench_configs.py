#!/usr/bin/env python3
"""Side measurements of the BASELINE.json configs bench.py does not time, at their full sizes on ONE MI355X (the driver's
contract bench is bench.py = config 3 with the pt integrator).  Prints one JSON line per config.

  config 2  Cornell scene, 1920x1080, 64 spp, parity integrator (the L2-checked algorithm): Msamples/s
  config 3  the 1 M-triangle terrain with the parity integrator (SURVEY §8d asks for both integrators)

  config 4  atrium (263 288 instanced triangles, 46 instances, 25 materials, 12 textures with mips), 1920x1080,
            pt integrator depth 2: Msamples/s
  config 5  two-bone skinned tube (30 720 triangles), 1920x1080, every frame: new pose -> set_mesh with unchanged
            counts -> device refit (BVH2 boxes bottom-up + 4-wide node refresh) -> TLAS -> 1 spp pt frame;
            ms per frame with the split the reference's RenderStats reports
"""
import argparse
import json
import time

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--spp", type=int, default=16)
    args = ap.parse_args()
    import torch
    from __graft_entry__ import load_package
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X")
    pkg = load_package()
    W, H = 1920, 1080

    # ---- configs 2 and 3 with the parity integrator (the algorithm that is L2-checked against the oracle) -------------
    for cfg, scene in ((2, pkg.scenes.cornell(W, H)), ("3 (parity integrator)", pkg.scenes.terrain(n=708, width=W, height_px=H))):
        ctx = pkg.RenderContext(device=0)
        ctx.init(W, H)
        scene.upload(ctx)
        for k, v in {"integrator": "parity", "jitter": "xor128", "spp": 64}.items():
            ctx.set_setting(k, v)
        for k in range(5):
            ctx.render_async(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
        ctx.wait()
        steps = 6
        t0 = time.perf_counter()
        for k in range(steps):
            ctx.render_async(scene.camera, pkg.CONVERGE)
        ctx.wait()
        el = time.perf_counter() - t0
        st = ctx.get_stats().as_dict()
        print(json.dumps({"config": cfg, "workload": "%s: %d triangles, 1920x1080, parity integrator (1 primary ray + one "
                          "shadow ray per light), 64 spp per step, xor128 jitter" % (scene.name, scene.triangle_count()),
                          "metric": "Msamples/s", "value": round(W * H * 64 * steps / el / 1e6, 1),
                          "ms_per_step": round(el / steps * 1e3, 3),
                          "rays_last_frame": {k: st[k] for k in ("primaryCount", "shadowCount")}}))
        del ctx

    # ---- config 4 -------------------------------------------------------------------------------------------------
    scene = pkg.scenes.atrium(W, H)
    ctx = pkg.RenderContext(device=0)
    ctx.init(W, H)
    t0 = time.time()
    scene.upload(ctx)
    t_up = time.time() - t0
    for k, v in {"integrator": "pt", "spp": args.spp, "max_depth": 2, "stage_timing": 1}.items():
        ctx.set_setting(k, v)
    for k in range(2):
        ctx.render_async(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
    ctx.wait()
    steps = 6
    t0 = time.perf_counter()
    for k in range(steps):
        ctx.render_async(scene.camera, pkg.CONVERGE)
    ctx.wait()
    el = time.perf_counter() - t0
    st = ctx.get_stats().as_dict()
    print(json.dumps({"config": 4, "workload": "atrium: %d triangles in %d instances, %d materials, %d textures" % (
        scene.triangle_count(), len(scene.instances), len(scene.host_materials), len(scene.textures)),
        "metric": "Msamples/s", "value": round(W * H * args.spp * steps / el / 1e6, 1), "spp_per_step": args.spp,
        "ms_per_step": round(el / steps * 1e3, 3), "upload_and_bvh_s": round(t_up, 3),
        "rays_last_frame": {k: st[k] for k in ("primaryCount", "secondaryCount", "deepCount", "shadowCount")}}))
    del ctx

    # ---- config 5 -------------------------------------------------------------------------------------------------
    scene = pkg.scenes.skinned_tube(0.0, width=W, height=H)
    ctx = pkg.RenderContext(device=0)
    ctx.init(W, H)
    scene.upload(ctx)
    for k, v in {"integrator": "pt", "spp": 1, "max_depth": 2, "stage_timing": 1}.items():
        ctx.set_setting(k, v)
    ctx.render_frame(scene.camera, pkg.RESET)
    t_pose = t_set = t_render = 0.0
    for name in ctx.KERNELS:
        ctx.get_kernel_time(name, reset=True)
    t_all = time.perf_counter()
    for f in range(1, args.frames + 1):
        t0 = time.perf_counter()
        pose = pkg.scenes.skinned_tube(float(f), width=W, height=H).meshes[0]
        t1 = time.perf_counter()
        ctx.set_mesh(0, pose["vertices"], pose["triangles"], pose["indices"])
        ctx.update()
        t2 = time.perf_counter()
        ctx.render_frame(scene.camera, pkg.RESET)
        t3 = time.perf_counter()
        t_pose += t1 - t0
        t_set += t2 - t1
        t_render += t3 - t2
    total = time.perf_counter() - t_all
    kt = {name: ctx.get_kernel_time(name) for name in ctx.KERNELS}
    n = args.frames
    print(json.dumps({"config": 5, "workload": "skinned tube: %d triangles, 1920x1080, 1 spp pt depth 2 per frame" % (
        scene.meshes[0]["triangles"].shape[0]),
        "metric": "ms/frame (device side: set_mesh + update + render)", "value": round((t_set + t_render) / n * 1e3, 3),
        "fps_device_side": round(n / (t_set + t_render), 1),
        "split_ms": {"host_pose_numpy": round(t_pose / n * 1e3, 3), "set_mesh_update": round(t_set / n * 1e3, 3),
                     "render_1spp": round(t_render / n * 1e3, 3)},
        "kernel_ms_per_frame": {k: round(v[0] / n, 4) for k, v in kt.items()},
        "frames": n, "wall_ms_per_frame_including_host_pose": round(total / n * 1e3, 3)}))
    del ctx

    # ---- config 5 with the skin on the device: the host only sends the joint matrices -----------------------------------
    scene = pkg.scenes.skinned_tube(0.0, width=W, height=H)
    v, idx, vn, joints, weights = pkg.scenes.skinned_tube_rig()
    scene.meshes[0]["triangles"] = pkg.scenes.make_triangles(v, idx, normals=vn,
                                                             material=scene.meshes[0]["triangles"]["material"][0])
    ctx = pkg.RenderContext(device=0)
    ctx.init(W, H)
    scene.upload(ctx)
    for k, v_ in {"integrator": "pt", "spp": 1, "max_depth": 2, "stage_timing": 1}.items():
        ctx.set_setting(k, v_)
    ctx.set_mesh_skin(0, joints, weights, vn)
    ctx.render_frame(scene.camera, pkg.RESET)
    for name in ctx.KERNELS:
        ctx.get_kernel_time(name, reset=True)
    t_pose = t_render = 0.0
    t_all = time.perf_counter()
    for f in range(1, args.frames + 1):
        t0 = time.perf_counter()
        ctx.pose_mesh(0, pkg.scenes.skinned_tube_joint_matrices(float(f)))
        ctx.update()
        t1 = time.perf_counter()
        ctx.render_frame(scene.camera, pkg.RESET)
        t2 = time.perf_counter()
        t_pose += t1 - t0
        t_render += t2 - t1
    total = time.perf_counter() - t_all
    kt = {name: ctx.get_kernel_time(name) for name in ctx.KERNELS}
    print(json.dumps({"config": "5 (device skinning)", "workload": "same tube, rfwhip_set_mesh_skin once + rfwhip_pose_mesh per frame",
        "metric": "ms/frame (wall, everything)", "value": round(total / n * 1e3, 3), "fps": round(n / total, 1),
        "split_ms": {"pose_mesh_update": round(t_pose / n * 1e3, 3), "render_1spp": round(t_render / n * 1e3, 3)},
        "kernel_ms_per_frame": {k: round(v_[0] / n, 4) for k, v_ in kt.items()}, "frames": n}))

    # ---- config 5 on the asset it names: CesiumMan (assets/models/CesiumMan), from the committed fixture (bind pose, rig, three
    # poses' joint matrices: tests/golden/asset_cesiumman.npz) — skin on the device, the host sends 19 joint matrices per frame ------
    import os
    fx_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "asset_cesiumman.npz")
    if os.path.exists(fx_path):
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
        from test_assets import _rig_scene
        fx = np.load(fx_path)
        scene = _rig_scene(pkg, fx["positions"], fx["normals"], fx["indices"], fx["node_transform"], W, H)
        ctx = pkg.RenderContext(device=0)
        ctx.init(W, H)
        scene.upload(ctx)
        for k, v_ in {"integrator": "pt", "spp": 1, "max_depth": 2, "stage_timing": 1}.items():
            ctx.set_setting(k, v_)
        ctx.set_mesh_skin(0, fx["joints"], fx["weights"], fx["normals"])
        ctx.render_frame(scene.camera, pkg.RESET)
        for name in ctx.KERNELS:
            ctx.get_kernel_time(name, reset=True)
        t_pose = t_render = 0.0
        t_all = time.perf_counter()
        for f in range(args.frames):
            t0 = time.perf_counter()
            ctx.pose_mesh(0, fx["joint_matrices"][f % len(fx["joint_matrices"])])
            ctx.update()
            t1 = time.perf_counter()
            ctx.render_frame(scene.camera, pkg.RESET)
            t2 = time.perf_counter()
            t_pose += t1 - t0
            t_render += t2 - t1
        total = time.perf_counter() - t_all
        kt = {name: ctx.get_kernel_time(name) for name in ctx.KERNELS}
        print(json.dumps({"config": "5 (CesiumMan, device skinning)",
            "workload": "CesiumMan (3273 vertices, 4672 triangles, 19 joints) on a floor, 1920x1080, 1 spp pt depth 2 per frame",
            "metric": "ms/frame (wall, everything)", "value": round(total / n * 1e3, 3), "fps": round(n / total, 1),
            "split_ms": {"pose_mesh_update": round(t_pose / n * 1e3, 3), "render_1spp": round(t_render / n * 1e3, 3)},
            "kernel_ms_per_frame": {k: round(v_[0] / n, 4) for k, v_ in kt.items()}, "frames": n}))


if __name__ == "__main__":
    main()
