"""Multi-GPU below the C ABI (rfwhip_group_* / rfwhip_comm_*, include/rfwhip.h): one host thread drives n contexts, one
gather per presented frame.  CPU tier: the host-emulation build, peer transport (memcpy).  GPU tier: n contexts on ONE
device with the peer transport (RCCL refuses a device twice), bit-equal to the single-context image; the RCCL transport
itself needs >= 2 devices and is skipped otherwise."""
import ctypes

import numpy as np
import pytest


def _render(pkg, target, scene, w, h, settings, frames=2):
    target.init(w, h)
    scene.upload(target)
    for k, v in settings.items():
        target.set_setting(k, v)
    for f in range(frames):
        target.render_async(scene.camera, pkg.RESET if f == 0 else pkg.CONVERGE)
    target.wait()
    return target.framebuffer()


@pytest.mark.parametrize("n", [2, 3, 5])
def test_group_of_emulated_contexts_equals_the_single_context(pkg, make_emu, emu_lib, n):
    """70 x 51: partial tiles, strips that do not divide among the ranks; pt integrator with connections.
    (Scenes: every rank that still has paths at depth 1 must have some at depth 2 — the reference traces the connections of
    a depth only while paths survive it, CUDART/src/Context.cpp:109-120, a per-batch rule here: DESIGN.md §6.)"""
    scene = pkg.scenes.terrain(n=24, width=70, height_px=51)
    settings = {"integrator": "pt", "spp": 4, "max_depth": 2}
    ref = _render(pkg, make_emu(), scene, 70, 51, settings)
    g = pkg._binding.RenderGroup(emu_lib, "rfwhip_", [0] * n, "peer")
    assert g.world == n and g.transport == "peer"
    img = _render(pkg, g, scene, 70, 51, settings)
    for st in g.get_stats():
        assert st.secondaryCount == 0 or st.deepCount > 0
    assert np.array_equal(img, ref)
    # a second gather of the same accumulator is the same image; a resize re-lays the staging out
    assert np.array_equal(g.framebuffer(), ref)
    scene2 = pkg.scenes.cornell(48, 40)
    settings2 = {"integrator": "parity", "spp": 2}
    ref2 = _render(pkg, make_emu(), scene2, 48, 40, settings2)
    assert np.array_equal(_render(pkg, g, scene2, 48, 40, settings2), ref2)
    g.destroy()


@pytest.mark.parametrize("n", [2, 4])
def test_frames_in_flight_hand_out_earlier_frames(pkg, make_emu, emu_lib, n):
    """n frames in flight: render(k), present_async(k % n), present_wait((k + 1) % n) — the image handed out during frame k is
    frame k - n + 1's accumulator, bit for bit (here: the accumulated samples of a converging series)."""
    scene = pkg.scenes.cornell(64, 48, geometric_emitter=True)
    settings = {"integrator": "pt", "spp": 1, "max_depth": 2}
    frames = 6
    want = []
    ref = make_emu()
    ref.init(64, 48)
    scene.upload(ref)
    for k, v in settings.items():
        ref.set_setting(k, v)
    for k in range(frames):
        ref.render_frame(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
        want.append(ref.framebuffer())
    g = pkg._binding.RenderGroup(emu_lib, "rfwhip_", [0, 0], "peer")
    g.init(64, 48)
    scene.upload(g)
    for k, v in settings.items():
        g.set_setting(k, v)
    with pytest.raises(RuntimeError):
        g.present_wait(0)  # nothing presented yet
    with pytest.raises(RuntimeError):
        g.present_async(4)  # RFWHIP_PRESENT_SLOTS slots
    shown = 0
    for k in range(frames):
        g.render_async(scene.camera, pkg.RESET if k == 0 else pkg.CONVERGE)
        g.present_async(k % n)
        if k >= n - 1:
            assert np.array_equal(g.present_wait((k + 1) % n), want[k - n + 1]), k
            shown += 1
    for k in range(frames - n + 1, frames):  # drain
        assert np.array_equal(g.present_wait(k % n), want[k]), k
    assert shown == frames - n + 1
    g.destroy()


def test_group_of_one_and_argument_errors(pkg, make_emu, emu_lib):
    scene = pkg.scenes.cornell(64, 48)
    settings = {"integrator": "parity", "spp": 2}
    ref = _render(pkg, make_emu(), scene, 64, 48, settings)
    g = pkg._binding.RenderGroup(emu_lib, "rfwhip_", [0], "auto")
    assert np.array_equal(_render(pkg, g, scene, 64, 48, settings), ref)
    g.destroy()
    with pytest.raises(RuntimeError):
        pkg._binding.RenderGroup(emu_lib, "rfwhip_", [], "peer")
    with pytest.raises(RuntimeError):
        pkg._binding.RenderGroup(emu_lib, "rfwhip_", [0, 0], "rccl")  # the emulation build has no RCCL
    # the comm front end needs RCCL as soon as the world is larger than one
    e = make_emu(0, 2)
    comm = ctypes.c_void_p()
    emu_lib.rfwhip_comm_create.restype = ctypes.c_int
    emu_lib.rfwhip_comm_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    assert emu_lib.rfwhip_comm_create(e._ctx, None, ctypes.byref(comm)) != 0


def test_comm_of_world_one_gathers_into_a_caller_buffer(pkg, make_emu, emu_lib):
    """rfwhip_comm_* with world 1 is the degenerate gather: present + copy, no transport."""
    scene = pkg.scenes.cornell(64, 48)
    e = make_emu()
    ref = _render(pkg, e, scene, 64, 48, {"integrator": "parity", "spp": 1}, frames=1)
    vp = ctypes.c_void_p
    for name, args in (("rfwhip_comm_create", [vp, vp, ctypes.POINTER(vp)]), ("rfwhip_comm_gather", [vp, vp]), ("rfwhip_comm_wait", [vp])):
        getattr(emu_lib, name).restype, getattr(emu_lib, name).argtypes = ctypes.c_int, args
    emu_lib.rfwhip_comm_destroy.restype, emu_lib.rfwhip_comm_destroy.argtypes = None, [vp]
    comm = vp()
    assert emu_lib.rfwhip_comm_create(e._ctx, None, ctypes.byref(comm)) == 0
    out = np.zeros((48, 64, 4), np.float32)  # (emulation: "device" memory is host memory)
    assert emu_lib.rfwhip_comm_gather(comm, out.ctypes.data) == 0
    assert emu_lib.rfwhip_comm_wait(comm) == 0
    assert np.array_equal(out, ref)
    emu_lib.rfwhip_comm_destroy(comm)


@pytest.mark.gpu
@pytest.mark.parametrize("n,integrator", [(2, "pt"), (4, "pt"), (8, "parity")])
def test_group_on_one_device_equals_the_single_context_gpu(pkg, make_hip, n, integrator):
    """n contexts on cuda:0 through rfwhip_group_* (peer transport): render -> gather pipelined over four frames, bit-equal
    to one context; the image also sits in the root's device buffer."""
    import torch
    scene = pkg.scenes.cornell(480, 270, geometric_emitter=(integrator == "pt"))
    settings = {"integrator": integrator, "spp": 4, "max_depth": 2}
    ref = _render(pkg, make_hip(), scene, 480, 270, settings, frames=4)
    g = pkg.render_group([0] * n, "peer")
    g.init(480, 270)
    scene.upload(g)
    for k, v in settings.items():
        g.set_setting(k, v)
    for f in range(4):  # nothing blocks the host between the frames and their gathers
        g.render_async(scene.camera, pkg.RESET if f == 0 else pkg.CONVERGE)
        g.gather()
    g.wait()
    ptr, dev = g.framebuffer_device()
    assert dev == 0
    host = np.empty((270, 480, 4), np.float32)
    torch.cuda.synchronize()
    from ctypes import c_void_p
    hip = ctypes.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(c_void_p(host.ctypes.data), c_void_p(ptr), host.nbytes, 2) == 0  # hipMemcpyDeviceToHost
    assert np.array_equal(host, ref)
    assert np.array_equal(g.framebuffer(), ref)
    # frames in flight: the pinned host image of slot 1 is the same frame
    g.present_async(1)
    assert np.array_equal(g.present_wait(1), ref)
    g.destroy()


def _moving_cameras(scene, frames):
    import copy
    cams = []
    for k in range(frames):
        cam = copy.deepcopy(scene.camera)
        x, y, z = cam.position
        cam.position = (x + 0.35 * k, y + 0.11 * (k % 3), z + 0.2 * k)
        cams.append(cam)
    return cams


def _frames_in_flight_do_not_tear(pkg, make_one, group, w, h, n_slots, frames, settings):
    """Every frame a NEW camera and a RESET: a presented image that holds strips of two frames (the write-after-read hazard on
    the root's staging image when a fast rank pushes frame k + 1 before the root has de-interleaved frame k) differs from the
    single-context image of its camera."""
    # (terrain: every rank keeps paths alive at every depth — the per-batch connection rule of DESIGN.md §6 never differs)
    scene = pkg.scenes.terrain(n=24, width=w, height_px=h)
    cams = _moving_cameras(scene, frames)
    ref = make_one()
    ref.init(w, h)
    scene.upload(ref)
    group.init(w, h)
    scene.upload(group)
    for k, v in settings.items():
        ref.set_setting(k, v), group.set_setting(k, v)
    want = []
    for cam in cams:
        ref.render_frame(cam, pkg.RESET)
        want.append(ref.framebuffer())
    assert not np.array_equal(want[0], want[1])
    for k, cam in enumerate(cams):
        group.render_async(cam, pkg.RESET)
        group.present_async(k % n_slots)
        if k >= n_slots - 1:
            assert np.array_equal(group.present_wait((k + 1) % n_slots), want[k - n_slots + 1]), k
    for k in range(frames - n_slots + 1, frames):
        assert np.array_equal(group.present_wait(k % n_slots), want[k]), k


def test_frames_in_flight_with_a_moving_camera_do_not_tear(pkg, make_emu, emu_lib):
    g = pkg._binding.RenderGroup(emu_lib, "rfwhip_", [0, 0, 0], "peer")
    _frames_in_flight_do_not_tear(pkg, make_emu, g, 64, 48, 3, 7, {"integrator": "pt", "spp": 1, "max_depth": 2})
    g.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("n,slots", [(4, 4), (8, 2)])
def test_frames_in_flight_with_a_moving_camera_do_not_tear_gpu(pkg, make_hip, n, slots):
    """n contexts on one device (the hardware schedules their streams in any order: the ranks are as imbalanced as it gets),
    peer transport, frames handed out late, the camera moving every frame."""
    g = pkg.render_group([0] * n, "peer")
    _frames_in_flight_do_not_tear(pkg, make_hip, g, 480, 270, slots, 14, {"integrator": "pt", "spp": 2, "max_depth": 2})
    g.destroy()


@pytest.mark.gpu
def test_group_over_rccl_needs_two_devices(pkg, make_hip):
    """The RCCL transport: ncclSend / ncclRecv between the ranks of one process.  Needs >= 2 visible devices."""
    import torch
    if torch.cuda.device_count() < 2:
        with pytest.raises(RuntimeError):
            pkg.render_group([0, 0], "rccl")  # a device listed twice is refused loudly, not downgraded
        pytest.skip("one visible device: the RCCL transport cannot be exercised here")
    scene = pkg.scenes.cornell(480, 270, geometric_emitter=True)
    settings = {"integrator": "pt", "spp": 4, "max_depth": 2}
    ref = _render(pkg, make_hip(), scene, 480, 270, settings, frames=3)
    g = pkg.render_group([0, 1], "rccl")
    assert g.transport == "rccl"
    img = _render(pkg, g, scene, 480, 270, settings, frames=3)
    assert np.array_equal(img, ref)
    g.destroy()


@pytest.mark.gpu
def test_two_process_comm_gather_needs_two_devices(pkg):
    """rfwhip_comm_* end to end, one process per device: `bench.py --gpus 2` under torch.distributed.run (RCCL send / recv issued
    by librfwhip.so).  Needs >= 2 visible devices; on one device the same launch is walked through with
    RFWHIP_BENCH_ONE_DEVICE=1 + RFWHIP_BENCH_TRY_COMM=1: RCCL refuses the duplicate device and bench.py must fall back to the
    torch gather LOUDLY (config.gather says so) and still produce the right image mean."""
    import json
    import os
    import subprocess
    import sys
    import torch
    from conftest import ROOT
    two = torch.cuda.device_count() >= 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if not two:
        env.update(RFWHIP_BENCH_ONE_DEVICE="1", RFWHIP_BENCH_TRY_COMM="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--spp", "8",
           "--width", "480", "--height", "270", "--grid", "64", "--no-roofline", "--no-cpu-baseline"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0
    if two:
        assert d["config"]["gather"] == "comm", d["config"]["gather"]
    else:
        assert d["config"]["gather"].startswith("torch (rfwhip_comm_create failed"), d["config"]["gather"]
    # the same two steps on one rank: the strip split does not change the image
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--spp", "8", "--width", "480",
                          "--height", "270", "--grid", "64", "--no-roofline", "--no-cpu-baseline"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert abs(d["image_mean"] - d1["image_mean"]) <= 1e-6 * abs(d1["image_mean"]), (d["image_mean"], d1["image_mean"])
