"""The N > 1 path on CPU: world_size 2 (and 3) over gloo.  Every rank renders its interleaved 8-row strips — with the
host-emulation build of the core (same strip logic as on the GPU) — the rank-local framebuffers are gathered with
torch.distributed and de-interleaved by the root exactly as bench.py does with RCCL; the result must equal the
single-rank image bit for bit (RNG keyed on global pixel ids / global packet order => independent of world)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, integrator, w, h, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import ctypes
    import build_emu
    from __graft_entry__ import load_package
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_package()
    lib = ctypes.CDLL(build_emu.build())
    ctx = pkg._binding.CoreBinding(lib, "rfwhip_", 0, rank, world)
    scene = pkg.scenes.cornell(w, h, geometric_emitter=(integrator == "pt"))
    ctx.init(w, h)
    scene.upload(ctx)
    ctx.set_setting("integrator", integrator)
    ctx.set_setting("spp", 2)
    ctx.render_frame(scene.camera, pkg.RESET)
    ctx.render_frame(scene.camera, pkg.CONVERGE)
    rows = ctx.local_rows()
    local = torch.zeros((rows, w, 4), dtype=torch.float32)
    ctx.read_local_framebuffer_device(local.data_ptr())
    # as bench.py: the gather lands directly in the [world][rows][w] staging image the de-interleave reads
    flat = torch.zeros((world, rows, w, 4), dtype=torch.float32) if rank == 0 else None
    gathered = list(flat.unbind(0)) if rank == 0 else None
    dist.gather(local, gathered, dst=0)
    if rank == 0:
        full = torch.zeros((h, w, 4), dtype=torch.float32)
        ctx.deinterleave_device(flat.data_ptr(), full.data_ptr())
        np.save(out_path, full.numpy())
        # a rank that owns only part of the image must refuse the full-image read
        try:
            ctx.framebuffer()
            raise AssertionError("read_framebuffer on a partial rank must fail")
        except RuntimeError:
            pass
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,integrator,size", [(2, "parity", (100, 70)), (2, "pt", (96, 64)), (3, "pt", (64, 52))])
def test_strip_gather_equals_single_rank(tmp_path, pkg, make_emu, world, integrator, size):
    w, h = size
    out = str(tmp_path / "full.npy")
    port = 29500 + (os.getpid() % 500) + world
    mp.spawn(_worker, args=(world, port, integrator, w, h, out), nprocs=world, join=True)
    got = np.load(out)
    ref = make_emu()
    scene = pkg.scenes.cornell(w, h, geometric_emitter=(integrator == "pt"))
    ref.init(w, h)
    scene.upload(ref)
    ref.set_setting("integrator", integrator)
    ref.set_setting("spp", 2)
    ref.render_frame(scene.camera, pkg.RESET)
    ref.render_frame(scene.camera, pkg.CONVERGE)
    assert np.array_equal(got, ref.framebuffer())


def test_strip_ownership_covers_every_row_once(pkg, make_emu):
    for world in (1, 2, 4, 8):
        owners = np.full(1080, -1)
        for rank in range(world):
            c = make_emu(rank, world)
            c.init(64, 1080)
            rows = c.local_rows()
            assert rows % 8 == 0 and rows == -(-(-(-1080 // 8)) // world) * 8
            for yl in range(rows):
                k = yl // 8                                   # local strip k: forwards in even periods, backwards in odd ones
                y = (k * world + (world - 1 - rank if k & 1 else rank)) * 8 + yl % 8
                if y < 1080:
                    assert owners[y] == -1
                    owners[y] = rank
        assert (owners >= 0).all()
        counts = np.bincount(owners, minlength=world)
        assert counts.max() - counts.min() <= 8    # balanced to one strip
