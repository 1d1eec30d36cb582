"""The C ABI library: loads, exports every symbol include/rfwhip.h declares, struct sizes match the reference's
layouts, and — on a box without a GPU — the product refuses to run instead of falling back to anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_library_is_built_and_exports_every_declared_symbol(pkg):
    import __graft_entry__ as g
    assert os.path.exists(pkg.LIB_PATH), "run __graft_entry__.build() first"
    lib = C.CDLL(pkg.LIB_PATH)
    syms = g.declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), s
    lib.rfwhip_version.restype = C.c_char_p
    assert b"gfx950" in lib.rfwhip_version()


def test_header_cites_the_reference_interface():
    text = open(os.path.join(ROOT, "include", "rfwhip.h")).read()
    for needle in ("context.h:74-111", "export.h:8-15", "context.h:94", "context.h:98", "context.h:108"):
        assert needle in text


def test_plugin_exports_the_two_factory_symbols(pkg):
    plugin = os.path.join(os.path.dirname(pkg.LIB_PATH), "HipRT.so")
    if not os.path.exists(plugin):
        pytest.skip("plugin not built")
    lib = C.CDLL(plugin)
    assert hasattr(lib, "createRenderContext") and hasattr(lib, "destroyRenderContext")  # export.h:8-15


def test_pod_sizes(pkg):
    abi = pkg.abi
    assert abi.TRIANGLE_DTYPE.itemsize == 160 and abi.MATERIAL_DTYPE.itemsize == 192
    assert abi.AREA_LIGHT_DTYPE.itemsize == 96 and abi.POINT_LIGHT_DTYPE.itemsize == 32
    assert abi.SPOT_LIGHT_DTYPE.itemsize == 48 and abi.DIRECTIONAL_LIGHT_DTYPE.itemsize == 32
    assert C.sizeof(abi.CameraView) == 56 and C.sizeof(abi.RenderStats) == 48 and C.sizeof(abi.Mesh) == 56
    # field offsets the kernels rely on (structs.h:35-60)
    t = abi.TRIANGLE_DTYPE
    assert t.fields["lightTriIdx"][1] == 12 and t.fields["material"][1] == 28 and t.fields["vN0"][1] == 32
    assert t.fields["area"][1] == 92 and t.fields["LOD"][1] == 108 and t.fields["vertex0"][1] == 112
    m = abi.MATERIAL_DTYPE
    assert m.fields["flags"][1] == 12 and m.fields["parameters"][1] == 16 and m.fields["map"][1] == 32


def test_no_gpu_means_loud_failure_not_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError) as e:
        pkg.RenderContext(device=0)
    assert "no HIP device" in str(e.value) or "hip" in str(e.value).lower()


def test_material_packing_matches_reference_rules(pkg):
    sc = pkg.scenes
    mats, ids = sc.pack_materials([sc.host_material(color=(0.5, 0.25, 2.0), roughness=1.0, metallic=0.2, eta=1.5)], [])
    m = mats[0]
    assert np.array_equal(m["diffuse"], np.array([0.5, 0.25, 2.0], np.float16))
    p0, p2 = int(m["parameters"][0]), int(m["parameters"][2])
    assert p0 & 0xFF == int(0.2 * 255) and (p0 >> 24) == 255       # metallic | ... | roughness (material_list.cpp:337)
    assert (p2 >> 24) == int(1.5 * 0.5 * 255) and ((p2 >> 8) & 0xFF) == 255  # eta*0.5, clearcoatGloss default 1
    assert m["flags"] & (1 << 0) and m["flags"] & (1 << 11) and not (m["flags"] & (1 << 2))
    assert ids[0]["texture"][0] == -1


def test_bench_counters_are_tied_to_the_sources_they_were_taken_on(tmp_path):
    """bench.py reports PMC-derived figures only when profiles/stage_counters.json was taken on the sources that run
    (csrc_hash): a stale file is recognised, a matching one accepted, a file of another workload ignored."""
    import json
    import bench
    h = bench.csrc_hash()
    assert len(h) == 16 and h == bench.csrc_hash()
    p = tmp_path / "stage_counters.json"
    base = {"workload": "terrain_1002k", "spp": 128, "streams": 4, "tag": "t", "kernels": {}}
    p.write_text(json.dumps(dict(base, csrc_hash=h)))
    pm, fresh = bench.load_stage_counters(str(p), "terrain_1002k", 128, 4)
    assert pm is not None and fresh
    p.write_text(json.dumps(dict(base, csrc_hash="0" * 16)))
    pm, fresh = bench.load_stage_counters(str(p), "terrain_1002k", 128, 4)
    assert pm is not None and not fresh
    assert bench.load_stage_counters(str(p), "atrium", 128, 4) == (None, False)
    assert bench.load_stage_counters(str(tmp_path / "missing.json"), "terrain_1002k", 128, 4) == (None, False)
    # the committed file belongs to the committed sources
    import os
    committed = os.path.join(os.path.dirname(bench.__file__), "profiles", "stage_counters.json")
    pm, fresh = bench.load_stage_counters(committed, "terrain_1002k", 256, 4)
    assert pm is not None and fresh, "profiles/stage_counters.json is stale: re-run tools/evidence.sh after changing csrc/"
