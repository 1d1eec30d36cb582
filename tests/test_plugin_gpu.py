"""The drop-in boundary end to end: a stand-in for rfw::system (tests/plugin/plugin_host.cpp) dlopens HipRT.so,
resolves createRenderContext / destroyRenderContext (export.h:8-15) and drives the plugin through the VIRTUAL
rfw::RenderContext interface in system::synchronize's order; the numbers it prints are checked analytically."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices,in_flight", [(None, 1), ("0,0,0", 1), ("0,0", 2), (None, 4)])
def test_plugin_through_the_virtual_interface(pkg, devices, in_flight):
    """devices "0,0,0": the plugin on an rfwhip_group of three contexts (strip split + gather below the C ABI; one GPU, so
    the peer transport) must print the very same numbers."""
    host = os.path.join(ROOT, "tests", "plugin", "plugin_host")
    plugin_dir = os.path.dirname(pkg.LIB_PATH)
    assert os.path.exists(host) and os.path.exists(os.path.join(plugin_dir, "HipRT.so")), "run __graft_entry__.build()"
    env = dict(os.environ)
    if devices:
        env.update(RFWHIP_DEVICES=devices, RFWHIP_TRANSPORT="peer")
    if in_flight > 1:  # frames in flight: the first render_frame hands out its own frame (there is no earlier one)
        env.update(RFWHIP_FRAMES_IN_FLIGHT=str(in_flight))
    r = subprocess.run([host, plugin_dir], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr
    out = dict(line.split(" ", 1) for line in r.stdout.strip().splitlines())
    inst, prim, dist = out["probe"].split()
    assert int(inst) == 0 and int(prim) in (0, 1) and abs(float(dist) - 4.0) < 0.05
    c = [float(x) for x in out["center"].split()]
    # albedo 0.5 * (ambient 0.1 + radiance 8 / d^2 * NdotL) with d ~= 4, NdotL ~= 1  (Context.cpp:228,270)
    assert all(abs(v - 0.5 * (0.1 + 8.0 / 16.0)) < 0.01 for v in c[:3]) and c[3] == 1.0
    corner = [float(x) for x in out["corner"].split()]
    assert corner == [0.25, 0.5, 0.75, 0.0]          # sky texel, alpha 0 (Context.cpp:194)
    assert out["primary"].split()[0] == str(64 * 48) and out["primary"].split()[2] == "1"
    assert out["threw"] == "1"
