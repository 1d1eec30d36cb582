"""rt::tex_wrap (rt_core.h): the wrap of a texel coordinate — `x % width` with x >= 0 (getShadingData.h:33-41) — comes from
v_rcp_f32 and two corrections on the device instead of the compiler's ~30-instruction signed remainder (a trilinear fetch needs
four).  Checked against `%` through rfwhip_kat on the host form (emulation build: plain `%`) and the device form: random pairs,
every texture size and mip level the scenes of this repository use, multiples of the width and their neighbours, the
threshold (2^22) on either side of which the device takes the fast or the plain path, and the largest operands."""
import numpy as np
import pytest


def _pairs():
    rng = np.random.default_rng(20260929)
    x = rng.integers(0, 2 ** 31, size=200_000, dtype=np.int64)
    w = rng.integers(1, 2 ** 31, size=200_000, dtype=np.int64)
    w[:120_000] = rng.integers(1, 8193, size=120_000)                 # texture widths
    x[:60_000] = rng.integers(0, 2 ** 22, size=60_000)                # (u + 1000) * width
    x[60_000:120_000] = rng.integers(0, 2 ** 24, size=60_000)
    edge_w = np.array([1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 32, 64, 96, 100, 128, 255, 256, 257, 512, 1000, 1024, 2048, 4096, 8192,
                       2 ** 16 - 1, 2 ** 16, 2 ** 22 - 1, 2 ** 22, 2 ** 22 + 1, 2 ** 24, 2 ** 31 - 1], dtype=np.int64)
    edge_x = np.array([0, 1, 2, 255, 256, 257, 255999, 256000, 256255, 256256, 2 ** 22 - 2, 2 ** 22 - 1, 2 ** 22, 2 ** 22 + 1,
                       2 ** 24 - 1, 2 ** 24, 2 ** 24 + 1, 2 ** 31 - 2, 2 ** 31 - 1], dtype=np.int64)
    ex, ew = np.meshgrid(edge_x, edge_w)
    mult = (edge_w[:, None] * np.array([1, 2, 3, 999, 1000, 1001, 1002, 4095, 16384])[None, :]).reshape(-1)
    mw = np.repeat(edge_w, 9)
    keep = mult < 2 ** 31 - 1
    mult, mw = mult[keep], mw[keep]
    mx = np.concatenate([mult - 1, mult, mult + 1]).clip(0, 2 ** 31 - 1)
    x = np.concatenate([x, ex.reshape(-1), mx])
    w = np.concatenate([w, ew.reshape(-1), mw, mw, mw])
    pad = (-len(x)) % 4
    return np.concatenate([x, np.zeros(pad, np.int64)]), np.concatenate([w, np.ones(pad, np.int64)])


def _check(ctx):
    x, w = _pairs()
    rec = np.zeros((len(x) // 4, 24), np.uint32)
    rec[:, 0:8:2] = x.reshape(-1, 4)
    rec[:, 1:8:2] = w.reshape(-1, 4)
    got = ctx.kat("tex_wrap", rec.view(np.float32))[:, :4].view(np.uint32).reshape(-1).astype(np.int64)
    want = x % w
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, (x[bad[:5]], w[bad[:5]], got[bad[:5]], want[bad[:5]])


def test_tex_wrap_host_form(make_emu):
    _check(make_emu())


@pytest.mark.gpu
def test_tex_wrap_device_form(make_hip):
    _check(make_hip())
