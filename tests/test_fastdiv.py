"""rt::fast_div (rt_types.h): the slot -> pixel mapping of every path divides 31-bit indices by per-frame constants through a
multiply-high and a shift instead of the ~30-instruction udiv sequence.  Checked against integer division through rfwhip_kat on
the host form (emulation build) and the device form: random pairs, every divisor the frames of this repository produce, powers of
two and their neighbours, and the largest operands."""
import numpy as np
import pytest


def _pairs():
    rng = np.random.default_rng(20260928)
    n = rng.integers(0, 2 ** 31, size=200_000, dtype=np.int64)
    d = rng.integers(1, 2 ** 31, size=200_000, dtype=np.int64)
    d[:50_000] = rng.integers(1, 4096, size=50_000)                  # tiles per row, small groups
    d[50_000:100_000] = rng.integers(1, 2 ** 28, size=50_000)        # slots of a sample group
    edge_d = np.array([1, 2, 3, 5, 7, 30, 60, 120, 240, 255, 256, 257, 2 ** 16 - 1, 2 ** 16, 2 ** 16 + 1, 2073600, 2088960,
                       66846720, 133693440, 2 ** 30 - 1, 2 ** 30, 2 ** 30 + 1, 2 ** 31 - 2, 2 ** 31 - 1], dtype=np.int64)
    edge_n = np.array([0, 1, 2, 239, 240, 241, 2 ** 16, 2 ** 24 - 1, 2 ** 24, 133693439, 133693440, 133693441, 2 ** 31 - 2, 2 ** 31 - 1],
                      dtype=np.int64)
    en, ed = np.meshgrid(edge_n, edge_d)
    mult = (edge_d[:, None] * np.arange(1, 9)[None, :]).reshape(-1)     # multiples of a divisor and their neighbours
    mult = mult[mult < 2 ** 31]
    mn = np.concatenate([mult - 1, mult, mult + 1]).clip(0, 2 ** 31 - 1)
    md = np.concatenate([np.repeat(edge_d, 8)[:len(mult)]] * 3)
    n = np.concatenate([n, en.reshape(-1), mn])
    d = np.concatenate([d, ed.reshape(-1), md])
    pad = (-len(n)) % 4
    return np.concatenate([n, np.zeros(pad, np.int64)]), np.concatenate([d, np.ones(pad, np.int64)])


def _check(ctx):
    n, d = _pairs()
    rec = np.zeros((len(n) // 4, 24), np.uint32)
    rec[:, 0:8:2] = n.reshape(-1, 4)
    rec[:, 1:8:2] = d.reshape(-1, 4)
    got = ctx.kat("fastdiv", rec.view(np.float32))[:, :4].view(np.uint32).reshape(-1).astype(np.int64)
    want = n // d
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, (n[bad[:5]], d[bad[:5]], got[bad[:5]], want[bad[:5]])


def test_fast_div_host_form(make_emu):
    _check(make_emu())


@pytest.mark.gpu
def test_fast_div_device_form(make_hip):
    _check(make_hip())
