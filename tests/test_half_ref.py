"""half -> float against the REFERENCE's own code: oracle/_ref/libhalfref.so is external/half2.1.0/half.hpp (the type
rfw::DeviceMaterial keeps its colours, absorption and uv scales in, structs.h:9-10,88-117) compiled where it lies under
/root/reference by oracle/Makefile's `ref` target.  All 65 536 bit patterns: the oracle's rfwo_half_to_float, the product's
rt::half_to_float in its host form (emulation build) and — GPU tier — the device form (v_cvt_f32_f16, through rfwhip_kat)."""
import numpy as np
import pytest


def _reference_table(orc):
    ref = orc.load_half_ref()
    if ref is None:
        pytest.skip("oracle/_ref/libhalfref.so is missing (built only where /root/reference exists)")
    bits = np.arange(65536, dtype=np.uint32)
    return bits, np.array([ref.rfw_ref_half_to_float(int(b)) for b in bits], dtype=np.float32)


def _same_floats(a, b):
    """bit-equal, any NaN matching any NaN (payloads: 10 mantissa bits shifted up, checked separately)"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    nan = np.isnan(a) & np.isnan(b)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | nan))


def _kat_half(ctx, bits):
    rec = np.zeros((len(bits) // 8, 24), np.float32)
    rec[:, :8] = bits.astype(np.uint32).reshape(-1, 8).view(np.float32)
    return ctx.kat("half_to_float", rec).reshape(-1)


def test_oracle_and_host_form_equal_the_reference_half_type(pkg, orc, make_emu):
    bits, want = _reference_table(orc)
    L = orc.load()
    got = np.array([L.rfwo_half_to_float(int(b)) for b in bits], dtype=np.float32)
    assert _same_floats(got, want)
    # known values of the format itself
    assert want[0x3C00] == 1.0 and want[0x3800] == 0.5 and want[0xC000] == -2.0 and want[0x0001] == np.float32(2.0 ** -24)
    assert want[0x7BFF] == 65504.0 and np.isinf(want[0x7C00]) and np.isnan(want[0x7E00])
    e = make_emu()
    assert _same_floats(_kat_half(e, bits), want)
    # and back: every finite half survives float -> half of the reference's packer
    ref = orc.load_half_ref()
    finite = np.isfinite(want)
    back = np.array([ref.rfw_ref_float_to_half(float(v)) for v in want[finite]], dtype=np.uint32)
    assert np.array_equal(back & 0x7FFF, bits[finite] & 0x7FFF)  # (+0 / -0 keep their sign too, but are equal as values)


@pytest.mark.gpu
def test_device_half_conversion_equals_the_reference_half_type(pkg, orc, make_hip):
    """v_cvt_f32_f16 with f16 denormals enabled (rt_core.h: half_to_float) over all 65 536 patterns."""
    bits, want = _reference_table(orc)
    h = make_hip()
    assert _same_floats(_kat_half(h, bits), want)
