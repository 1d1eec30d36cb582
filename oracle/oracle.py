"""CPU ORACLE binding (test infrastructure, NOT product code) — see rfw_oracle.h.  PARITY UNPINNED.

Drives oracle/_build/librfworacle.so through the same host-side RenderContext mirror the product uses, so parity
tests feed both with one scene description.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "librfworacle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in ("rfw_oracle.c", "rfw_oracle.h", "rfw_oracle_math.h", "Makefile")]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "rfwhip_abi.h"))
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        r = subprocess.run(["make", "-C", HERE, "-B"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stdout)
    return LIB_PATH


def build_ref():
    """oracle/_ref/libbluenoise.so + libhalfref.so from the reference's own blue_noise.h / half.hpp (`make ref`); a no-op
    without /root/reference."""
    r = subprocess.run(["make", "-C", HERE, "ref"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle/_ref build failed:\n" + r.stdout)
    return os.path.join(HERE, "_ref", "libbluenoise.so")


def load_half_ref():
    """The reference's half_float::half conversions (oracle/_ref/libhalfref.so, built from external/half2.1.0/half.hpp)."""
    path = os.path.join(HERE, "_ref", "libhalfref.so")
    if not os.path.exists(path):
        build_ref()
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.rfw_ref_half_to_float.restype, L.rfw_ref_half_to_float.argtypes = C.c_float, [C.c_uint16]
    L.rfw_ref_float_to_half.restype, L.rfw_ref_float_to_half.argtypes = C.c_uint16, [C.c_float]
    return L


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        L = _lib
        u32p = C.POINTER(C.c_uint32)
        fp = C.POINTER(C.c_float)
        L.rfwo_xor128_next.restype, L.rfwo_xor128_next.argtypes = C.c_uint32, [u32p]
        L.rfwo_rng_rand.restype, L.rfwo_rng_rand.argtypes = C.c_float, [u32p]
        L.rfwo_xor128_jump.restype, L.rfwo_xor128_jump.argtypes = None, [u32p, C.c_uint64]
        L.rfwo_wang_hash.restype, L.rfwo_wang_hash.argtypes = C.c_uint32, [C.c_uint32]
        L.rfwo_random_int.restype, L.rfwo_random_int.argtypes = C.c_uint32, [u32p]
        L.rfwo_random_float.restype, L.rfwo_random_float.argtypes = C.c_float, [u32p]
        L.rfwo_half_to_float.restype, L.rfwo_half_to_float.argtypes = C.c_float, [C.c_uint16]
        L.rfwo_intersect_triangle.restype = C.c_int
        L.rfwo_intersect_triangle.argtypes = [fp, fp, C.c_float, fp, fp, fp, fp, fp, fp]
        L.rfwo_intersect_aabb.restype = C.c_int
        L.rfwo_intersect_aabb.argtypes = [fp, fp, fp, fp, C.c_float, fp, fp]
        L.rfwo_triangle_area.restype, L.rfwo_triangle_area.argtypes = C.c_float, [fp, fp, fp]
        L.rfwo_pack_normal.restype, L.rfwo_pack_normal.argtypes = C.c_uint32, [fp]
        L.rfwo_unpack_normal.restype, L.rfwo_unpack_normal.argtypes = None, [C.c_uint32, fp]
        L.rfwo_evaluate_bsdf.restype, L.rfwo_evaluate_bsdf.argtypes = None, [fp, u32p, fp, fp, fp, fp, fp]
        L.rfwo_sample_bsdf.restype = None
        L.rfwo_sample_bsdf.argtypes = [fp, fp, u32p, fp, fp, C.c_float, C.c_int, C.c_float, C.c_float, fp, fp, fp]
        L.rfwo_read_local_framebuffer.restype, L.rfwo_read_local_framebuffer.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
        L.rfwo_kat.restype, L.rfwo_kat.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
    return _lib


def f3(x):
    return (C.c_float * 3)(*[float(v) for v in x])


def OracleContext(pkg, rank=0, world=1):
    """Factory: an oracle-backed RenderContext.  `pkg` is the loaded rendering_fw_amd package (for the shared
    host-side binding class)."""
    base = pkg._binding.CoreBinding

    class _Oracle(base):
        def __init__(self):
            super().__init__(load(), "rfwo_", 0, rank, world)

        # get_counters(): the base class reads 8 x uint64 — the oracle fills the same order (rfw_oracle.h)

        def local_framebuffer(self):
            out = np.empty((self.local_rows(), self.width, 4), np.float32)
            self._check(self._lib.rfwo_read_local_framebuffer(self._ctx, out.ctypes.data))
            return out

    return _Oracle()
