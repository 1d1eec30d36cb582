"""RenderContext over librfwhip.so — the HIP rendercore behind the reference's plugin interface
(RFW/system/context/rfw/context/context.h:74-111).  There is no CPU path: if the library or a HIP device is missing
the constructor raises."""
import ctypes as C
import os

import numpy as np

from . import abi
from ._binding import CoreBinding, RenderGroup

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librfwhip.so")
_lib = None


def load_library():
    """dlopen the in-tree librfwhip.so (RTLD_GLOBAL is not needed; HIP is linked in)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "librfwhip.so is missing (%s): build it with __graft_entry__.build(); "
                "the rendercore has no fallback path" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
    return _lib


class RenderContext(CoreBinding):
    def __init__(self, device=0, rank=0, world=1):
        super().__init__(load_library(), "rfwhip_", device, rank, world)
        vp, u32, i32, fp = C.c_void_p, C.c_uint32, C.c_int, C.c_float
        for name, res, args in [
            ("get_kernel_time", i32, [vp, i32, C.POINTER(fp), C.POINTER(u32), i32]),
            ("get_setting", i32, [vp, C.c_char_p, C.c_char_p, C.c_size_t]),
            ("get_settings", i32, [vp, C.POINTER(C.c_char_p), C.c_size_t]),
            ("version", C.c_char_p, []),
        ]:
            f = self._fn(name)
            f.restype, f.argtypes = res, args

    # ---- measurement hooks -------------------------------------------------------------------------------------------
    KERNELS = ("generate", "extend", "shade", "connect", "finalize", "refit")

    def get_kernel_time(self, which, reset=False):
        ms, n = C.c_float(), C.c_uint32()
        idx = self.KERNELS.index(which) if isinstance(which, str) else int(which)
        self._check(self._fn("get_kernel_time")(self._ctx, idx, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def get_setting(self, key):
        buf = C.create_string_buffer(128)
        self._check(self._fn("get_setting")(self._ctx, str(key).encode(), buf, 128))
        return buf.value.decode()

    def get_settings(self):
        keys = (C.c_char_p * 32)()
        n = self._fn("get_settings")(self._ctx, keys, 32)
        return {keys[i].decode(): self.get_setting(keys[i].decode()) for i in range(n)}

    def version(self):
        return self._fn("version")().decode()


def render_group(devices, transport="auto"):
    """n devices of one node behind one object (rfwhip_group_*): RenderGroup over the in-tree librfwhip.so."""
    return RenderGroup(load_library(), "rfwhip_", devices, transport)


def comm_unique_id():
    """The 128-byte id rank 0 hands to the other ranks before rfwhip_comm_create (include/rfwhip.h)."""
    lib = load_library()
    buf = C.create_string_buffer(128)
    lib.rfwhip_comm_unique_id.restype, lib.rfwhip_comm_unique_id.argtypes = C.c_int, [C.c_void_p, C.c_size_t]
    if lib.rfwhip_comm_unique_id(buf, 128) != 0:
        lib.rfwhip_last_error.restype = C.c_char_p
        raise RuntimeError((lib.rfwhip_last_error() or b"unknown error").decode(errors="replace"))
    return buf.raw


class RenderComm:
    """One process per device: this rank's end of the strip gather (rfwhip_comm_*, RCCL issued by librfwhip.so).  Collective:
    every rank of the context's world constructs it with the same id, and every rank calls gather()."""

    def __init__(self, ctx, id_bytes):
        self._lib = load_library()
        vp, i32 = C.c_void_p, C.c_int
        for name, res, args in [("rfwhip_comm_create", i32, [vp, vp, C.POINTER(vp)]), ("rfwhip_comm_gather", i32, [vp, vp]),
                                ("rfwhip_comm_wait", i32, [vp]), ("rfwhip_comm_destroy", None, [vp]), ("rfwhip_last_error", C.c_char_p, [])]:
            f = getattr(self._lib, name)
            f.restype, f.argtypes = res, args
        self._c = vp()
        self._id = C.create_string_buffer(bytes(id_bytes), 128) if id_bytes is not None else None
        self._check(self._lib.rfwhip_comm_create(ctx._ctx, self._id, C.byref(self._c)))

    def _check(self, code):
        if code != 0:
            raise RuntimeError((self._lib.rfwhip_last_error() or b"unknown error").decode(errors="replace"))

    def gather(self, full_rgba_device_ptr=0):
        """Enqueue present -> send / receive -> de-interleave (root: into the given device buffer, 0 = an internal one)."""
        self._check(self._lib.rfwhip_comm_gather(self._c, C.c_void_p(full_rgba_device_ptr or None)))

    def wait(self):
        self._check(self._lib.rfwhip_comm_wait(self._c))

    def destroy(self):
        if self._c:
            self._lib.rfwhip_comm_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
