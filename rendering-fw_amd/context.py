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
