"""rendering-fw_amd — MI355X-native (gfx950, HIP) wavefront rendercore for MeirBon/rendering-fw.

The product is csrc/ (hand-written HIP kernels + the C ABI of include/rfwhip.h, built into librfwhip.so, and the
rfw::RenderContext plugin csrc/plugin/HipRT.cpp).  The Python modules are the headless host side used by the tests
and bench.py: a mirror of the RenderContext interface over ctypes plus the caller-side scene assembly.

The directory name contains a hyphen (it is the reference repository's name + "_amd"), so it is imported through
`__graft_entry__.load_package()` under the module name `rendering_fw_amd`.
"""
from . import _binding, abi, scenes
from . import gltf
from . import obj
from .abi import CONVERGE, RESET
from .camera import Camera
from .context import LIB_PATH, RenderComm, RenderContext, comm_unique_id, load_library, render_group
from .build import build as build_native
from .build import build_strict

__all__ = ["abi", "scenes", "gltf", "obj", "Camera", "RenderContext", "load_library", "render_group", "RenderComm", "comm_unique_id", "LIB_PATH", "build_native", "build_strict", "RESET", "CONVERGE"]
