"""ctypes / numpy mirrors of include/rfwhip_abi.h (byte layouts of the reference's plugin-boundary PODs).

Reference layouts: RFW/system/context/rfw/context/structs.h:24-255, device_structs.h:95-103, context.h:50-72,
camera.h:27-37, RFW/system/bvh/include/bvh/bvh_node.h:23-28.  Sizes are asserted at import time.
"""
import ctypes as C

import numpy as np

# ---- numpy dtypes (bulk data) -----------------------------------------------------------------------------------
TRIANGLE_DTYPE = np.dtype(
    [
        ("u", "<f4", 3), ("lightTriIdx", "<i4"),
        ("v", "<f4", 3), ("material", "<u4"),
        ("vN0", "<f4", 3), ("Nx", "<f4"),
        ("vN1", "<f4", 3), ("Ny", "<f4"),
        ("vN2", "<f4", 3), ("Nz", "<f4"),
        ("T", "<f4", 3), ("area", "<f4"),
        ("B", "<f4", 3), ("LOD", "<f4"),
        ("vertex0", "<f4", 3), ("dummy1", "<f4"),
        ("vertex1", "<f4", 3), ("dummy2", "<f4"),
        ("vertex2", "<f4", 3), ("dummy3", "<f4"),
    ]
)
assert TRIANGLE_DTYPE.itemsize == 160

MAP_DESC_DTYPE = np.dtype(
    [("width", "<i2"), ("height", "<i2"), ("uscale", "<f2"), ("vscale", "<f2"), ("uoffs", "<f2"), ("voffs", "<f2"),
     ("addr", "<u4")]
)
assert MAP_DESC_DTYPE.itemsize == 16

MATERIAL_DTYPE = np.dtype(
    [("diffuse", "<f2", 3), ("transmittance", "<f2", 3), ("flags", "<u4"), ("parameters", "<u4", 4),
     ("map", MAP_DESC_DTYPE, 10)]
)
assert MATERIAL_DTYPE.itemsize == 192

MATERIAL_TEX_IDS_DTYPE = np.dtype([("texture", "<i4", 11)])
assert MATERIAL_TEX_IDS_DTYPE.itemsize == 44

AREA_LIGHT_DTYPE = np.dtype(
    [("position", "<f4", 3), ("energy", "<f4"), ("normal", "<f4", 3), ("area", "<f4"), ("radiance", "<f4", 3),
     ("dummy0", "<i4"), ("vertex0", "<f4", 3), ("triIdx", "<i4"), ("vertex1", "<f4", 3), ("instIdx", "<i4"),
     ("vertex2", "<f4", 3), ("dummy1", "<i4")]
)
assert AREA_LIGHT_DTYPE.itemsize == 96
POINT_LIGHT_DTYPE = np.dtype([("position", "<f4", 3), ("energy", "<f4"), ("radiance", "<f4", 3), ("dummy", "<i4")])
assert POINT_LIGHT_DTYPE.itemsize == 32
SPOT_LIGHT_DTYPE = np.dtype(
    [("position", "<f4", 3), ("cosInner", "<f4"), ("radiance", "<f4", 3), ("cosOuter", "<f4"),
     ("direction", "<f4", 3), ("energy", "<f4")]
)
assert SPOT_LIGHT_DTYPE.itemsize == 48
DIRECTIONAL_LIGHT_DTYPE = np.dtype(
    [("direction", "<f4", 3), ("energy", "<f4"), ("radiance", "<f4", 3), ("dummy", "<i4")]
)
assert DIRECTIONAL_LIGHT_DTYPE.itemsize == 32

BVH_NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("left_first", "<i4"), ("count", "<i4")])
assert BVH_NODE_DTYPE.itemsize == 32

# MatPropFlags, structs.h:67-83
MAT_IS_DIELECTRIC = 0
MAT_DIFFUSE_MAP_IS_HDR = 1
MAT_HAS_DIFFUSE_MAP = 2
MAT_HAS_NORMAL_MAP = 3
MAT_HAS_SPECULARITY_MAP = 4
MAT_HAS_ROUGHNESS_MAP = 5
MAT_IS_ANISOTROPIC = 6
MAT_HAS_2ND_NORMAL_MAP = 7
MAT_HAS_3RD_NORMAL_MAP = 8
MAT_HAS_2ND_DIFFUSE_MAP = 9
MAT_HAS_3RD_DIFFUSE_MAP = 10
MAT_HAS_SMOOTH_NORMALS = 11
MAT_HAS_ALPHA = 12
MAT_HAS_ALPHA_MAP = 13

TEX_FLOAT4 = 0
TEX_UINT = 1

RESET = 0  # rfw::RenderStatus, context.h:19-23
CONVERGE = 1


# ---- ctypes structs (small, passed by pointer/value) --------------------------------------------------------------
class Mesh(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("normals", C.c_void_p), ("texCoords", C.c_void_p),
                ("triangles", C.c_void_p), ("indices", C.c_void_p), ("vertexCount", C.c_size_t),
                ("triangleCount", C.c_size_t)]


class Texture(C.Structure):
    _fields_ = [("type", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("texelCount", C.c_uint32),
                ("texAddr", C.c_uint32), ("_pad", C.c_uint32), ("data", C.c_void_p)]


class LightCount(C.Structure):
    _fields_ = [("areaLightCount", C.c_uint32), ("pointLightCount", C.c_uint32), ("spotLightCount", C.c_uint32),
                ("directionalLightCount", C.c_uint32)]


class CameraPOD(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("direction", C.c_float * 3), ("focalDistance", C.c_float),
                ("aperture", C.c_float), ("brightness", C.c_float), ("contrast", C.c_float), ("FOV", C.c_float),
                ("aspectRatio", C.c_float), ("clampValue", C.c_float), ("pixelCount", C.c_int32 * 2)]


class CameraView(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("p1", C.c_float * 3), ("p2", C.c_float * 3), ("p3", C.c_float * 3),
                ("aperture", C.c_float), ("spreadAngle", C.c_float)]


class RenderStats(C.Structure):
    _fields_ = [("primaryTime", C.c_float), ("primaryCount", C.c_uint32), ("secondaryTime", C.c_float),
                ("secondaryCount", C.c_uint32), ("deepTime", C.c_float), ("deepCount", C.c_uint32),
                ("shadowTime", C.c_float), ("shadowCount", C.c_uint32), ("shadeTime", C.c_float),
                ("finalizeTime", C.c_float), ("animationTime", C.c_float), ("renderTime", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Counters(C.Structure):
    _fields_ = [("rays_extend", C.c_uint64), ("rays_shadow", C.c_uint64), ("inner_extend", C.c_uint64),
                ("tris_extend", C.c_uint64), ("inner_shadow", C.c_uint64), ("tris_shadow", C.c_uint64),
                ("shaded", C.c_uint64), ("samples", C.c_uint64),
                # rendercore only (the oracle fills the first eight): node visits served by the LDS top-of-tree cache
                ("lds_extend", C.c_uint64), ("lds_shadow", C.c_uint64),
                ("extend_ticks", C.c_uint64), ("extend_launches_timed", C.c_uint64)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


assert C.sizeof(Mesh) == 56
assert C.sizeof(Texture) == 32
assert C.sizeof(CameraPOD) == 60
assert C.sizeof(CameraView) == 56
assert C.sizeof(RenderStats) == 48
