"""Host-side mirror of rfw::RenderContext (RFW/system/context/rfw/context/context.h:74-111) over a C ABI.

`CoreBinding` speaks to any shared library that exports the entry points of include/rfwhip.h under a given prefix.
The product instantiates it with librfwhip.so / "rfwhip_" (context.py); the test oracle re-uses it with its own
library and prefix.  Method names, argument meaning and error behaviour follow the reference interface: failures
raise RuntimeError (the reference throws std::runtime_error across the plugin boundary, context.h:84-91).
"""
import ctypes as C

import numpy as np

from . import abi


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class CoreBinding:
    def __init__(self, lib, prefix, device=0, rank=0, world=1, borrowed=None):
        """borrowed: an existing context pointer owned by somebody else (a RenderGroup): used, never destroyed."""
        self._lib = lib
        self._p = prefix
        self._ctx = C.c_void_p()
        self._borrowed = borrowed is not None
        self._declare()
        if borrowed is not None:
            self._ctx = C.c_void_p(borrowed)
        else:
            self._check(self._fn("create")(int(device), int(rank), int(world), C.byref(self._ctx)))
        self.rank, self.world = rank, world
        self.width = self.height = 0

    # ---- plumbing -------------------------------------------------------------------------------------------------
    def _fn(self, name):
        return getattr(self._lib, self._p + name)

    def _has(self, name):
        return hasattr(self._lib, self._p + name)

    def _declare(self):
        vp, sz, u32, i32, fp = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_float
        sig = {
            "last_error": (C.c_char_p, []),
            "create": (i32, [i32, i32, i32, C.POINTER(vp)]),
            "cleanup": (i32, [vp]),
            "destroy": (None, [vp]),
            "init": (i32, [vp, u32, u32]),
            "set_sky": (i32, [vp, vp, sz, sz]),
            "set_blue_noise": (i32, [vp, vp, sz]),
            "set_textures": (i32, [vp, vp, sz]),
            "set_materials": (i32, [vp, vp, vp, sz]),
            "set_mesh": (i32, [vp, sz, C.POINTER(abi.Mesh)]),
            "set_instance": (i32, [vp, sz, sz, vp, vp]),
            "set_lights": (i32, [vp, abi.LightCount, vp, vp, vp, vp]),
            "update": (i32, [vp]),
            "camera_get_view": (None, [C.POINTER(abi.CameraPOD), C.POINTER(abi.CameraView)]),
            "render": (i32, [vp, C.POINTER(abi.CameraPOD), i32]),
            "wait": (i32, [vp]),
            "read_framebuffer": (i32, [vp, vp]),
            "local_rows": (u32, [vp]),
            "set_probe_index": (i32, [vp, u32, u32]),
            "get_probe_results": (i32, [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(fp)]),
            "get_stats": (i32, [vp, C.POINTER(abi.RenderStats)]),
            "set_setting": (i32, [vp, C.c_char_p, C.c_char_p]),
            "read_primary_hits": (i32, [vp, vp, vp, vp, vp, vp]),
            "get_bvh": (i32, [vp, sz, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(sz)]),
            "trace_rays": (i32, [vp, sz, vp, vp, fp, fp, vp, vp, vp, vp, vp]),
        }
        for name, (res, args) in sig.items():
            f = self._fn(name)
            f.restype, f.argtypes = res, args
        # device-side presents: only the rendercore (and its emulation build) export these
        for name, (res, args) in {"set_mesh_skin": (i32, [vp, sz, vp, vp, vp, sz]),
                                  "pose_mesh": (i32, [vp, sz, vp, sz]),
                                  "set_mesh_morph": (i32, [vp, sz, vp, vp, vp, sz, sz]),
                                  "morph_mesh": (i32, [vp, sz, vp, sz]),
                                  "read_framebuffer_device": (i32, [vp, vp]),
                                  "read_local_framebuffer_stream": (i32, [vp, vp, vp]),
                                  "deinterleave_stream": (i32, [vp, vp, vp, vp]),
                                  "read_local_framebuffer_device": (i32, [vp, vp]),
                                  "deinterleave_device": (i32, [vp, vp, vp]),
                                  "kat": (i32, [vp, i32, sz, vp, vp]),
                                  "get_counters": (i32, [vp, C.POINTER(abi.Counters), i32])}.items():
            if self._has(name):
                f = self._fn(name)
                f.restype, f.argtypes = res, args

    def _check(self, code):
        if code != 0:
            msg = self._fn("last_error")()
            raise RuntimeError((msg or b"unknown error").decode(errors="replace"))

    # ---- rfw::RenderContext ------------------------------------------------------------------------------------------
    def get_supported_targets(self):
        return ["BUFFER"]  # RenderTarget::BUFFER, context.h:27-34

    def init(self, width, height):
        self._check(self._fn("init")(self._ctx, int(width), int(height)))
        self.width, self.height = int(width), int(height)

    def cleanup(self):
        if self._ctx:
            self._check(self._fn("cleanup")(self._ctx))

    def destroy(self):
        if self._ctx and not self._borrowed:
            self._fn("destroy")(self._ctx)
        self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def render_frame(self, camera, status=abi.RESET):
        """render_frame(const Camera&, RenderStatus): synchronous like the reference (glFinish at the end)."""
        self.render_async(camera, status)
        self.wait()

    def render_async(self, camera, status=abi.RESET):
        pod = camera.pod() if hasattr(camera, "pod") else camera
        self._check(self._fn("render")(self._ctx, C.byref(pod), int(status)))

    def wait(self):
        self._check(self._fn("wait")(self._ctx))

    def set_materials(self, materials, tex_ids=None):
        m = np.ascontiguousarray(materials, dtype=abi.MATERIAL_DTYPE)
        if tex_ids is None:
            tex_ids = np.full(len(m), -1, dtype=np.int32).repeat(11).reshape(len(m), 11).view(abi.MATERIAL_TEX_IDS_DTYPE)
        t = np.ascontiguousarray(tex_ids)
        self._check(self._fn("set_materials")(self._ctx, m.ctypes.data, t.ctypes.data, len(m)))

    def set_textures(self, textures):
        """textures: list of dicts {type, width, height, data(np.ndarray)}; data may include appended mip levels."""
        arr = (abi.Texture * max(1, len(textures)))()
        keep = []
        for i, t in enumerate(textures):
            data = np.ascontiguousarray(t["data"])
            keep.append(data)
            per = 4 if t["type"] == abi.TEX_FLOAT4 else 1
            arr[i] = abi.Texture(t["type"], t["width"], t["height"], data.size // per, 0, 0, data.ctypes.data)
        self._check(self._fn("set_textures")(self._ctx, C.cast(arr, C.c_void_p), len(textures)))

    def set_mesh(self, index, vertices, triangles, indices=None):
        v = _f32(vertices).reshape(-1, 4)
        tr = np.ascontiguousarray(triangles, dtype=abi.TRIANGLE_DTYPE)
        idx = None if indices is None else np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        m = abi.Mesh(v.ctypes.data, None, None, tr.ctypes.data, None if idx is None else idx.ctypes.data, len(v), len(tr))
        self._check(self._fn("set_mesh")(self._ctx, int(index), C.byref(m)))

    def set_instance(self, i, mesh_idx, transform, normal_matrix=None):
        """transform: 4x4 in maths (row, col) convention; sent column-major like glm::mat4.  normal_matrix defaults to
        the inverse-transpose of the upper 3x3 (system.cpp:347)."""
        t = np.asarray(transform, dtype=np.float64).reshape(4, 4)
        if normal_matrix is None:
            normal_matrix = np.linalg.inv(t[:3, :3]).T
        n = np.asarray(normal_matrix, dtype=np.float64).reshape(3, 3)
        tc, nc = _f32(t.T).ravel(), _f32(n.T).ravel()
        self._check(self._fn("set_instance")(self._ctx, int(i), int(mesh_idx), tc.ctypes.data, nc.ctypes.data))

    def set_sky(self, pixels, width, height):
        p = _f32(pixels).reshape(-1, 3)
        assert len(p) == width * height
        self._check(self._fn("set_sky")(self._ctx, p.ctypes.data, int(width), int(height)))

    def set_mesh_skin(self, index, joints, weights, base_normals):
        """Device skinning: per-vertex joints (V x 4 uint32), weights (V x 4) and bind-pose normals (V x 3|4) of mesh
        `index`, whose last set_mesh vertices are the bind pose."""
        j = np.ascontiguousarray(joints, dtype=np.uint32).reshape(-1, 4)
        w = _f32(weights).reshape(-1, 4)
        n = _f32(base_normals).reshape(len(j), -1)
        n4 = np.zeros((len(j), 4), np.float32)
        n4[:, :3] = n[:, :3]
        self._keep = (j, w, n4)
        self._check(self._fn("set_mesh_skin")(self._ctx, int(index), j.ctypes.data, w.ctypes.data, n4.ctypes.data, len(j)))

    def pose_mesh(self, index, joint_matrices):
        """joint_matrices: (J, 4, 4) row-major numpy matrices acting on column vectors (object -> posed)."""
        m = _f32(joint_matrices).reshape(-1, 4, 4)
        cm = np.ascontiguousarray(np.transpose(m, (0, 2, 1)))  # column-major storage
        self._check(self._fn("pose_mesh")(self._ctx, int(index), cm.ctypes.data, len(cm)))

    def set_mesh_morph(self, index, base_normals, target_positions, target_normals):
        """Device morph targets of mesh `index` (whose last set_mesh vertices are the base pose): base normals (V x 3|4) and
        per target the position / normal displacements, (T, V, 3|4) each."""
        def f4(a, lead):
            a = _f32(a).reshape(lead + (-1,))
            out = np.zeros(lead + (4,), np.float32)
            out[..., :3] = a[..., :3]
            return out
        tp = _f32(target_positions)
        t, v = tp.shape[0], tp.shape[1]
        bn, tp4, tn4 = f4(base_normals, (v,)), f4(tp, (t, v)), f4(target_normals, (t, v))
        self._check(self._fn("set_mesh_morph")(self._ctx, int(index), bn.ctypes.data, tp4.ctypes.data, tn4.ctypes.data, t, v))

    def morph_mesh(self, index, weights):
        w = _f32(weights).reshape(-1)
        self._check(self._fn("morph_mesh")(self._ctx, int(index), w.ctypes.data, len(w)))

    def set_blue_noise(self, table):
        """The reference's 5 x 65536-word blue-noise table (createBlueNoiseBuffer()); see scenes.synthetic_blue_noise."""
        t = np.ascontiguousarray(table, dtype=np.uint32).reshape(-1)
        self._check(self._fn("set_blue_noise")(self._ctx, t.ctypes.data, int(t.size)))

    def set_lights(self, area=None, point=None, spot=None, directional=None):
        def prep(a, dt):
            a = np.zeros(0, dtype=dt) if a is None else np.ascontiguousarray(a, dtype=dt)
            return a, (a.ctypes.data if len(a) else None)

        a, pa = prep(area, abi.AREA_LIGHT_DTYPE)
        p, pp = prep(point, abi.POINT_LIGHT_DTYPE)
        s, ps = prep(spot, abi.SPOT_LIGHT_DTYPE)
        d, pd = prep(directional, abi.DIRECTIONAL_LIGHT_DTYPE)
        self._check(self._fn("set_lights")(self._ctx, abi.LightCount(len(a), len(p), len(s), len(d)), pa, pp, ps, pd))

    def get_probe_results(self):
        inst, prim, dist = C.c_uint32(), C.c_uint32(), C.c_float()
        self._check(self._fn("get_probe_results")(self._ctx, C.byref(inst), C.byref(prim), C.byref(dist)))
        return inst.value, prim.value, dist.value

    def set_probe_index(self, x, y):
        self._check(self._fn("set_probe_index")(self._ctx, int(x), int(y)))

    def set_setting(self, key, value):
        self._check(self._fn("set_setting")(self._ctx, str(key).encode(), str(value).encode()))

    def update(self):
        self._check(self._fn("update")(self._ctx))

    def get_stats(self):
        s = abi.RenderStats()
        self._check(self._fn("get_stats")(self._ctx, C.byref(s)))
        return s

    # ---- headless BUFFER target + test hooks ----------------------------------------------------------------------
    def camera_view(self, camera):
        v = abi.CameraView()
        pod = camera.pod() if hasattr(camera, "pod") else camera
        self._fn("camera_get_view")(C.byref(pod), C.byref(v))
        return v

    def framebuffer(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self._fn("read_framebuffer")(self._ctx, out.ctypes.data))
        return out

    def read_framebuffer_device(self, device_ptr):
        self._check(self._fn("read_framebuffer_device")(self._ctx, C.c_void_p(device_ptr)))

    def read_local_framebuffer_device(self, device_ptr):
        """This rank's strips (local_rows() x width float4) into caller-owned device memory (a torch tensor's
        data_ptr())."""
        self._check(self._fn("read_local_framebuffer_device")(self._ctx, C.c_void_p(device_ptr)))

    def deinterleave_device(self, gathered_ptr, out_ptr):
        """Root side of the multi-GPU gather: [world][local_rows][width] float4 -> [height][width] float4."""
        self._check(self._fn("deinterleave_device")(self._ctx, C.c_void_p(gathered_ptr), C.c_void_p(out_ptr)))

    def read_local_framebuffer_stream(self, device_ptr, stream):
        """Stream-ordered present of this rank's strips on the caller's hipStream_t (an int, e.g.
        torch.cuda.current_stream().cuda_stream); no host synchronisation."""
        self._check(self._fn("read_local_framebuffer_stream")(self._ctx, C.c_void_p(device_ptr), C.c_void_p(stream)))

    def deinterleave_stream(self, gathered_ptr, out_ptr, stream):
        self._check(self._fn("deinterleave_stream")(self._ctx, C.c_void_p(gathered_ptr), C.c_void_p(out_ptr),
                                                    C.c_void_p(stream)))

    def local_rows(self):
        return int(self._fn("local_rows")(self._ctx))

    def primary_hits(self):
        n = self.width * self.height
        t, u, v = (np.empty(n, np.float32) for _ in range(3))
        prim, inst = np.empty(n, np.int32), np.empty(n, np.int32)
        self._check(self._fn("read_primary_hits")(self._ctx, t.ctypes.data, prim.ctypes.data, inst.ctypes.data,
                                                   u.ctypes.data, v.ctypes.data))
        shp = (self.height, self.width)
        return {"t": t.reshape(shp), "prim": prim.reshape(shp), "inst": inst.reshape(shp), "u": u.reshape(shp),
                "v": v.reshape(shp)}

    def trace_rays(self, org, dir, t_min=1e-5, t_max=1e34):
        """Closest hits of arbitrary world-space rays (n x 3 each) against the resident scene."""
        o, d = _f32(org).reshape(-1, 3), _f32(dir).reshape(-1, 3)
        n = len(o)
        t, u, v = (np.empty(n, np.float32) for _ in range(3))
        prim, inst = np.empty(n, np.int32), np.empty(n, np.int32)
        self._check(self._fn("trace_rays")(self._ctx, n, o.ctypes.data, d.ctypes.data, t_min, t_max, t.ctypes.data,
                                            prim.ctypes.data, inst.ctypes.data, u.ctypes.data, v.ctypes.data))
        return {"t": t, "prim": prim, "inst": inst, "u": u, "v": v}

    # known-answer hook: RFWHIP_KAT_* (include/rfwhip_abi.h)
    KAT = {"bsdf_eval": 0, "bsdf_pdf": 1, "bsdf_sample": 2, "tangent_space": 3, "pack_normal": 4,
           "random_barycentrics": 5, "point_on_light": 6, "light_pick_prob": 7, "blue_noise": 8, "hash": 9, "half_to_float": 10, "fastdiv": 11, "tex_wrap": 12}

    def kat(self, function, records):
        """One of the path tracer's functions on n records (n x 24 float32, integers as bit patterns) -> n x 8 float32."""
        rec = np.ascontiguousarray(records, dtype=np.float32).reshape(-1, 24)
        out = np.zeros((len(rec), 8), np.float32)
        self._check(self._fn("kat")(self._ctx, int(self.KAT[function]), len(rec), rec.ctypes.data, out.ctypes.data))
        return out

    def get_counters(self, reset=False):
        """Traversal statistics since the last reset (count_traversal=1): rays, popped inner nodes, triangle tests."""
        c = abi.Counters()
        self._check(self._fn("get_counters")(self._ctx, C.byref(c), int(reset)))
        return c.as_dict()

    def get_bvh(self, mesh_index):
        nn, np_ = C.c_size_t(), C.c_size_t()
        self._check(self._fn("get_bvh")(self._ctx, int(mesh_index), None, 0, None, 0, C.byref(nn), C.byref(np_)))
        nodes = np.zeros(nn.value, dtype=abi.BVH_NODE_DTYPE)
        prims = np.zeros(np_.value, dtype=np.uint32)
        self._check(self._fn("get_bvh")(self._ctx, int(mesh_index), nodes.ctypes.data, len(nodes), prims.ctypes.data,
                                        len(prims), C.byref(nn), C.byref(np_)))
        return nodes, prims


class RenderGroup:
    """n devices driven by ONE host thread through rfwhip_group_* (include/rfwhip.h): context i renders the strips of rank i
    of world n, one gather per presented frame lands the image on the root's device.  Looks like one RenderContext to the
    scene code: every set_* is repeated per context (each device holds the whole scene)."""
    TRANSPORTS = {"auto": 0, "rccl": 1, "peer": 2}

    def __init__(self, lib, prefix, devices, transport="auto"):
        self._lib, self._p = lib, prefix
        vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
        for name, res, args in [("group_create", i32, [C.POINTER(i32), i32, i32, C.POINTER(vp)]), ("group_destroy", None, [vp]),
                                ("group_size", i32, [vp]), ("group_transport", i32, [vp]), ("group_context", vp, [vp, i32]),
                                ("group_init", i32, [vp, u32, u32]), ("group_update", i32, [vp]),
                                ("group_set_setting", i32, [vp, C.c_char_p, C.c_char_p]),
                                ("group_render", i32, [vp, C.POINTER(abi.CameraPOD), i32]), ("group_gather", i32, [vp]),
                                ("group_wait", i32, [vp]), ("group_read_framebuffer", i32, [vp, vp]),
                                ("group_framebuffer_device", i32, [vp, C.POINTER(vp), C.POINTER(i32)]),
                                ("group_present_async", i32, [vp, i32]), ("group_present_wait", i32, [vp, i32, C.POINTER(vp)]),
                                ("last_error", C.c_char_p, [])]:
            f = self._fn(name)
            f.restype, f.argtypes = res, args
        devs = (i32 * len(devices))(*[int(d) for d in devices])
        self._g = vp()
        self._check(self._fn("group_create")(devs, len(devices), self.TRANSPORTS[transport], C.byref(self._g)))
        n = self._fn("group_size")(self._g)
        self.contexts = [CoreBinding(lib, prefix, rank=i, world=n, borrowed=self._fn("group_context")(self._g, i)) for i in range(n)]
        self.world, self.width, self.height = n, 0, 0
        self.transport = {v: k for k, v in self.TRANSPORTS.items()}[self._fn("group_transport")(self._g)]

    def _fn(self, name):
        return getattr(self._lib, self._p + name)

    def _check(self, code):
        if code != 0:
            raise RuntimeError((self._fn("last_error")() or b"unknown error").decode(errors="replace"))

    def __getattr__(self, name):
        # scene setters and friends: the same call on every context (set_sky, set_mesh, set_lights, set_blue_noise, ...)
        if name.startswith("set_") or name in ("pose_mesh", "morph_mesh"):
            def every(*a, **k):
                for c in self.contexts:
                    getattr(c, name)(*a, **k)
            return every
        raise AttributeError(name)

    def init(self, width, height):
        self._check(self._fn("group_init")(self._g, int(width), int(height)))
        self.width, self.height = int(width), int(height)
        for c in self.contexts:
            c.width, c.height = self.width, self.height

    def update(self):
        self._check(self._fn("group_update")(self._g))

    def set_setting(self, key, value):
        self._check(self._fn("group_set_setting")(self._g, str(key).encode(), str(value).encode()))

    def render_async(self, camera, status=abi.RESET):
        pod = camera.pod() if hasattr(camera, "pod") else camera
        self._check(self._fn("group_render")(self._g, C.byref(pod), int(status)))

    def gather(self):
        self._check(self._fn("group_gather")(self._g))

    def wait(self):
        self._check(self._fn("group_wait")(self._g))

    def render_frame(self, camera, status=abi.RESET):
        self.render_async(camera, status)
        self.wait()

    def framebuffer(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self._fn("group_read_framebuffer")(self._g, out.ctypes.data))
        return out

    def present_async(self, slot):
        """Enqueue gather + device-to-host copy of the image into pinned host slot 0 / 1 (frames in flight)."""
        self._check(self._fn("group_present_async")(self._g, int(slot)))

    def present_wait(self, slot):
        """Block until slot's copy has landed; returns the image as a numpy view of the pinned buffer."""
        ptr = C.c_void_p()
        self._check(self._fn("group_present_wait")(self._g, int(slot), C.byref(ptr)))
        buf = (C.c_float * (self.width * self.height * 4)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=np.float32).reshape(self.height, self.width, 4)

    def framebuffer_device(self):
        """(device pointer, device ordinal) of the root-side image of the last completed gather."""
        ptr, dev = C.c_void_p(), C.c_int()
        self._check(self._fn("group_framebuffer_device")(self._g, C.byref(ptr), C.byref(dev)))
        return ptr.value, dev.value

    def get_stats(self):
        return [c.get_stats() for c in self.contexts]

    def destroy(self):
        if self._g:
            for c in self.contexts:
                c.destroy()  # (borrowed: forgets the pointer)
            self._fn("group_destroy")(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
