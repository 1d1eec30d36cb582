"""Host-emulation build of the rendercore sources — TEST INFRASTRUCTURE ONLY.

Compiles rendering-fw_amd/csrc/{rfwhip_api.cpp,bvh_build.cpp,kernels.hip,lbvh.hip} with g++ and -DRFWHIP_HOST_EMULATION into
tests/_emu/librfwhip_emu.so: device memory becomes heap memory and every kernel launch becomes a plain loop over the
same per-ray / per-path functions (rt_core.h, the *_item functions of kernels.hip).  This lets the CPU test tier
(-m "not gpu") check the host logic (BVH build, TLAS, packing, strip interleave, xor128 jump-ahead, counters) and
the device arithmetic against the oracle without a GPU.  The product never loads this library:
rendering_fw_amd.context.load_library() only opens rendering-fw_amd/librfwhip.so and raises if it is missing."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "rendering-fw_amd", "csrc")
OUT_DIR = os.path.join(ROOT, "tests", "_emu")
OUT = os.path.join(OUT_DIR, "librfwhip_emu.so")


def build(force=False, defines=(), tag=""):
    """defines / tag: a variant of the library built with other compile-time constants (e.g. ("-DRT_LDS_NODES=64",), "_lds64")."""
    out = OUT if not tag else os.path.join(OUT_DIR, "librfwhip_emu%s.so" % tag)
    return _build(force, list(defines), out)


def _build(force, defines, OUT):
    srcs = [os.path.join(CSRC, f) for f in ("rfwhip_api.cpp", "rfwhip_group.cpp", "bvh_build.cpp", "kernels.hip", "lbvh.hip")]
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip", ".cpp", ".inc"))]
    deps += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DRFWHIP_HOST_EMULATION", "-fvisibility=hidden",
           "-mavx2", "-mfma", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + defines
    for s in srcs:
        cmd += ["-x", "c++", s]
    cmd += ["-o", OUT, "-lpthread"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("emulation build failed:\n" + r.stdout)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
