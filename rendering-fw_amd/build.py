"""In-tree build of the native pieces with hipcc for gfx950 (no CMake, no JIT cache: the .so files must travel to the
GPU box with the repository snapshot)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

CORE_SOURCES = ["rfwhip_api.cpp", "rfwhip_group.cpp", "bvh_build.cpp", "kernels.hip", "lbvh.hip"]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-I" + INCLUDE, "-I" + CSRC,
          "-Wall", "-Wno-unused-function", "-Wno-unused-result"] + os.environ.get("RFWHIP_EXTRA_FLAGS", "").split()


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip", ".cpp", ".hpp", ".inc"))]
    d += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    pdir = os.path.join(CSRC, "plugin")
    if os.path.isdir(pdir):
        for root, _, files in os.walk(pdir):
            d += [os.path.join(root, f) for f in files]
    return d


def build(force=False, verbose=False):
    """Compile librfwhip.so (core + C ABI) and HipRT.so (the rfw::RenderContext plugin). Returns the paths."""
    deps = _deps()
    outs = []
    core = os.path.join(HERE, "librfwhip.so")
    if force or _stale(core, deps):
        objs = [os.path.join(CSRC, src + ".o") for src in CORE_SOURCES]
        _run_parallel([[HIPCC] + COMMON + ["-c", os.path.join(CSRC, src), "-o", obj] for src, obj in zip(CORE_SOURCES, objs)], verbose)
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", core, "-lpthread", "-ldl"], verbose)
    outs.append(core)
    plugin_src = os.path.join(CSRC, "plugin", "HipRT.cpp")
    if os.path.exists(plugin_src):
        plugin = os.path.join(HERE, "HipRT.so")  # no "lib" prefix: system.cpp:119-121 dlopens "<Name>.so"
        if force or _stale(plugin, deps + [core]):
            _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-I" + INCLUDE,
                  "-I" + os.path.join(CSRC, "plugin"), plugin_src, "-o", plugin, "-L" + HERE, "-lrfwhip",
                  "-Wl,-rpath,$ORIGIN"], verbose)
        outs.append(plugin)
        # stand-in for rfw::system's side of the boundary (tests/plugin/plugin_host.cpp), used by the gpu tests
        host_src = os.path.join(os.path.dirname(HERE), "tests", "plugin", "plugin_host.cpp")
        if os.path.exists(host_src):
            host = os.path.join(os.path.dirname(HERE), "tests", "plugin", "plugin_host")
            if force or _stale(host, deps + [host_src]):
                _run(["g++", "-O1", "-std=c++17", "-I" + INCLUDE, "-I" + os.path.join(CSRC, "plugin"), host_src, "-o", host,
                      "-ldl"], verbose)
            outs.append(host)
    return outs


def build_strict(force=False, verbose=False):
    """The VALIDATION build (csrc/rt_strict_math.h): the same sources with -DRT_STRICT_MATH -ffp-contract=off into
    tests/_strict/librfwhip_strict.so — test infrastructure (tests/test_strict_gpu.py compares it bit for bit with the host
    emulation built the same way); never loaded by the product.  Built here so that it travels to the GPU box with the tree."""
    out_dir = os.path.join(os.path.dirname(HERE), "tests", "_strict")
    out = os.path.join(out_dir, "librfwhip_strict.so")
    if not force and not _stale(out, _deps()):
        return out
    os.makedirs(out_dir, exist_ok=True)
    objs = [os.path.join(out_dir, src + ".o") for src in CORE_SOURCES]
    _run_parallel([[HIPCC] + COMMON + ["-DRT_STRICT_MATH", "-ffp-contract=off", "-c", os.path.join(CSRC, src), "-o", obj]
                   for src, obj in zip(CORE_SOURCES, objs)], verbose)
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out, "-lpthread", "-ldl"], verbose)
    return out


def _run_parallel(cmds, verbose):
    """The translation units of one library side by side (hipcc spends half a minute on each of the two .hip files)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(cmds), max(1, (os.cpu_count() or 2) // 2))) as pool:
        for f in [pool.submit(_run, c, verbose) for c in cmds]:
            f.result()


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("native build failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
    if verbose and r.stdout.strip():
        print(r.stdout, file=sys.stderr)


if __name__ == "__main__":
    print("\n".join(build(force="--force" in sys.argv, verbose=True)))
