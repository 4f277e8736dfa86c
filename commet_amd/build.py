"""Build recipes (in-tree, no JIT cache): the HIP library and the C++ host tools."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "commet_amd")
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libcommet_hip.so")
BIN_DIR = os.path.join(PKG, "bin")
ARCH = "gfx950"


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, **kw):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, **kw)


def hipcc():
    for c in ("hipcc", "/opt/rocm/bin/hipcc"):
        p = shutil.which(c)
        if p:
            return p
    raise RuntimeError("hipcc not found: the HIP path cannot be built")


def _sources(*dirs, exts=(".hip", ".hpp", ".h", ".cpp")):
    out = []
    for d in dirs:
        for f in sorted(os.listdir(d)):
            if f.endswith(exts):
                out.append(os.path.join(d, f))
    return out


def build_lib(force=False):
    """libcommet_hip.so: kernels + C ABI, cross-compiled for gfx950."""
    srcs = _sources(CSRC, os.path.join(CSRC, "capi"), os.path.join(CSRC, "host"), os.path.join(ROOT, "include"))   # capi.hip includes capi/*.hpp and host/*.hpp
    if force or _newer(LIB, srcs):
        _run([hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall",
              "-o", LIB, os.path.join(CSRC, "capi.hip"), "-lz"])
    return LIB


def build_tools(force=False):
    """Host tools (C++), linked against the C-ABI library."""
    host = os.path.join(CSRC, "host")
    if not os.path.isdir(host):
        return []
    os.makedirs(BIN_DIR, exist_ok=True)
    built = []
    srcs = _sources(host, os.path.join(ROOT, "include"))
    for tool in ("index_and_search",):
        src = os.path.join(host, tool + ".cpp")
        if not os.path.exists(src):
            continue
        out = os.path.join(BIN_DIR, tool)
        if force or _newer(out, srcs + [LIB]):
            _run([hipcc(), "-O2", "-std=c++17", "-Wall", "-o", out, src, "-L" + PKG, "-lcommet_hip",
                  "-Wl,-rpath,$ORIGIN/..", "-lpthread", "-lz"])
        built.append(out)
    for tool in ("bvop", "filter_reads", "extract_reads"):
        src = os.path.join(host, tool + ".cpp")
        if not os.path.exists(src):
            continue
        out = os.path.join(BIN_DIR, tool)
        if force or _newer(out, srcs):
            _run(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"), "-o", out, src, "-lz"])
        built.append(out)
    return built


def build_all(force=False):
    build_lib(force)
    build_tools(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
