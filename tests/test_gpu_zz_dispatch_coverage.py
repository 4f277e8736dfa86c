"""Last file of the -m gpu suite (pytest takes the files in name order): every kernel instantiation compiled into
libcommet_hip.so must have been launched by the parity tests that ran before in this process.

The library notes the entry point of every launch (commet_launched_kernels); the addresses are resolved against the
library's own symbol table (nm), so the list of instantiations is the compiler's, not a hand-kept one: a new template
argument behind the hand-written dispatch of capi.hip that no test reaches fails here.  Microbenchmark kernels are
exempt (they decide nothing)."""
import ctypes as C
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

EXEMPT = ("membench_kernel", "membench_window_kernel", "ldsbench_kernel")


def _symbols(path):
    """{offset: demangled kernel instantiation} of the kernel handles and host stubs, plus commet_version's offset"""
    out, version_at = {}, None
    for ln in subprocess.run(["nm", "-C", "--defined-only", path], check=True, capture_output=True, text=True).stdout.split("\n"):
        parts = ln.split(" ", 2)
        if len(parts) < 3:
            continue
        addr, _, name = parts
        if name == "commet_version":
            version_at = int(addr, 16)
        m = re.match(r"(?:void )?commet::(?:__device_stub__)?(\w+_kernel(?:<[^(]*>)?)\(", name)
        if m:
            out[int(addr, 16)] = m.group(1).replace("unsigned int", "u32").replace("unsigned long", "u64")
    return out, version_at


def test_every_compiled_kernel_instantiation_was_launched_by_the_suite():
    from commet_amd import lib as L
    h = L.load()
    syms, version_at = _symbols(L.LIB_PATH)
    compiled = {n for n in syms.values() if not n.startswith(EXEMPT)}
    assert len(compiled) >= 60 and version_at is not None
    base = C.cast(h.commet_version, C.c_void_p).value - version_at
    n = C.c_int(0)
    arr = (C.c_void_p * 512)()
    assert h.commet_launched_kernels(arr, 512, C.byref(n)) == 0
    launched = set()
    for i in range(min(n.value, 512)):
        off = arr[i] - base
        assert off in syms, f"launched entry point at +{off:#x} is not a kernel symbol of the library"
        launched.add(syms[off])
    if len(launched) < 12:
        pytest.skip("run as the last file of the whole -m gpu suite (this process has launched only %d kernels)" % len(launched))
    missing = sorted(compiled - launched)
    assert not missing, "kernel instantiations no test of the suite reached: " + ", ".join(missing)
