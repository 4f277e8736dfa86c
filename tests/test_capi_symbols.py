"""The C-ABI library loads on a GPU-less box and exports every symbol include/commet_hip.h declares
(no compute calls here); without a device the path refuses to run — there is no CPU fallback."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "commet_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(commet_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    from commet_amd import build, lib
    if not os.path.exists(lib.LIB_PATH):
        build.build_lib()
    return lib.LIB_PATH


def test_header_symbols_are_exported_and_bound(lib_path):
    from commet_amd import lib
    names = declared_symbols()
    assert len(names) >= 25
    h = ctypes.CDLL(lib_path)
    for n in names:
        assert hasattr(h, n), f"{n} declared in commet_hip.h but not exported"
        assert n in lib.SIGNATURES, f"{n} has no ctypes signature in commet_amd/lib.py"
    assert sorted(lib.SIGNATURES) == names


def test_no_silent_cpu_fallback(lib_path):
    import commet_amd
    from commet_amd import lib
    h = lib.load()
    assert b"gfx950" in h.commet_version()
    if h.commet_device_count() == 0:
        with pytest.raises(commet_amd.CommetError, match="no HIP device|no CPU fallback"):
            commet_amd.Context(k=12)


def test_product_code_never_touches_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "commet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                t = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"oracle_binding|commet_oracle|liboracle|oracle/_ref|oracle_cli", t):
                    bad.append(f)
    assert not bad, f"product code references the CPU checker: {bad}"
