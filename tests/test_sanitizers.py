"""Sanitizer builds of the host C++ (CPU only; the GPU pool offers no sanitizers): the chunk planner
(read_iter.hpp behind libcommet_plan.so) and the host tools bvop / filter_reads / extract_reads are compiled with
-fsanitize=address,undefined, the multi-threaded filter_reads and the ingest parser also with -fsanitize=thread,
and the existing scenario tests are run through those builds in a child pytest.  Any sanitizer report aborts the
instrumented process, which fails the child run."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "commet_amd", "csrc", "host")
SAN = os.path.join(ROOT, "commet_amd", "_san")
ASAN_FLAGS = ["-g", "-O1", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
TSAN_FLAGS = ["-g", "-O1", "-fno-omit-frame-pointer", "-fsanitize=thread"]
ENV = dict(ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
           TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")


def _gxx(out, args):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include")] + args + ["-o", out], check=True)
    return out


def _child_pytest(files, env, select=None):
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + files + (["-k", select] if select else [])
    p = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, **ENV, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    tail = p.stdout.decode()[-3000:]
    assert p.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
    return tail


def test_planner_under_asan_ubsan():
    lib = _gxx(os.path.join(SAN, "asan", "libcommet_plan.so"), ASAN_FLAGS + ["-shared", "-fPIC", os.path.join(HOST, "plan_capi.cpp")])
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE, check=True).stdout.decode().strip()
    _child_pytest(["tests/test_host_plan.py"], dict(COMMET_PLAN_LIB=lib, LD_PRELOAD=asan_rt))


def test_host_tools_under_asan_ubsan():
    d = os.path.join(SAN, "asan", "bin")
    for tool in ("bvop", "filter_reads", "extract_reads"):
        _gxx(os.path.join(d, tool), ASAN_FLAGS + [os.path.join(HOST, tool + ".cpp"), "-lz"])
    _child_pytest(["tests/test_host_tools.py"], dict(COMMET_BIN_DIR=d))


def test_threaded_filter_reads_under_tsan():
    d = os.path.join(SAN, "tsan", "bin")
    for tool in ("bvop", "filter_reads", "extract_reads"):
        _gxx(os.path.join(d, tool), TSAN_FLAGS + [os.path.join(HOST, tool + ".cpp"), "-lz"])
    _child_pytest(["tests/test_host_tools.py"], dict(COMMET_BIN_DIR=d, COMMET_INGEST_THREADS="4", COMMET_FILTER_PIECE_BYTES="1500"), select="filter or extract")


def test_ingest_parser_and_packer_under_asan_ubsan_and_tsan():
    """the threaded parser + 2-bit packer libcommet_hip.so runs in front of hipMemcpyAsync (host/ingest_pack.hpp), through
    its CPU-only driver: out-of-order pieces, staging buffers that end inside reads, 8 workers"""
    src = os.path.join(HOST, "ingest_check.cpp")
    exe = _gxx(os.path.join(SAN, "asan", "bin", "ingest_check"), ASAN_FLAGS + [src, "-lz"])
    _child_pytest(["tests/test_ingest_pack.py"], dict(COMMET_INGEST_CHECK=exe))
    exe = _gxx(os.path.join(SAN, "tsan", "bin", "ingest_check"), TSAN_FLAGS + [src, "-lz"])
    _child_pytest(["tests/test_ingest_pack.py"], dict(COMMET_INGEST_CHECK=exe))
