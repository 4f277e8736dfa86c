"""bvop and filter_reads (host-only C++ tools Commet.py needs) against the
compiled reference tools, byte for byte: stdout, exit code and the `.bv` files."""
import os
import re
import subprocess

import numpy as np
import pytest

import util
from conftest import ROOT, ref_tool

BIN = os.path.join(ROOT, "commet_amd", "bin")


@pytest.fixture(scope="module", autouse=True)
def _tools_built():
    if not all(os.path.exists(os.path.join(BIN, t)) for t in ("bvop", "filter_reads")):
        from commet_amd import build
        os.makedirs(BIN, exist_ok=True)
        for tool in ("bvop", "filter_reads"):
            subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o",
                            os.path.join(BIN, tool), os.path.join(ROOT, "commet_amd", "csrc", "host", tool + ".cpp"), "-lz"],
                           check=True)


def _run(tool, args, cwd):
    return subprocess.run([tool] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def _both(name, args, cwd_ours, cwd_ref):
    ref = ref_tool(name)
    if not ref:
        pytest.skip(f"oracle/_ref/{name} not built")
    a = _run(os.path.join(BIN, name), args, cwd_ours)
    b = _run(ref, args, cwd_ref)
    return a, b


def _strip_time(b):
    return re.sub(rb"Total  time : .* s", b"Total  time : T s", b)


def _make_fasta(path, seed, n=300):
    rng = np.random.default_rng(seed)
    reads = util.random_reads(rng, n, 1, 120, n_rate=0.03, lower_rate=0.2, other_rate=0.01)
    reads[5] = b"A" * 80                      # Shannon 0
    reads[6] = b"AC" * 40                     # Shannon 1
    reads[7] = b"N" * 30
    util.write_fasta(path, reads, rng=rng, multiline=(seed % 2 == 0))
    return reads


@pytest.mark.parametrize("seed,opts", [
    (1, []), (2, ["-l", "50"]), (3, ["-n", "2"]), (4, ["-e", "1.9"]), (5, ["-l", "30", "-n", "1", "-e", "1.5"]),
    (6, ["-m", "40"]), (7, ["-m", "0"]), (8, ["-l", "64", "-e", "1.95", "-m", "25"]), (9, ["-c", "my comment", "-l", "10"]),
    (10, ["-e", "0"]), (11, ["-m", "300"]), (12, ["-n", "0", "-m", "7"]),
])
def test_filter_reads_matches_reference(tmp_path, seed, opts):
    for d in ("ours", "ref"):
        os.makedirs(tmp_path / d)
        _make_fasta(str(tmp_path / d / "reads.fa"), seed)
    a, b = _both("filter_reads", ["reads.fa"] + opts + ["-o", "out.bv"], str(tmp_path / "ours"), str(tmp_path / "ref"))
    assert a.returncode == b.returncode
    assert _strip_time(a.stdout) == _strip_time(b.stdout)
    assert open(tmp_path / "ours" / "out.bv", "rb").read() == open(tmp_path / "ref" / "out.bv", "rb").read()


def test_filter_reads_default_output_name_and_errors(tmp_path):
    for d in ("ours", "ref"):
        os.makedirs(tmp_path / d / "sub")
        _make_fasta(str(tmp_path / d / "sub" / "x.fa"), 3, n=50)
    a, b = _both("filter_reads", ["sub/x.fa", "-l", "20"], str(tmp_path / "ours"), str(tmp_path / "ref"))
    assert a.returncode == b.returncode == 0 and _strip_time(a.stdout) == _strip_time(b.stdout)
    assert open(tmp_path / "ours" / "sub" / "x.fa.bv", "rb").read() == open(tmp_path / "ref" / "sub" / "x.fa.bv", "rb").read()
    a, b = _both("filter_reads", ["nope.fa"], str(tmp_path / "ours"), str(tmp_path / "ref"))
    assert a.returncode == b.returncode == 1 and a.stderr == b.stderr
    a, b = _both("filter_reads", ["sub/x.fa", "-z"], str(tmp_path / "ours"), str(tmp_path / "ref"))
    assert a.returncode == b.returncode == 1 and a.stderr == b.stderr and a.stdout == b.stdout
    a, b = _both("filter_reads", ["-v"], str(tmp_path / "ours"), str(tmp_path / "ref"))
    assert (a.returncode, a.stdout) == (b.returncode, b.stdout)


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 100, 1001])
def test_bvop_matches_reference(tmp_path, n):
    rng = np.random.default_rng(n)
    for d in ("ours", "ref"):
        os.makedirs(tmp_path / d)
        util.write_bv(str(tmp_path / d / "a.bv"), "file a\nsecond line", rng.random(n) < 0.5)
        util.write_bv(str(tmp_path / d / "b.bv"), "file b", rng.random(n) < 0.3)
        rng = np.random.default_rng(n)      # same content in both dirs
    for args in (["a.bv", "-i"], ["a.bv", "-a", "b.bv", "-p", "o.bv"], ["a.bv", "-o", "b.bv", "-p", "o.bv"],
                 ["a.bv", "-d", "b.bv", "-p", "o.bv"], ["a.bv", "-n", "-p", "o.bv"], ["a.bv", "-n"],
                 ["a.bv", "-a", "b.bv", "-i"], ["a.bv", "-n", "-i", "-p", "o.bv"], ["-v"], ["-h"], ["a.bv", "b.bv"]):
        a, b = _both("bvop", args, str(tmp_path / "ours"), str(tmp_path / "ref"))
        assert a.returncode == b.returncode, args
        assert a.stdout == b.stdout, args
        assert a.stderr == b.stderr, args
        if "-p" in args:
            assert open(tmp_path / "ours" / "o.bv", "rb").read() == open(tmp_path / "ref" / "o.bv", "rb").read(), args


def test_bvop_info_line_is_what_commet_py_parses(tmp_path):
    sel = np.zeros(1000, dtype=bool)
    sel[::3] = True
    util.write_bv(str(tmp_path / "x.bv"), "some/path.fa in setB", sel)
    out = _run(os.path.join(BIN, "bvop"), ["x.bv", "-i"], str(tmp_path)).stdout.decode()
    assert int(out.split("\n")[-2].split()[0]) == int(sel.sum())          # Commet.py:257
    assert out == "some/path.fa in setB\nReads:\n  334 / 1000 reads selected\n"


def test_bvop_size_mismatch(tmp_path):
    util.write_bv(str(tmp_path / "a.bv"), "a", [True] * 10)
    util.write_bv(str(tmp_path / "b.bv"), "b", [True] * 11)
    r = _run(os.path.join(BIN, "bvop"), ["a.bv", "-a", "b.bv"], str(tmp_path))
    assert r.returncode == 1 and b"not the same size" in r.stderr
