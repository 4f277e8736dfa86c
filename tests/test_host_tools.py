"""bvop, filter_reads (host-only C++ tools Commet.py needs) and extract_reads (8f-4) against
the compiled reference tools, byte for byte: stdout, exit code and the files written."""
import os
import re
import subprocess

import numpy as np
import pytest

import util
from conftest import ROOT, ref_tool

# COMMET_BIN_DIR: prebuilt (e.g. sanitizer-instrumented, tests/test_sanitizers.py) tools to test instead
BIN = os.environ.get("COMMET_BIN_DIR") or os.path.join(ROOT, "commet_amd", "bin")


@pytest.fixture(scope="module", autouse=True)
def _tools_built():
    if not all(os.path.exists(os.path.join(BIN, t)) for t in ("bvop", "filter_reads", "extract_reads")):
        from commet_amd import build
        os.makedirs(BIN, exist_ok=True)
        for tool in ("bvop", "filter_reads", "extract_reads"):
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


# ---------------------------------------------------------------- extract_reads (src/extract_reads.cpp)
def _make_reads(path, fmt, seed, n=200, **kw):
    rng = np.random.default_rng(seed)
    reads = util.random_reads(rng, n, 1, 150, n_rate=0.02, lower_rate=0.1)
    util.write_reads(path, reads, fmt, rng=rng, **kw)
    return reads


@pytest.mark.parametrize("fmt", ["fa", "fq", "fa.gz", "fq.gz"])
@pytest.mark.parametrize("seed,density,kw", [(1, 0.5, {}), (2, 0.1, {"multiline": True}), (3, 1.0, {}), (4, 0.0, {}),
                                             (5, 0.7, {"crlf": True}), (6, 0.3, {"multiline": True})])
def test_extract_reads_matches_reference(tmp_path, fmt, seed, density, kw):
    for d in ("ours", "ref"):
        os.makedirs(tmp_path / d)
        reads = _make_reads(str(tmp_path / d / ("reads." + fmt)), fmt, seed, **kw)
        sel = np.random.default_rng(100 + seed).random(len(reads)) < density
        util.write_bv(str(tmp_path / d / "sel.bv"), "reads in x", sel)
    args = ["reads." + fmt, "sel.bv", "-o", "out"]
    a, b = _both("extract_reads", args, str(tmp_path / "ours"), str(tmp_path / "ref"))
    assert a.returncode == b.returncode == 0
    assert a.stdout == b.stdout and a.stderr == b.stderr
    ours, ref = open(tmp_path / "ours" / "out", "rb").read(), open(tmp_path / "ref" / "out", "rb").read()
    assert ours == ref                                   # gzip inputs: identical compressed bytes too
    if fmt.endswith(".gz"):
        import gzip
        ours = gzip.decompress(ours)
    if not kw.get("multiline") and not kw.get("crlf"):    # what was extracted is what the bv selects
        got = util.parse_reads(str(tmp_path / "ours" / "out"))
        assert got == [r for r, s in zip(reads, sel) if s]
    if not fmt.endswith(".gz"):                           # stdout sink
        a, b = _both("extract_reads", args[:2], str(tmp_path / "ours"), str(tmp_path / "ref"))
        assert a.returncode == b.returncode == 0 and a.stdout == b.stdout == ours


def test_extract_reads_round_trip_with_filter_reads(tmp_path):
    """filter_reads -> extract_reads -> filter_reads: every extracted read passes the same filter."""
    reads = _make_fasta(str(tmp_path / "reads.fa"), 11)
    fr, er = os.path.join(BIN, "filter_reads"), os.path.join(BIN, "extract_reads")
    assert _run(fr, ["reads.fa", "-l", "60", "-n", "1", "-o", "f.bv"], str(tmp_path)).returncode == 0
    assert _run(er, ["reads.fa", "f.bv", "-o", "kept.fa"], str(tmp_path)).returncode == 0
    kept = util.parse_reads(str(tmp_path / "kept.fa"))
    _, n, bits = util.read_bv(str(tmp_path / "f.bv"))
    sel = util.bools_from_bits(bits, n)
    assert n == len(reads) and 0 < len(kept) == int(np.sum(sel)) < len(reads)
    assert _run(fr, ["kept.fa", "-l", "60", "-n", "1", "-o", "g.bv"], str(tmp_path)).returncode == 0
    _, n2, bits2 = util.read_bv(str(tmp_path / "g.bv"))
    assert n2 == len(kept) and all(util.bools_from_bits(bits2, n2))


def test_extract_reads_edge_cases(tmp_path):
    for d in ("ours", "ref"):
        os.makedirs(tmp_path / d)
        with open(tmp_path / d / "e.fa", "wb") as fh:       # empty record in the middle, no trailing newline
            fh.write(b">a\nACGT\n>b\n>c\nGGCC\n\nTT\n>d\nAAAA")
        util.write_bv(str(tmp_path / d / "all.bv"), "c", [True] * 4)
        util.write_bv(str(tmp_path / d / "some.bv"), "c", [True, False, True, True])
        util.write_bv(str(tmp_path / d / "short.bv"), "c", [True] * 3)
        with open(tmp_path / d / "q.fq", "wb") as fh:       # blank lines between records
            fh.write(b"@a\nACGT\n+\nIIII\n\n@b\nGG\n+b\nII\n\n\n@c\nTTT\n+\nIII\n")
        util.write_bv(str(tmp_path / d / "q.bv"), "c", [True, False, True])
    for args in (["e.fa", "all.bv"], ["e.fa", "some.bv"], ["e.fa", "short.bv"], ["q.fq", "q.bv"], ["e.fa"], [],
                 ["-v"], ["-h"], ["e.fa", "all.bv", "-x"], ["e.fa", "all.bv", "extra"], ["nope.fa", "all.bv"],
                 ["e.fa", "all.bv", "-o", "no/such/dir/out"]):
        a, b = _both("extract_reads", args, str(tmp_path / "ours"), str(tmp_path / "ref"))
        assert a.returncode == b.returncode, args
        assert a.stdout == b.stdout, args
        assert a.stderr == b.stderr, args
