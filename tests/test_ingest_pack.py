"""The host half of the ingest (commet_amd/csrc/host/ingest_pack.hpp: threaded FASTA / FASTQ parser + 2-bit plane
packer, AVX2 or scalar) against a numpy restatement of the layout, through the CPU-only driver host/ingest_check.cpp
(the HIP library runs the same code in front of hipMemcpyAsync).  COMMET_INGEST_CHECK: a prebuilt (sanitizer) driver."""
import os
import subprocess

import numpy as np
import pytest

import util
from conftest import ROOT

SRC = os.path.join(ROOT, "commet_amd", "csrc", "host", "ingest_check.cpp")
EXE = os.environ.get("COMMET_INGEST_CHECK") or os.path.join(ROOT, "commet_amd", "bin", "ingest_check")


@pytest.fixture(scope="module", autouse=True)
def _built():
    if os.environ.get("COMMET_INGEST_CHECK"):
        return
    deps = [SRC] + [os.path.join(os.path.dirname(SRC), h) for h in ("ingest_pack.hpp", "fasta_source.hpp")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        os.makedirs(os.path.dirname(EXE), exist_ok=True)
        subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"), "-o", EXE, SRC, "-lz"], check=True)


def expected(reads_per_file):
    """planes image of the set: read r starts at triple (goff[r] >> 5) + r; bit j of word w = base 32 w + j;
    hi = G/T, lo = C/T, valid = ACGTacgt (kernels.hpp); unused triples are zero"""
    reads = [r for fl in reads_per_file for r in fl]
    n = len(reads)
    goff = np.zeros(n + 1, dtype=np.uint64)
    if n:
        goff[1:] = np.cumsum([len(r) for r in reads])
    triples = (int(goff[-1]) >> 5) + n + 1
    planes = np.zeros((triples, 3), dtype=np.uint32)
    for r, seq in enumerate(reads):
        if not seq:
            continue
        a = np.frombuffer(seq, dtype=np.uint8)
        u = a & 0xDF
        valid = (u == ord("A")) | (u == ord("C")) | (u == ord("G")) | (u == ord("T"))
        hi = valid & (((a >> 2) & 1) == 1)
        lo = valid & ((((a >> 1) ^ (a >> 2)) & 1) == 1)
        t0 = (int(goff[r]) >> 5) + r
        for col, bits in enumerate((hi, lo, valid)):
            pad = np.zeros((-len(bits)) % 32, dtype=bool)
            words = np.packbits(np.concatenate([bits, pad]), bitorder="little").view("<u4")
            planes[t0:t0 + len(words), col] = words
    return goff, planes


def run_check(tmp_path, files, extra=()):
    out = str(tmp_path / "out.bin")
    p = subprocess.run([EXE] + list(extra) + [out] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    raw = open(out, "rb").read()
    head = np.frombuffer(raw[:56], dtype=np.uint64)
    n_reads, n_bases, triples, mn, mx, n_empty, n_files = (int(x) for x in head)
    pos = 56
    file_reads = np.frombuffer(raw[pos:pos + 8 * n_files], dtype=np.uint64); pos += 8 * n_files
    empty = np.frombuffer(raw[pos:pos + 8 * n_empty], dtype=np.uint64); pos += 8 * n_empty
    goff = np.frombuffer(raw[pos:pos + 8 * n_reads], dtype=np.uint64); pos += 8 * n_reads
    planes = np.frombuffer(raw[pos:pos + 12 * triples], dtype=np.uint32).reshape(triples, 3)
    return dict(n_reads=n_reads, n_bases=n_bases, min_len=mn, max_len=mx, empty=empty, file_reads=file_reads, goff=goff, planes=planes)


def _check(tmp_path, specs, extra=()):
    """specs: [(format, reads, multiline, crlf)] one per file"""
    rng = np.random.default_rng(3)
    files, per_file = [], []
    for i, (fmt, reads, multiline, crlf) in enumerate(specs):
        path = str(tmp_path / f"f{i}.{fmt}")
        util.write_reads(path, reads, fmt, rng=rng, multiline=multiline, crlf=crlf)
        files.append(path)
        per_file.append(util.parse_reads(path))          # the sequences as the tools see them ('\r' kept, lines joined)
    goff, planes = expected(per_file)
    for mode in ((), ("--arrays",)):
        got = run_check(tmp_path, files, tuple(extra) + mode)
        reads = [r for fl in per_file for r in fl]
        assert got["n_reads"] == len(reads) and got["n_bases"] == int(goff[-1])
        assert list(got["file_reads"]) == [len(fl) for fl in per_file]
        assert np.array_equal(got["goff"], goff[:-1])
        assert np.array_equal(got["planes"], planes), mode
        lens = [len(r) for r in reads]
        assert (got["min_len"], got["max_len"]) == ((min(lens), max(lens)) if lens else (0, 0))
        assert list(got["empty"]) == [i for i, ln in enumerate(lens) if ln == 0]


@pytest.mark.parametrize("seed", range(6))
def test_packer_matches_layout_on_random_files(tmp_path, seed):
    rng = np.random.default_rng(seed)
    specs = []
    for f in range(int(rng.integers(1, 4))):
        reads = util.random_reads(rng, int(rng.integers(1, 400)), 1, 300, n_rate=0.03, lower_rate=0.2, other_rate=0.01)
        fmt = ["fa", "fq", "fa.gz", "fq.gz"][int(rng.integers(0, 4))]
        specs.append((fmt, reads, bool(rng.integers(0, 2)) and fmt.startswith("fa"), bool(seed == 3)))
    _check(tmp_path, specs, extra=("--threads", "3"))


def test_reads_longer_than_a_staging_buffer_and_tiny_batches(tmp_path):
    """staging buffers of 7 triples / 3 reads: every buffer boundary falls inside reads; one read of 70 000 bases on one
    line, one spread over 61-column lines (the word boundary is crossed inside most lines)"""
    rng = np.random.default_rng(11)
    long1 = util.random_reads(rng, 1, 70_000, 70_000, n_rate=0.001)[0]
    long2 = util.random_reads(rng, 1, 33_333, 33_333, n_rate=0.001)[0]
    reads = util.random_reads(rng, 50, 1, 200) + [long1] + util.random_reads(rng, 20, 31, 33, lower_rate=0) + [long2]
    path = str(tmp_path / "long.fa")
    with open(path, "wb") as fh:
        for i, r in enumerate(reads):
            fh.write(b">r%d\n" % i)
            if r is long2:
                for j in range(0, len(r), 61):
                    fh.write(r[j:j + 61] + b"\n")
            else:
                fh.write(r + b"\n")
    goff, planes = expected([util.parse_reads(path)])
    for extra in (("--stage-triples", "7", "--stage-reads", "3", "--threads", "2"), ("--stage-triples", "100", "--arrays")):
        got = run_check(tmp_path, [path], extra)
        assert np.array_equal(got["planes"], planes) and np.array_equal(got["goff"], goff[:-1]), extra
        assert got["max_len"] == 70_000


def test_many_pieces_many_threads(tmp_path):
    """a file large enough to be cut into pieces (> 8 MiB): 8 workers, pieces uploaded out of order"""
    rng = np.random.default_rng(5)
    n, L = 90_000, 100
    codes = rng.integers(0, 4, size=n * L, dtype=np.uint8)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    bases[rng.integers(0, n * L, size=n)] = ord("N")
    from commet_amd import synth
    path = str(tmp_path / "big.fa")
    synth.write_fasta_fast(path, bases, n, L)
    assert os.path.getsize(path) > (8 << 20)
    got = run_check(tmp_path, [path], ("--threads", "8", "--stage-triples", "5000", "--stage-reads", "700"))
    goff, planes = expected([[bases[i * L:(i + 1) * L].tobytes() for i in range(n)]])
    assert got["n_reads"] == n and np.array_equal(got["planes"], planes) and np.array_equal(got["goff"], goff[:-1])
    assert (got["min_len"], got["max_len"]) == (L, L)
