"""Pins the CPU checker (oracle/) against golden vectors produced by the
reference itself (tests/golden/make_golden.py): HashKey values, whole-tool
outputs on randomised scenarios, and the reference's own ABCDE dataset."""
import gzip
import hashlib
import json
import os
import shutil

import numpy as np
import pytest

import oracle_binding as ob
import util
from scenarios import GoldenScenario, run_oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_hash_keys_match_reference_kat():
    cases = json.load(open(os.path.join(GOLD, "keys_kat.json")))
    assert len(cases) > 100
    checked = 0
    for c in cases:
        keys, pos = ob.keys_of_read(c["seq"], c["k"], bool(c["reverse"]))
        assert len(pos) == len(c["rows"]), c
        for (p, a, b, cc, d), kk, pp in zip(c["rows"], keys, pos):
            assert p == pp and (a, b, cc, d) == tuple(int(x) for x in kk), (c["k"], c["reverse"], c["seq"])
            # the bit-plane identities the HIP kernels rely on
            assert cc == a ^ b and d == a | b
            checked += 1
    assert checked > 2000


def test_max_kmer_constants():
    # SURVEY Q3 (index_and_search.cpp:73,146)
    assert ob.max_kmer(33) == 10 ** 9
    assert ob.max_kmer(32) == 5 * 10 ** 8
    assert ob.max_kmer(21) == 244140
    assert ob.max_kmer(20) == 122070
    assert ob.max_kmer(34) == 2 * 10 ** 9


@pytest.mark.parametrize("name", GoldenScenario.names())
def test_oracle_reproduces_reference_outputs(tmp_path, name):
    scn = GoldenScenario(name)
    out, log = str(tmp_path / "out"), str(tmp_path / "log")
    rc, res, chunks, kmers = run_oracle(scn, out, log)
    assert rc == 0
    scn.check_against_golden(out, log)


def test_bv_roundtrip_and_format(tmp_path):
    rng = np.random.default_rng(3)
    for n in (0, 1, 7, 8, 9, 64, 1000):
        b = rng.random(n) < 0.5
        p = str(tmp_path / f"x{n}.bv")
        bits = util.bits_from_bools(b)
        assert ob.load().ok_bv_write(p.encode(), b"some/file.fa in setX", bits.ctypes.data, n) == 0
        raw = open(p, "rb").read()
        head = b"some/file.fa in setX\n#%d\n" % n
        assert raw.startswith(head) and len(raw) == len(head) + n // 8 + 1      # boolean_vector.h:302-346
        assert oct(os.stat(p).st_mode & 0o777) == "0o600"
        cm, nn, got = util.read_bv(p)
        assert nn == n and np.array_equal(util.bools_from_bits(got, n), b)
        assert ob.load().ok_bv_nb_one(bits.ctypes.data, n) == int(b.sum())


@pytest.fixture(scope="module")
def abcde_dir(tmp_path_factory):
    src = os.path.join(GOLD, "abcde")
    if not os.path.exists(os.path.join(src, "expected.json")):
        pytest.skip("ABCDE golden data not generated")
    d = tmp_path_factory.mktemp("abcde")
    os.makedirs(d / "ABCDE_bench")
    for f, copies in (("A", "A"), ("B", "BD"), ("C", "CE")):      # B==D, C==E byte-identical in the reference
        data = gzip.open(os.path.join(src, f + ".fa.gz")).read()
        for c in copies:
            open(d / "ABCDE_bench" / (c + ".fa"), "wb").write(data)
    return str(d)


@pytest.mark.slow
@pytest.mark.parametrize("label", ["three_sets"])
def test_oracle_abcde_matrix(abcde_dir, label):
    """config[0] of BASELINE.json on the CPU checker: the reference's ABCDE_bench through
    Commet.py's job sequence, every .bv byte-identical to the reference's."""
    import sys
    sys.path.insert(0, GOLD)
    from make_golden import commet_jobs
    exp = json.load(open(os.path.join(GOLD, "abcde", "expected.json")))[label]
    sets = [(n, f) for n, f in exp["sets"]]
    names = [s[0] for s in sets]
    out = "out_oracle_" + label
    os.makedirs(os.path.join(abcde_dir, out), exist_ok=True)

    def cfg_line(si, restrict_to=None):
        name, files = sets[si]
        parts = [f if restrict_to is None else f + "," + out + "/" + os.path.basename(f) + "_in_" + names[restrict_to] + ".bv"
                 for f in files]
        return name + ":" + ";".join(parts)

    cwd = os.getcwd()
    os.chdir(abcde_dir)
    try:
        for j, (kind, idx, searches, restr) in enumerate(commet_jobs(names)):
            open("i.txt", "w").write(cfg_line(idx, restr) + "\n")
            open("s.txt", "w").write("".join(cfg_line(s) + "\n" for s in searches))
            rc, *_ = ob.index_and_search("i.txt", "s.txt", out, out, exp["k"], exp["t"])
            assert rc == 0
    finally:
        os.chdir(cwd)
    for f, h in exp["sha256"].items():
        got = open(os.path.join(abcde_dir, out, f), "rb").read().replace(out.encode() + b"/", b"out/")
        assert hashlib.sha256(got).hexdigest() == h, f
