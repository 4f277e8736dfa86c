"""The drop-in tool (commet_amd/bin/index_and_search) at the process boundary
Commet.py uses: same flags, same .bv bytes, same log line as the reference."""
import gzip
import hashlib
import json
import os
import re
import subprocess
import sys

import pytest

import util
from scenarios import GoldenScenario, Scenario, compare_runs, run_oracle, run_tool
from conftest import ROOT, ref_tool

pytestmark = pytest.mark.gpu

TOOL = os.path.join(ROOT, "commet_amd", "bin", "index_and_search")
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module", autouse=True)
def _tool_built():
    if not os.path.exists(TOOL):
        from commet_amd import build
        build.build_lib()
        build.build_tools()
    assert os.path.exists(TOOL)


@pytest.mark.parametrize("name", GoldenScenario.names())
def test_cli_reproduces_reference_golden(tmp_path, name):
    scn = GoldenScenario(name)
    out, log = str(tmp_path / "out"), str(tmp_path / "log")
    p = run_tool(TOOL, scn, out, log)
    assert p.returncode == 0, p.stderr.decode()[-800:]
    scn.check_against_golden(out, log)


def _strip_times(text):
    return re.sub(r"(Index  time|Search time|Total  time): .* s", r"\1: T s", text)


@pytest.mark.parametrize("seed", range(2000, 2050))
def test_cli_matches_oracle_and_reference(tmp_path, seed):
    mixed = seed >= 2030        # FASTQ / gzip inputs, format chosen per file
    scn = Scenario(str(tmp_path / "scn"), seed, **({"formats": ("fa", "fq", "fa.gz", "fq.gz"), "crlf": False} if mixed else {}))
    out_g, log_g = str(tmp_path / "out_gpu"), str(tmp_path / "log_gpu")
    out_o, log_o = str(tmp_path / "out_orc"), str(tmp_path / "log_orc")
    p = run_tool(TOOL, scn, out_g, log_g)
    assert p.returncode == 0, p.stderr.decode()[-800:]
    rc, *_ = run_oracle(scn, out_o, log_o)
    assert rc == 0
    compare_runs(out_o, log_o, out_g, log_g, scn)
    ref = ref_tool("index_and_search")
    if ref:   # the compiled reference, when it travelled with the snapshot: stdout must match too
        out_r, log_r = str(tmp_path / "out_ref"), str(tmp_path / "log_ref")
        q = run_tool(ref, scn, out_r, log_r)
        assert q.returncode == 0
        compare_runs(out_r, log_r, out_g, log_g, scn)
        assert _strip_times(q.stdout.decode()) == _strip_times(p.stdout.decode())
        for f in os.listdir(log_r):
            a = _strip_times(open(os.path.join(log_r, f)).read())
            b = _strip_times(open(os.path.join(log_g, f)).read())
            assert a == b
        for f in os.listdir(out_r):
            if f.endswith(".bv"):
                assert oct(os.stat(os.path.join(out_g, f)).st_mode & 0o777) == "0o600"


def _full_mode_outputs(out_dir, log_dir):
    bvs = {f: open(os.path.join(out_dir, f), "rb").read() for f in sorted(os.listdir(out_dir)) if f.endswith(".bv")}
    logs = {f: open(os.path.join(log_dir, f)).read().strip().split("\n")[3:] for f in sorted(os.listdir(log_dir))
            if f.endswith(".log")}
    return bvs, logs


@pytest.mark.parametrize("name", [n for n in GoldenScenario.names()
                                  if os.path.isdir(os.path.join(GOLD, "scenarios", n, "expected_full"))])
def test_cli_full_mode_reproduces_reference_golden(tmp_path, name):
    """-f: three-pass symmetric comparison of the index set and the first search set (index_and_search.cpp:304-391)"""
    scn = GoldenScenario(name)
    out, log = str(tmp_path / "out"), str(tmp_path / "log")
    p = run_tool(TOOL, scn, out, log, extra_args=["-f"])
    assert p.returncode == 0, p.stderr.decode()[-800:]
    exp_dir = os.path.join(scn.dir, "expected_full")
    bvs, logs = _full_mode_outputs(out, log)
    exp_bvs = {f: open(os.path.join(exp_dir, f), "rb").read() for f in sorted(os.listdir(exp_dir)) if f.endswith(".bv")}
    assert bvs == exp_bvs
    assert logs == json.load(open(os.path.join(exp_dir, "log_lines.json")))


@pytest.mark.parametrize("seed", range(2100, 2115))
def test_cli_full_mode_matches_reference_live(tmp_path, seed):
    ref = ref_tool("index_and_search")
    if not ref:
        pytest.skip("oracle/_ref/index_and_search did not travel with the snapshot")
    scn = Scenario(str(tmp_path / "scn"), seed)
    p = run_tool(TOOL, scn, str(tmp_path / "og"), str(tmp_path / "lg"), extra_args=["-f"])
    q = run_tool(ref, scn, str(tmp_path / "or"), str(tmp_path / "lr"), extra_args=["-f"])
    assert p.returncode == q.returncode == 0, p.stderr.decode()[-500:]
    assert _full_mode_outputs(str(tmp_path / "og"), str(tmp_path / "lg")) == _full_mode_outputs(str(tmp_path / "or"), str(tmp_path / "lr"))
    assert _strip_times(q.stdout.decode()) == _strip_times(p.stdout.decode())


def test_cli_flags_and_errors(tmp_path):
    r = subprocess.run([TOOL], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"Usage : ./index_and_search" in r.stderr
    r = subprocess.run([TOOL, "-v"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"index_and_search version" in r.stdout
    r = subprocess.run([TOOL, "-z"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"Unknown option -z" in r.stderr          # index_and_search.cpp:166-169
    r = subprocess.run([TOOL, "-k"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"needs an argument" in r.stderr
    r = subprocess.run([TOOL, "-i", "nonexistent.txt", "-s", "x"], cwd=str(tmp_path), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Cannot read file nonexistent.txt" in r.stderr
    # two index sets -> refused (index_and_search.cpp:197-200)
    (tmp_path / "a.fa").write_bytes(b">1\nACGTACGTACGTACGTACGTAAAA\n")
    (tmp_path / "two.txt").write_text("x:a.fa\ny:a.fa\n")
    (tmp_path / "one.txt").write_text("x:a.fa\n")
    r = subprocess.run([TOOL, "-i", "two.txt", "-s", "one.txt", "-k", "8"], cwd=str(tmp_path), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Only one set of files is allowed for indexing" in r.stderr
    # bv of the wrong size (fasta_file.h:108-111)
    util.write_bv(str(tmp_path / "bad.bv"), "c", [True, False, True])
    (tmp_path / "bad.txt").write_text("x:a.fa,bad.bv\n")
    r = subprocess.run([TOOL, "-i", "bad.txt", "-s", "one.txt", "-k", "8"], cwd=str(tmp_path), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"boolean vector size are not equal" in r.stderr
    # -o / -l directories are created, a set-config without ':' is named SET<n>
    (tmp_path / "noname.txt").write_text("a.fa\n")
    r = subprocess.run([TOOL, "-i", "one.txt", "-s", "noname.txt", "-k", "8", "-t", "1", "-o", "newout", "-l", "newlog"],
                       cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "newout" / "a.fa_in_x.bv").exists() and (tmp_path / "newlog" / "SET1_in_x.log").exists()
    assert util.last_log_line(str(tmp_path / "newlog" / "SET1_in_x.log")) == "[indexed 1, searched 1, shared 1]"


@pytest.fixture(scope="module")
def abcde_dir(tmp_path_factory):
    src = os.path.join(GOLD, "abcde")
    d = tmp_path_factory.mktemp("abcde")
    os.makedirs(d / "ABCDE_bench")
    for f, copies in (("A", "A"), ("B", "BD"), ("C", "CE")):
        data = gzip.open(os.path.join(src, f + ".fa.gz")).read()
        for c in copies:
            open(d / "ABCDE_bench" / (c + ".fa"), "wb").write(data)
    return str(d)


@pytest.mark.parametrize("label", ["three_sets", "five_sets"])
def test_cli_abcde_matrix(abcde_dir, label):
    """BASELINE config[0]: the reference's ABCDE_bench, k=32 t=2, through Commet.py's N^2-1 job sequence
    (Commet.py:570-574, 186-240); every output .bv byte-identical to the reference's."""
    sys.path.insert(0, GOLD)
    from make_golden import commet_jobs
    exp = json.load(open(os.path.join(GOLD, "abcde", "expected.json")))[label]
    sets = [(n, f) for n, f in exp["sets"]]
    names = [s[0] for s in sets]
    out = "out"
    work = os.path.join(abcde_dir, label)
    os.makedirs(os.path.join(work, out))
    os.symlink(os.path.join(abcde_dir, "ABCDE_bench"), os.path.join(work, "ABCDE_bench"))

    def cfg_line(si, restrict_to=None):
        name, files = sets[si]
        parts = [f if restrict_to is None else f + "," + out + "/" + os.path.basename(f) + "_in_" + names[restrict_to] + ".bv"
                 for f in files]
        return name + ":" + ";".join(parts)

    jobs = commet_jobs(names)
    assert len(jobs) == len(names) ** 2 - 1
    for j, (kind, idx, searches, restr) in enumerate(jobs):
        open(os.path.join(work, "i.txt"), "w").write(cfg_line(idx, restr) + "\n")
        open(os.path.join(work, "s.txt"), "w").write("".join(cfg_line(s) + "\n" for s in searches))
        r = subprocess.run([TOOL, "-i", "i.txt", "-s", "s.txt", "-o", out, "-l", out, "-k", str(exp["k"]), "-t",
                            str(exp["t"])], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-500:]
    for f, h in exp["sha256"].items():
        got = open(os.path.join(work, out, f), "rb").read()
        exp_bytes = open(os.path.join(GOLD, "abcde", label, f), "rb").read()
        assert got == exp_bytes, f
        assert hashlib.sha256(got).hexdigest() == h
