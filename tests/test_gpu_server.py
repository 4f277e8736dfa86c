"""Resident mode of the drop-in tool (`index_and_search --serve SOCKET`, clients with COMMET_SERVER=SOCKET): the same CLI
tests as tests/test_gpu_cli.py, run through ONE server process that keeps its contexts and the read sets it has loaded
in HBM — stdout, exit codes, `.bv` bytes and log lines must not change, and the sets must really be reused."""
import os
import re
import subprocess
import time

import pytest

import test_gpu_cli as cli
from test_gpu_cli import abcde_dir  # noqa: F401  (fixture)
from scenarios import GoldenScenario

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def server(tmp_path_factory):
    from commet_amd import build
    build.build_lib()
    build.build_tools()
    sock = str(tmp_path_factory.mktemp("srv") / "s.sock")
    p = subprocess.Popen([cli.TOOL, "--serve", sock], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    for _ in range(600):
        if os.path.exists(sock) or p.poll() is not None:
            break
        time.sleep(0.1)
    assert p.poll() is None and os.path.exists(sock), "the server did not come up"
    old = os.environ.get("COMMET_SERVER")
    os.environ["COMMET_SERVER"] = sock
    yield sock
    subprocess.run([cli.TOOL, "--server-stop"], timeout=60)
    p.wait(timeout=60)
    if old is None:
        del os.environ["COMMET_SERVER"]
    else:
        os.environ["COMMET_SERVER"] = old
    assert not os.path.exists(sock)


def _stats():
    out = subprocess.run([cli.TOOL, "--server-stats"], stdout=subprocess.PIPE, check=True).stdout.decode()
    return {k: int(v) for k, v in re.findall(r"(requests|sets resident|cache hits|loads|evictions|contexts) (\d+)", out)}


def test_flags_errors_and_golden_scenarios_through_the_server(server, tmp_path):
    cli.test_cli_flags_and_errors(tmp_path)
    for i, name in enumerate(GoldenScenario.names()):
        d = tmp_path / f"g{i}"
        os.makedirs(d)
        cli.test_cli_reproduces_reference_golden(d, name)
    st = _stats()
    assert st["requests"] >= len(GoldenScenario.names()) and st["contexts"] >= 2


@pytest.mark.parametrize("label", ["three_sets", "five_sets"])
def test_abcde_job_sequence_reuses_resident_sets(server, abcde_dir, label):  # noqa: F811
    before = _stats()
    cli.test_cli_abcde_matrix(abcde_dir, label)            # Commet.py's N^2 - 1 invocations, byte-compared with the reference
    after = _stats()
    n = 3 if label == "three_sets" else 5
    jobs = n * n - 1
    assert after["requests"] - before["requests"] == jobs
    # every set is parsed once (the five-set run finds the three-set run's A, B, C already resident)
    assert after["loads"] - before["loads"] <= n
    assert after["cache hits"] - before["cache hits"] >= 2 * jobs - n


def test_full_mode_and_changed_file(server, tmp_path):
    full = [n for n in GoldenScenario.names() if os.path.isdir(os.path.join(GoldenScenario(n).dir, "expected_full"))]
    assert full
    for i, name in enumerate(full[:4]):
        d = tmp_path / f"f{i}"
        os.makedirs(d)
        cli.test_cli_full_mode_reproduces_reference_golden(d, name)
    # a file that changes (size / mtime) is loaded again, not served from the cache
    import numpy as np
    import util
    rng = np.random.default_rng(3)
    d = tmp_path / "chg"
    os.makedirs(d)
    reads = util.random_reads(rng, 300, 40, 90)
    util.write_fasta(str(d / "a.fa"), reads)
    util.write_fasta(str(d / "b.fa"), util.related_reads(rng, reads, 200, 40, 90, share=0.7))
    (d / "i.txt").write_text("A:a.fa\n")
    (d / "s.txt").write_text("B:b.fa\n")
    cmd = [cli.TOOL, "-i", "i.txt", "-s", "s.txt", "-o", "o", "-l", "o", "-k", "16", "-t", "1"]
    r1 = subprocess.run(cmd, cwd=str(d), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    line1 = [ln for ln in r1.stdout.decode().splitlines() if ln.startswith("[indexed")][-1]
    s1 = _stats()
    r2 = subprocess.run(cmd, cwd=str(d), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert [ln for ln in r2.stdout.decode().splitlines() if ln.startswith("[indexed")][-1] == line1
    s2 = _stats()
    assert s2["cache hits"] - s1["cache hits"] == 2 and s2["loads"] == s1["loads"]
    time.sleep(0.02)
    util.write_fasta(str(d / "a.fa"), reads[:100])                                  # the index set shrinks
    r3 = subprocess.run(cmd, cwd=str(d), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    line3 = [ln for ln in r3.stdout.decode().splitlines() if ln.startswith("[indexed")][-1]
    assert line3.startswith("[indexed 100,") and line3 != line1
    assert _stats()["loads"] == s2["loads"] + 1
