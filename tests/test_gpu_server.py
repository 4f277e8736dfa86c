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


def test_multi_device_server_routes_by_resident_sets_and_runs_devices_side_by_side(tmp_path):
    """`--serve SOCKET --devices 2`: the process forks one single-device server per GPU before it touches the GPU and routes the clients'
    requests (round 6).  Rehearsed on ONE GPU (COMMET_SERVER_DEVICE_LIST=0,0: both children on device 0): jobs started in parallel are
    answered with the bytes of the plain tool, a job goes back to the device server that holds its files, and both servers get work."""
    import numpy as np  # noqa: F401
    import util
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(12)
    d = tmp_path / "w"
    os.makedirs(d)
    pools = [util.random_reads(rng, 400, 40, 120) for _ in range(4)]
    for i, pool in enumerate(pools):
        util.write_fasta(str(d / f"a{i}.fa"), pool)
        util.write_fasta(str(d / f"b{i}.fa"), util.related_reads(rng, pool, 300, 40, 120, share=0.6))
        (d / f"i{i}.txt").write_text(f"A{i}:a{i}.fa\n")
        (d / f"s{i}.txt").write_text(f"B{i}:b{i}.fa\n")
    env0 = {k: v for k, v in os.environ.items() if k != "COMMET_SERVER"}

    def cmd(i, out):
        return [cli.TOOL, "-i", f"i{i}.txt", "-s", f"s{i}.txt", "-o", out, "-l", out, "-k", "20", "-t", "2"]

    plain = []
    for i in range(4):                                             # the plain tool, one process per job
        r = subprocess.run(cmd(i, f"p{i}"), cwd=str(d), env=env0, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0
        plain.append([ln for ln in r.stdout.decode().splitlines() if ln.startswith("[indexed")][-1])
    sock = str(tmp_path / "m.sock")
    env_srv = dict(env0, COMMET_SERVER_DEVICE_LIST="0,0")
    p = subprocess.Popen([cli.TOOL, "--serve", sock, "--devices", "2"], env=env_srv, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    try:
        for _ in range(1200):
            if os.path.exists(sock) or p.poll() is not None:
                break
            time.sleep(0.1)
        assert p.poll() is None and os.path.exists(sock), "the multi-device server did not come up"
        env_c = dict(env0, COMMET_SERVER=sock)

        def job(i, rnd):
            r = subprocess.run(cmd(i, f"m{rnd}_{i}"), cwd=str(d), env=env_c, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0, r.stderr.decode()
            return [ln for ln in r.stdout.decode().splitlines() if ln.startswith("[indexed")][-1]

        for rnd in range(3):                                       # four jobs at a time, three rounds
            with ThreadPoolExecutor(4) as pool:
                got = list(pool.map(lambda i: job(i, rnd), range(4)))
            assert got == plain
            for i in range(4):
                a = open(d / f"p{i}" / f"b{i}.fa_in_A{i}.bv", "rb").read()
                b = open(d / f"m{rnd}_{i}" / f"b{i}.fa_in_A{i}.bv", "rb").read()
                assert a == b
        out = subprocess.run([cli.TOOL, "--server-stats"], env=env_c, stdout=subprocess.PIPE, check=True).stdout.decode()
        per_dev = [dict((k, int(v)) for k, v in re.findall(r"(requests|cache hits|loads) (\d+)", ln)) for ln in out.splitlines() if ln.startswith("device server")]
        router = [ln for ln in out.splitlines() if ln.startswith("router:")][0]
        assert len(per_dev) == 2 and all(x["requests"] >= 1 for x in per_dev)                 # both device servers got jobs
        assert sum(x["requests"] for x in per_dev) == 12
        assert sum(x["loads"] for x in per_dev) == 8 and sum(x["cache hits"] for x in per_dev) == 16   # every set parsed ONCE: rounds 2 and 3 went where the files were
        assert int(re.search(r"held files of the job (\d+)", router).group(1)) == 8
    finally:
        subprocess.run([cli.TOOL, "--server-stop"], env=dict(env0, COMMET_SERVER=sock), timeout=120)
        p.wait(timeout=120)
    assert not os.path.exists(sock) and not os.path.exists(sock + ".0")
