"""The N>1 host logic on CPU: world_size-2 (and 8) processes share out the pair chains of the
N x N matrix, meet only at barriers, and agree on the MAX elapsed time — over the default
backend (sharding's TCP store: no torch in a rank process) and over gloo."""
import json
import os
import re
import subprocess
import sys

import pytest

from commet_amd import sharding
from conftest import ROOT


def test_job_dag_matches_commet_py():
    # Commet.py schedules N^2-1 invocations: 3 sets -> 8, 5 -> 24, 10 -> 99
    for n, expect in ((2, 3), (3, 8), (5, 24), (10, 99)):
        jobs = sharding.commet_jobs(n)
        assert len(jobs) == expect
        assert [j for j in jobs if j[0] == "J1"][0][2] == list(range(1, n))
    chains = sharding.pair_chains(5)
    assert len(chains) == 10
    for ch in chains:
        (k1, a1, s1, r1), (k2, a2, s2, r2), (k3, a3, s3, r3) = ch
        assert (k1, k2, k3) == ("J1", "J2", "J3")
        ref, i = a1, s1[0]
        assert ref < i and (a2, s2, r2) == (i, [ref], ref) and (a3, s3, r3) == (ref, [i], i)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_assignment_is_a_partition_and_balanced(world):
    chains = sharding.pair_chains(10)
    cost = [1.0 + (c % 4) for c in range(len(chains))]
    parts = [sharding.assign_chains(chains, world, r, cost) for r in range(world)]
    flat = sorted(c for p in parts for c in p)
    assert flat == list(range(len(chains)))
    loads = [sum(cost[c] for c in p) for p in parts]
    assert max(loads) - min(loads) <= max(cost)


@pytest.mark.parametrize("n,world", [(10, 1), (10, 2), (10, 4), (10, 8), (5, 8), (3, 2), (16, 8), (17, 8)])
def test_every_set_has_one_owner_and_left_over_sets_go_to_the_cheapest_ranks(n, world):
    pairs = [(a, b) for a in range(n - 1) for b in range(a + 1, n)]
    runs = sharding.assign_pairs_contiguous([2.0] * len(pairs), world)
    cost = [2.0 * len(r) for r in runs]
    owner = sharding.assign_owners(n, world, cost)
    assert len(owner) == n and all(0 <= o < world for o in owner)
    per = [owner.count(r) for r in range(world)]
    assert max(per) - min(per) <= 1                                   # nobody parses two sets more than anybody else
    extra = [r for r in range(world) if per[r] == max(per)] if max(per) != min(per) else []
    if extra and n >= world:
        assert max(cost[r] for r in extra) <= min(cost[r] for r in range(world) if r not in extra) + 2.0 or len(set(cost)) == 1


@pytest.mark.parametrize("world", [1, 2, 3, 8, 16])
def test_contiguous_runs_partition_the_pairs(world):
    n = 10
    pairs = [(a, b) for a in range(n - 1) for b in range(a + 1, n)]
    size = [1.0 + 0.3 * (s % 4) for s in range(n)]
    cost = [size[a] + size[b] for a, b in pairs]
    runs = sharding.assign_pairs_contiguous(cost, world)
    assert len(runs) == world
    assert [c for r in runs for c in r] == list(range(len(pairs)))             # contiguous, in order, nothing lost
    loads = [sum(cost[c] for c in r) for r in runs]
    assert max(loads) - min(loads) <= 2 * max(cost)
    if world <= 8:
        # few reference sets per rank: J1's index of S_ref is built once per (rank, ref)
        builds = sum(len({pairs[c][0] for c in r}) for r in runs)
        assert builds <= (n - 1) + world


def _matrix_case(tmp_path):
    """5 sets (one of two files, one with a filter that empties a file), k=20; returns (names, files, bvs)"""
    import numpy as np
    import util
    from commet_amd import synth
    k, t, n, L = 20, 2, 1500, 80
    names = ["s0", "s1", "s2", "s3", "s4"]
    files = [["s0.fa"], ["s1a.fa", "s1b.fa"], ["s2.fa"], ["s3.fa"], ["s4.fa"]]
    for s, fl in enumerate(files):
        b, o = synth.synth_set(s, n, L, copy_frac=0.3)
        if len(fl) == 1:
            synth.write_fasta(str(tmp_path / fl[0]), b, o)
        else:
            h = n // 2
            synth.write_fasta(str(tmp_path / fl[0]), b[: h * L], o[: h + 1])
            synth.write_fasta(str(tmp_path / fl[1]), b[h * L:], o[h:] - o[h])
    rng = np.random.default_rng(1)
    bvs = []
    for s, fl in enumerate(files):
        row = []
        for j, f in enumerate(fl):
            cnt = len(util.parse_fasta(str(tmp_path / f)))
            sel = rng.random(cnt) < 0.9
            if (s, j) == (1, 1):
                sel[:] = False                                   # a file with no selected read
            util.write_bv(str(tmp_path / (f + ".bv")), "filter of " + f, sel)
            row.append(f + ".bv")
        bvs.append(row)
    (tmp_path / "sets.txt").write_text("".join(
        f"{names[s]}: " + "; ".join(f"{f},{b}" for f, b in zip(files[s], bvs[s])) + "\n" for s in range(len(names))))
    return k, t, names, files, bvs


def _launch(world, args, cwd, timeout=600, extra_env=None, launcher="torchrun"):
    """launcher: "torchrun" (the driver's form: python -m torch.distributed.run ...) or "spawn" (sharding.spawn_ranks, what
    `bench.py --gpus N` and `python -m commet_amd.matrix --gpus N` do themselves: plain child processes)"""
    env = dict(os.environ, OMP_NUM_THREADS="1", COMMET_SCRATCH=str(cwd), COMMET_DIST_TIMEOUT_S="120", **(extra_env or {}))
    if launcher == "spawn":
        code = (f"import sys; sys.path.insert(0, {ROOT!r}); from commet_amd import sharding; "
                f"sys.exit(sharding.spawn_ranks({world}, [sys.executable] + {args!r}))")
        cmd = [sys.executable, "-c", code]
        env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    else:
        port = 29500 + (os.getpid() * 7 + world) % 2000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + args
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, cwd=cwd, timeout=timeout)


@pytest.mark.parametrize("world,handover,backend,launcher", [
    (2, "default", "tcp", "torchrun"), (2, "default", "tcp", "spawn"), (8, "default", "tcp", "spawn"), (2, "image", "tcp", "spawn"),
    (2, "canary-fail", "tcp", "spawn"), (2, "canary-hang", "tcp", "torchrun"), (2, "canary-ok", "tcp", "spawn"),
    (2, "default", "gloo", "torchrun"), (2, "image", "gloo", "torchrun"), (8, "default", "gloo", "torchrun")])
def test_matrix_driver_host_logic_over_ranks(tmp_path, world, handover, backend, launcher):
    """The N x N driver over `world` processes (the CPU checker stands in for the GPU engine): every set is parsed by exactly
    one rank, the others take it from its owner — device to device through an exported descriptor (the default: the engine's
    export / import, probed between the ranks first) or as a packed image in the scratch directory (COMMET_MATRIX_IPC=0;
    and, rank by rank, when the CANARY — the fresh child process that imports the first real set before any rank does —
    fails or hangs: the importers then ask the owners for images); outputs equal Commet.py's job sequence run in one process.
    Ranks over the default backend (TCP store of rank 0; no torch in the rank processes) and over gloo, started by
    torch.distributed.run (the driver's form) and by sharding.spawn_ranks (plain child processes)."""
    import json
    import oracle_binding as ob
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import commet_jobs
    k, t, names, files, bvs = _matrix_case(tmp_path)
    env = {"COMMET_RANKS_BACKEND": backend}
    if handover == "image":
        env["COMMET_MATRIX_IPC"] = "0"
    if handover.startswith("canary-"):
        env.update(COMMET_TEST_CANARY=handover[7:], COMMET_IPC_CANARY_S="3")
    p = _launch(world, [os.path.join(ROOT, "tests", "matrix_worker.py"), "sets.txt", "out/", str(k), str(t)], str(tmp_path),
                extra_env=env, launcher=launcher)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    res = json.load(open(tmp_path / "out" / "result.json"))
    assert res["world"] == world and len(res["per_rank"]) == world
    # what every rank ended up with: a rank that takes no set from another never meets the canary's verdict
    takers = [r for r in res["per_rank"] if r["sets_loaded"] > 0]
    assert takers
    assert {r["handover"] for r in takers} == ({"image"} if handover in ("image", "canary-fail", "canary-hang") else {"ipc"})
    if handover.startswith("canary-"):
        verdicts = {r.get("ipc_canary") for r in takers}
        assert len(verdicts) == 1 and next(iter(verdicts)).startswith({"ok": "passed", "fail": "failed (exit code 1)", "hang": "failed (no answer"}[handover[7:]])
    assert all(r["backend"] == backend and r["torch_loaded"] == (backend == "gloo") for r in res["per_rank"])
    assert sum(r["sets_parsed"] for r in res["per_rank"]) == len(names)          # one parse per set on the node
    assert sum(r["pairs"] for r in res["per_rank"]) == 10
    assert sum(r["j1_builds"] for r in res["per_rank"]) <= 4 + 2 * world          # (a rank may split its first J1 by the sets that have arrived)
    # the static cut: every rank's predicted share of the pairs' cost is within one pair of the mean, and they add up
    shares = [r["predicted_share"] for r in res["per_rank"]]
    sizes = [sum(os.path.getsize(tmp_path / f) for f in fl) for fl in files]
    one_pair = 2.0 * max(sizes) / sum(sizes[a] + sizes[b] for a in range(5) for b in range(a + 1, 5))
    assert abs(sum(shares) - 1.0) < 1e-2 and max(shares) - min(shares) <= 2 * one_pair + 1e-3
    assert res["load_overlaps_jobs"] is True                                     # own sets parsed beside the jobs, no barrier
    assert not [f for f in os.listdir(tmp_path) if f.startswith("commet_pk_")]   # the scratch images are gone

    def cfg(si, restrict_to=None):
        parts = []
        for f, b in zip(files[si], bvs[si]):
            parts.append(f + "," + (b if restrict_to is None else f"orc/{f}_in_{names[restrict_to]}.bv"))
        return names[si] + ":" + ";".join(parts)

    os.makedirs(tmp_path / "orc")
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        for kind, idx, searches, restr in commet_jobs(names):
            open("i.txt", "w").write(cfg(idx, restr) + "\n")
            open("s.txt", "w").write("".join(cfg(s) + "\n" for s in searches))
            rc, *_ = ob.index_and_search("i.txt", "s.txt", "orc", "orc", k, t)
            assert rc == 0
    finally:
        os.chdir(cwd)
    checked = 0
    for f in sorted(os.listdir(tmp_path / "orc")):
        if f.endswith(".bv"):
            assert open(tmp_path / "out" / f, "rb").read().split(b"\n", 1)[1] == open(tmp_path / "orc" / f, "rb").read().split(b"\n", 1)[1], f
            checked += 1
    assert checked == 6 * 4          # 6 files, each searched in the 4 other sets


@pytest.mark.parametrize("launcher,backend", [("torchrun", "tcp"), ("spawn", "tcp"), ("torchrun", "gloo")])
def test_matrix_driver_failing_rank_ends_the_group(tmp_path, launcher, backend):
    """rank 1 raises while parsing: the launcher must come back non-zero within seconds, not after a barrier timeout"""
    import time
    k, t, names, files, bvs = _matrix_case(tmp_path)
    t0 = time.time()
    p = _launch(2, [os.path.join(ROOT, "tests", "matrix_worker.py"), "sets.txt", "out/", str(k), str(t), "1"], str(tmp_path), timeout=300,
                launcher=launcher, extra_env={"COMMET_RANKS_BACKEND": backend})
    assert p.returncode != 0
    assert b"injected failure while parsing" in p.stdout
    assert time.time() - t0 < 90


def test_a_rank_that_raises_tells_the_others(tmp_path):
    """rank 1 raises inside matrix.run and its caller CATCHES (as bench.py's matrix leg does): matrix.run has told the store, so
    rank 0 — three sets into its own work, or waiting in a gather — gets an error naming rank 1 within seconds, not a timeout"""
    import time
    k, t, names, files, bvs = _matrix_case(tmp_path)
    t0 = time.time()
    p = _launch(2, [os.path.join(ROOT, "tests", "matrix_worker.py"), "sets.txt", "out/", str(k), str(t), "1"], str(tmp_path), timeout=300,
                launcher="spawn", extra_env={"COMMET_TEST_SOFT_FAIL": "1"})
    out = p.stdout.decode()
    assert p.returncode == 5, out[-2000:]
    assert "injected failure while parsing" in out and "rank 1 (gave up: RuntimeError: injected failure while parsing)" in out
    assert time.time() - t0 < 60


def test_an_import_that_never_returns_ends_the_rank_after_its_last_words(tmp_path):
    """rank 0's import of another rank's set hangs (a HIP call cannot be cancelled): after COMMET_IPC_IMPORT_LIMIT_S the rank's
    watchdog calls the caller's fatal_hook (bench.py prints its headline there) and ends the process with code 4; the launcher
    ends the job"""
    import time
    k, t, names, files, bvs = _matrix_case(tmp_path)
    t0 = time.time()
    p = _launch(2, [os.path.join(ROOT, "tests", "matrix_worker.py"), "sets.txt", "out/", str(k), str(t)], str(tmp_path), timeout=300,
                launcher="spawn", extra_env={"COMMET_TEST_IMPORT_HANG": "0", "COMMET_IPC_IMPORT_LIMIT_S": "2"})
    out = p.stdout.decode()
    assert p.returncode == 4, out[-2000:]
    assert "commet_readset_import did not return within 2 s" in out
    assert "did not return" in open(tmp_path / "out" / "last_words_rank0.txt").read()
    assert time.time() - t0 < 60


def test_store_tells_the_ranks_when_one_of_them_is_gone(tmp_path):
    """no launcher to end the group: three ranks started by hand, rank 2 leaves without a word while the others wait at a
    barrier — the store answers their wait with an error at once (not after COMMET_DIST_TIMEOUT_S)"""
    import time
    code = (f"import os, sys; sys.path.insert(0, {ROOT!r}); from commet_amd import sharding\n"
            "me = int(os.environ['RANK'])\n"
            "try:\n    r = sharding.Ranks()\n    if me == 2: os._exit(0)\n    r.barrier(); r.barrier()\n"
            "except RuntimeError as ex:\n    print('RANK', me, ex); sys.exit(7)\n")
    env = dict(os.environ, WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(31000 + os.getpid() % 2000), COMMET_DIST_TIMEOUT_S="120")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, text=True)
             for r in range(3)]
    outs = [p.communicate(timeout=100)[0] for p in procs]
    assert [p.returncode for p in procs] == [7, 7, 0], outs
    assert "rank 2 left the job" in outs[0]
    assert "rank 2 left the job" in outs[1] or "lost the store" in outs[1]      # (rank 0, and its store, may be gone first)
    assert time.time() - t0 < 60


@pytest.mark.parametrize("launcher", ["torchrun", "spawn"])
def test_two_ranks_default_backend_has_no_torch(tmp_path, launcher):
    """the default backend of sharding.Ranks: the same protocol as test_two_ranks_gloo, and `torch` is not in a rank's sys.modules
    (a rank process that holds torch's ROCm runtime beside the system's was what hung commet_readset_import in round 3)"""
    p = _launch(2, [os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), "6"], str(tmp_path), timeout=300, launcher=launcher)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    r0 = json.load(open(tmp_path / "rank0.json"))
    r1 = json.load(open(tmp_path / "rank1.json"))
    assert r0["backend"] == r1["backend"] == "tcp" and r0["torch_loaded"] is False and r1["torch_loaded"] is False
    assert sorted(r0["mine"] + r1["mine"]) == list(range(15))
    assert r0["everyone"] == r1["everyone"] == [r0["mine"], r1["mine"]]
    assert r0["total_jobs"] == r1["total_jobs"] == 15 * 3 * 2
    assert r0["elapsed"] == r1["elapsed"] >= 0.19
    assert r0["first"] == r1["first"] == "from rank 1"


def test_two_ranks_gloo(tmp_path):
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), "6", "gloo"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=300)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    r0 = json.load(open(tmp_path / "rank0.json"))
    r1 = json.load(open(tmp_path / "rank1.json"))
    assert r0["world"] == r1["world"] == 2
    assert sorted(r0["mine"] + r1["mine"]) == list(range(15))           # 6 sets -> 15 pair chains
    assert r0["everyone"] == r1["everyone"] == [r0["mine"], r1["mine"]]
    assert r0["total_jobs"] == r1["total_jobs"] == 15 * 3 * 2            # 3 jobs per chain, 2 steps
    assert r0["elapsed"] == r1["elapsed"] >= 0.19                        # MAX over ranks: rank 1 sleeps 2 x 0.1 s
    assert r0["backend"] == "gloo" and r0["torch_loaded"] is True and r0["first"] == r1["first"] == "from rank 1"


def test_bench_json_contract_fields():
    """bench.py's JSON keys (the driver parses them): checked statically, the values need a GPU."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "workload"):
        assert re.search(r'"%s"' % key, src), key


def _bv_bodies(d):
    return {f: open(os.path.join(d, f), "rb").read().split(b"\n", 1)[1] for f in sorted(os.listdir(d)) if f.endswith(".bv")}


def test_matrix_driver_single_rank_loads_sets_beside_the_jobs(tmp_path, monkeypatch):
    """One rank: the sets are loaded (last set first) by a second thread while the jobs run.  Same .bv files and matrices as
    with everything loaded up front (COMMET_MATRIX_PIPELINE=0), which the multi-rank test pins to Commet.py's job sequence."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from commet_amd import matrix
    from oracle_engine import OracleEngine
    k, t, names, files, bvs = _matrix_case(tmp_path)
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    OracleEngine.fail_on_rank = None
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("COMMET_MATRIX_PIPELINE", mode)
        res[mode] = matrix.run("sets.txt", f"out{mode}/", k=k, t=t, verbose=False, engine_factory=OracleEngine)
    assert res["1"]["load_overlaps_jobs"] is True and res["0"]["load_overlaps_jobs"] is False
    assert res["1"]["per_rank"][0]["sets_parsed"] == len(names) == res["0"]["per_rank"][0]["sets_parsed"]
    a, b = _bv_bodies(tmp_path / "out1"), _bv_bodies(tmp_path / "out0")
    assert a == b and len(a) == 6 * 4
    for f in ("matrix_plain.csv", "matrix_percentage.csv", "matrix_normalized.csv"):
        assert open(tmp_path / "out1" / f).read() == open(tmp_path / "out0" / f).read()
    assert res["1"]["reads_searched"] == res["0"]["reads_searched"]


def test_matrix_driver_single_rank_loader_failure_reaches_the_caller(tmp_path, monkeypatch):
    """the loading thread raises while parsing: the job thread, which waits for that set, must raise it (no hang)"""
    import time
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from commet_amd import matrix
    from oracle_engine import OracleEngine
    k, t, names, files, bvs = _matrix_case(tmp_path)
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setenv("COMMET_MATRIX_PIPELINE", "1")
    OracleEngine.fail_on_rank = 0
    t0 = time.time()
    try:
        with pytest.raises(Exception, match="injected failure while parsing"):
            matrix.run("sets.txt", "out/", k=k, t=t, verbose=False, engine_factory=OracleEngine)
    finally:
        OracleEngine.fail_on_rank = None
    assert time.time() - t0 < 60


@pytest.mark.parametrize("world", [1, 3])
def test_default_filter_vectors_equal_the_tools(tmp_path, world):
    """Commet.py's default filter options remove no read: the driver writes the filter .bv files from the parser's record
    counts (matrix.default_filter_bv) instead of running filter_reads over every file again.  Same bytes as the tool's
    (COMMET_MATRIX_FILTER_TOOL=1 runs it), for FASTA / FASTQ / gzip files, read counts that are and are not multiples of 8,
    sets of several files; and the same results behind them.  One rank and three."""
    import filecmp
    import numpy as np
    import util
    rng = np.random.default_rng(3)
    reads = [util.random_reads(rng, n, 30, 120, n_rate=0.02) for n in (1501, 800, 64, 7)]
    for name, r, fmt in (("a.fa", reads[0], "fa"), ("b.fq", reads[1], "fq"), ("c.fa.gz", reads[2], "fa.gz"), ("d.fa", reads[3], "fa")):
        util.write_reads(str(tmp_path / name), r, fmt, rng=rng)
    (tmp_path / "sets.txt").write_text("A: a.fa; d.fa\nB: b.fq\nC: c.fa.gz\n")
    for out, env in (("out_s/", {}), ("out_t/", {"COMMET_MATRIX_FILTER_TOOL": "1"})):
        p = _launch(world, [os.path.join(ROOT, "tests", "matrix_worker.py"), "sets.txt", out, "20", "2"], str(tmp_path), launcher="spawn", extra_env=env) \
            if world > 1 else subprocess.run([sys.executable, os.path.join(ROOT, "tests", "matrix_worker.py"), "sets.txt", out, "20", "2"], cwd=str(tmp_path),
                                             env=dict({k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK")}, **env),
                                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert p.returncode == 0, p.stdout.decode()[-2000:]
    names = sorted(f for f in os.listdir(tmp_path / "out_t") if f.endswith((".bv", ".csv")))
    assert [f for f in names if "_in_" not in f and f.endswith(".bv")] == ["a.fa.bv", "b.fq.bv", "c.fa.gz.bv", "d.fa.bv"]
    assert not [f for f in names if not filecmp.cmp(tmp_path / "out_t" / f, tmp_path / "out_s" / f, shallow=False)]
    assert sorted(f for f in os.listdir(tmp_path / "out_s") if f.endswith((".bv", ".csv"))) == names


def test_bench_matrix_leg_sizes_itself_to_the_host(tmp_path):
    """the default matrix leg (10 x 50 M reads as FASTA in the scratch root when there are several GPUs) must not drive the host out of
    memory: what it needs is priced and compared with what the host has free"""
    sys.path.insert(0, ROOT)
    import bench
    free = bench.host_memory_free(str(tmp_path))
    assert free is not None and 0 < free <= os.statvfs(str(tmp_path)).f_bavail * os.statvfs(str(tmp_path)).f_frsize
    big, small = bench.matrix_memory_needed(10, 50_000_000, 100, 8), bench.matrix_memory_needed(10, 10_000_000, 100, 8)
    assert big > 10 * 50_000_000 * 100 and small < big / 4            # at least the bases themselves; scales with the reads
    assert small > 10 * 10_000_000 * 100


def test_bench_gpus_n_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset) must start the two ranks itself — as a
    child process, the parent never touching the GPU — and print ONE line with n_gpus = 2 (the launch path only: the
    ranks meet at the barriers of the timed region and leave; the GPU work behind it is covered by the -m gpu tests)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0", "--rendezvous-only"],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.split("\n") if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["self_launched"] is True
    assert out["ranks_backend"] == "tcp" and out["torch_in_ranks"] is False   # plain child processes, no torch in any of them
    assert out["devices"] == [0, 1]                       # one rank per GPU: rank r works on device LOCAL_RANK = r
    assert out["ms_per_step"] >= 20.0                     # MAX over the ranks (rank 1 sleeps 20 ms per step)


def test_bench_under_a_launcher_does_not_launch_again(tmp_path):
    """the driver's own form: torch.distributed.run ... bench.py --gpus 2 (WORLD_SIZE set): the ranks run as they are"""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29671", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--rendezvous-only"],
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.split("\n") if ln.startswith("{")]
    out = json.loads(lines[0])
    assert len(lines) == 1 and out["n_gpus"] == 2 and out["self_launched"] is False
    assert out["ranks_backend"] == "tcp" and out["torch_in_ranks"] is False   # the launcher holds torch, the ranks do not


def test_device_of_a_rank_is_local_rank_modulo_what_the_process_sees(monkeypatch):
    """a launcher may hand every rank ONE visible device (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank): device LOCAL_RANK
    does not exist there — every rank must end on its device 0 (bench.py and matrix.HipEngine both ask sharding.pick_device)"""
    from commet_amd import sharding
    monkeypatch.delenv("COMMET_FORCE_DEVICE", raising=False)
    assert [sharding.pick_device(r, 8) for r in range(8)] == list(range(8))      # a node's eight devices, all visible
    assert [sharding.pick_device(r, 1) for r in range(8)] == [0] * 8             # one visible device per rank
    assert [sharding.pick_device(r, 4) for r in range(8)] == [0, 1, 2, 3, 0, 1, 2, 3]
    assert sharding.pick_device(5, None) == 5 and sharding.pick_device(5, 0) == 5  # count unknown: LOCAL_RANK as it is
    monkeypatch.setenv("COMMET_FORCE_DEVICE", "0")                                 # the rehearsal knob wins
    assert [sharding.pick_device(r, 8) for r in range(4)] == [0] * 4
    src = open(os.path.join(ROOT, "commet_amd", "matrix.py")).read() + open(os.path.join(ROOT, "bench.py")).read()
    assert src.count("sharding.pick_device(") >= 3 and 'os.environ.get("COMMET_FORCE_DEVICE", local_rank)' not in src


def test_bench_ranks_with_one_visible_device_each(tmp_path):
    """the launch path under such a launcher (BENCH_FAKE_DEVICE_COUNT stands in for commet_device_count(): no HIP call on this path)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "COMMET_FORCE_DEVICE")}
    env["BENCH_FAKE_DEVICE_COUNT"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0", "--rendezvous-only"],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.split("\n") if ln.startswith("{")][0])
    assert out["n_gpus"] == 3 and out["devices"] == [0, 0, 0]


def test_what_the_ranks_send_each_other_is_json_not_pickle():
    """values of the TCP store come off a socket: they are decoded as tagged JSON (bytes, tuples and dicts with non-string keys
    survive), never unpickled"""
    import numpy as np
    from commet_amd import sharding
    src = open(os.path.join(ROOT, "commet_amd", "sharding.py")).read()
    assert "pickle" not in src.replace("unpickle", "")
    cases = [None, True, 7, -1.5, "scratch/dir", b"\x00\xffIPC" * 30, (1, 2, ("a", b"z")), [1, [2, (3,)]],
             {(0, 1): 5, (8, 9): 0}, {0: 11, 1: 12}, {"jobs_s": 1.25, "handover": "ipc", "pairs": [(0, 1), (0, 2)]},
             ({(0, 1): 3}, {"rank": 1, "jobs_s": 0.5}, {0: 100, 1: 200}), {"__t__": 1, "k": 2}, float("inf")]
    for obj in cases:
        assert sharding._decode_value(sharding._encode_value(obj)) == obj, obj
    assert sharding._decode_value(sharding._encode_value([np.int64(5), np.float64(0.5), np.arange(3)])) == [5, 0.5, [0, 1, 2]]
    with pytest.raises(TypeError):
        sharding._encode_value(object())
    with pytest.raises(ValueError):
        sharding._decode_value(b"\x80\x04\x95\x05\x00\x00\x00\x00\x00\x00\x00K\x01.")      # a pickle: not JSON, refused


def test_bench_line_says_so_at_top_level_when_the_matrix_leg_failed():
    """a failed matrix leg must show where a reader of the per-N lines looks: top-level "matrix": {"error": ...} (the linear
    weak-scaling `value` beside no matrix object would read as a clean run); the rank the failure began in leaves non-zero"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'out["matrix"] = {"error": matrix_detail["error"], "world": world}' in src
    assert "if rank != 0 and failed_here:" in src and "os._exit(3)" in src
