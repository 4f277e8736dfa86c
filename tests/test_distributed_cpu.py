"""The N>1 host logic on CPU: world_size-2 gloo processes share out the pair chains of the
N x N matrix, meet only at barriers, and agree on the MAX elapsed time."""
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


def test_two_ranks_gloo(tmp_path):
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), "6"]
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


def test_bench_json_contract_fields():
    """bench.py's JSON keys (the driver parses them): checked statically, the values need a GPU."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "workload"):
        assert re.search(r'"%s"' % key, src), key
