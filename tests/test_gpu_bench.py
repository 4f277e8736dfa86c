"""bench.py end to end on the GPU at a small size: the JSON contract of the line (the driver parses it), the roofline and
cpu_baseline objects, the same-job CPU leg (--cpu-full: the reference tool on the whole job, `.bv` byte-compared with the GPU's
tags) and the matrix leg through the driver — on one rank and on two ranks that bench.py starts itself (plain child processes,
no torch; both on GPU 0 here)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _bench(args, tmp_path, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(COMMET_SCRATCH=str(tmp_path), **(env or {}))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=str(tmp_path), env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.split("\n") if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_small_job_with_the_same_job_cpu_leg(tmp_path):
    out = _bench(["--reads", "300000", "-k", "25", "--steps", "2", "--warmup", "1", "--cpu-sample", "100000", "--cpu-full",
                  "--matrix-sets", "3", "--matrix-reads", "200000"], tmp_path)
    for key in CONTRACT:
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["unit"] == "reads/s" and out["value"] > 0 and out["dtype"] == "u32"
    assert out["config"]["workload"].startswith("2 synthetic sets x 300000 x 100 bp reads, k=25 t=2")
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["kernel"] in r["kernels"] and r["compulsory_bytes"] > 0
    cb = out["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0
    fj = cb["full_job"]
    assert fj["bv_bytes_equal_gpu"] is True and fj["log_numbers_equal_gpu"] is True and fj["cores"] == 1
    m = out["matrix"]
    assert m["world"] == 1 and m["reads_per_s_incl_load_and_filter"] > 0 and "3 x 3 matrix" in m["workload"]
    assert out["detail"]["first_job_ms"] > 0 and out["detail"]["matrix"]["per_rank"][0]["torch_loaded"] is False


def test_bench_starts_two_ranks_itself_and_hands_sets_over_device_to_device(tmp_path):
    out = _bench(["--gpus", "2", "--reads", "200000", "-k", "25", "--steps", "2", "--warmup", "1", "--matrix-sets", "4", "--matrix-reads", "150000"],
                 tmp_path, env={"COMMET_FORCE_DEVICE": "0"})
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["jobs"] == 2 and "cpu_baseline" not in out
    m, per_rank = out["matrix"], out["detail"]["matrix"]["per_rank"]
    assert m["world"] == 2 and m["handover"] == ["ipc"] and len(per_rank) == 2
    assert all(p["backend"] == "tcp" and p["torch_loaded"] is False for p in per_rank)
    assert sum(p["sets_parsed"] for p in per_rank) == 4 and sum(p["pairs"] for p in per_rank) == 6
    assert all(p.get("ipc_canary", "passed") == "passed" for p in per_rank if p["sets_loaded"])
