"""tools/schedule_sim.py: the N-rank schedule predicted from a one-GPU run's job log — on a log made from a known cost model the fit
must recover the model, one rank must reproduce the log's own total, and more ranks must shorten the run within what the pair cut allows."""
import importlib.util
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("schedule_sim", os.path.join(ROOT, "tools", "schedule_sim.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _log(n_sets, j1=(60.0, 105.0), jx=(2.0, 15.0, 100.0)):
    """the rows matrix.run leaves on ONE rank (pipelined: last reference set first, a target's J3 batch once its last reference is through)"""
    rows = []
    for ref in range(n_sets - 2, -1, -1):
        targets = list(range(ref + 1, n_sets))
        n = len(targets)
        rows.append(["J1", ref, targets, 50.0, 1.0, j1[0] + j1[1] * n])
        rows.append(["J2", ref, targets, 10.0, 1.0, jx[0] + jx[1] * n + jx[2] * math.ceil(n / 4)])
    for i in range(1, n_sets):
        refs = list(range(i))
        n = len(refs)
        rows.append(["J3", i, refs, 10.0, 1.0, jx[0] + jx[1] * n + jx[2] * math.ceil(n / 4)])
    return rows


def test_fit_recovers_the_model_and_one_rank_reproduces_the_log():
    sim = _tool()
    rows = _log(10)
    model = sim.fit(rows)
    assert all(abs(a - b) < 1e-3 for a, b in zip(model["j1"], (60.0, 105.0)))
    assert all(abs(a - b) < 1e-3 for a, b in zip(model["jx"], (2.0, 15.0, 100.0)))
    one = sim.simulate(10, 1, model, parse_s=0.0, import_s=0.0)
    assert abs(one["total_s"] - sum(r[5] for r in rows) * 1e-3) < 1e-3 and one["per_rank"][0]["pairs"] == 45


def test_more_ranks_are_faster_within_the_pair_cut():
    sim = _tool()
    model = sim.fit(_log(10))
    one = sim.simulate(10, 1, model, parse_s=0.3, import_s=0.08)
    prev = one["total_s"]
    for w in (2, 4, 8):
        s = sim.simulate(10, w, model, parse_s=0.3, import_s=0.08)
        assert sum(o["pairs"] for o in s["per_rank"]) == 45 and len(s["per_rank"]) == w
        assert s["total_s"] < prev                                     # every doubling helps
        assert s["total_s"] >= one["per_rank"][0]["jobs_s"] / w * 0.9  # ... and nobody beats the even share
        assert all(o["first_job_at_s"] >= 0.3 for o in s["per_rank"])  # no job before its rank has parsed a set
        prev = s["total_s"]
    assert s["imbalance"] < 1.35
