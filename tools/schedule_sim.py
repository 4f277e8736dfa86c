#!/usr/bin/env python3
"""schedule_sim.py — what the N x N driver's static schedule predicts for N = 2 / 4 / 8 ranks, from a ONE-GPU run's record.

No multi-GPU node is available to the builder; what can be checked without one is the schedule: commet_amd.matrix cuts the
(ref, i) pairs into contiguous runs (sharding.assign_pairs_contiguous) and gives every set one parser (sharding.assign_owners);
a rank then parses its sets, takes the others device to device as they appear and runs, per reference set, J1 (index of S_ref
once for its targets), the J2 jobs of the targets in shared passes, and the J3 jobs of a target once its last reference set is
through (Commet.py:186-240, 570-574 is the job DAG being sharded).  This tool replays that logic — the same functions for the cut
and the owner map, the same order of work per rank — on costs MEASURED on one GPU: the `job_log` / `parse_log` rows that
matrix.run leaves in its per-rank profile (bench.py: detail.matrix.per_rank[0]).

Cost model (fitted to the log's rows, least squares):
    J1 call of a reference set against n targets     = a1 + b1 * n                       (index build of S_ref + one pass per target)
    J2 / J3 call of n jobs that share one search set = a2 + b2 * n + c2 * ceil(n / 4)    (index builds per job + shared passes of 4 jobs)
    parse of a set                                   = the mean of parse_log
    a set taken from another rank                    = --import-s (default 0.08 s: 2.4 GB over xGMI + the IPC open)
Everything is scaled by reads where sets differ in size (they do not in the bench's synthetic matrix).

usage: python tools/schedule_sim.py BENCH.json [--leg matrix|matrix_configs2|ragged] [--world 2 4 8] [--import-s 0.08] [--json OUT]
"""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from commet_amd import sharding  # noqa: E402


def lstsq(rows, ys):
    """least squares without numpy's help for tiny systems (normal equations, Gauss-Jordan)"""
    m = len(rows[0])
    A = [[sum(r[i] * r[j] for r in rows) for j in range(m)] for i in range(m)]
    b = [sum(r[i] * y for r, y in zip(rows, ys)) for i in range(m)]
    for i in range(m):
        A[i][i] += 1e-9
    for c in range(m):
        p = max(range(c, m), key=lambda r: abs(A[r][c]))
        A[c], A[p], b[c], b[p] = A[p], A[c], b[p], b[c]
        for r in range(m):
            if r != c and A[c][c]:
                f = A[r][c] / A[c][c]
                A[r] = [x - f * y for x, y in zip(A[r], A[c])]
                b[r] -= f * b[c]
    return [b[i] / A[i][i] if A[i][i] else 0.0 for i in range(m)]


def fit(job_log):
    j1 = [(len(r[2]), r[5]) for r in job_log if r[0] == "J1"]
    jx = [(len(r[2]), r[5]) for r in job_log if r[0] in ("J2", "J3")]
    a1, b1 = lstsq([[1.0, n] for n, _ in j1], [ms for _, ms in j1]) if len(j1) >= 2 else (0.0, j1[0][1] / max(1, j1[0][0]))
    if len(jx) >= 3:
        a2, b2, c2 = lstsq([[1.0, n, math.ceil(n / 4)] for n, _ in jx], [ms for _, ms in jx])
    else:
        a2, c2, b2 = 0.0, 0.0, sum(ms for _, ms in jx) / max(1, sum(n for n, _ in jx))
    res1 = max((abs(a1 + b1 * n - ms) / ms for n, ms in j1), default=0.0)
    res2 = max((abs(a2 + b2 * n + c2 * math.ceil(n / 4) - ms) / ms for n, ms in jx), default=0.0)
    return dict(j1=(a1, b1), jx=(a2, b2, c2), worst_rel_residual=(round(res1, 3), round(res2, 3)))


def simulate(n_sets, world, model, parse_s, import_s, sizes=None, canary_s=0.0, ready_first=True, interleave=False, blocking_first=False):
    """-> per-rank dicts + total seconds.  Mirrors matrix.run: pair cut, owner map, loader thread, job thread.
    canary_s: no rank imports a set before the canary process has its verdict.  ready_first: the job thread starts with a reference
    set that is resident together with one of its targets (round 6; False = the fixed order of rounds 4-5, for comparison).
    interleave: the loader takes foreign sets that have appeared between two parses of its own — simulated and NOT adopted: it delays
    the second set of the two ranks that parse two, which other ranks wait for (configs[3] at eight ranks: 2.77 against 2.53 s)."""
    sizes = sizes or [1.0] * n_sets
    pairs = [(ref, i) for ref in range(n_sets - 1) for i in range(ref + 1, n_sets)]
    pair_cost = [sizes[a] + sizes[b] for a, b in pairs]
    runs = sharding.assign_pairs_contiguous(pair_cost, world)
    owner = sharding.assign_owners(n_sets, world, [sum(pair_cost[c] for c in runs[r]) for r in range(world)])
    mine = {r: [pairs[c] for c in runs[r]] for r in range(world)}
    a1, b1 = model["j1"]
    a2, b2, c2 = model["jx"]

    def j1_s(n):
        return (a1 + b1 * n) * 1e-3

    def jx_s(n):
        return (a2 + b2 * n + c2 * math.ceil(n / 4)) * 1e-3 if n else 0.0

    # ---- loader threads: when is set s resident on rank r? -------------------------------------------------------
    needed = {r: sorted({s for p in mine[r] for s in p}) for r in range(world)}
    needed_by_others = {r: {s for q in range(world) if q != r for p in mine[q] for s in p} for r in range(world)}
    own_order, order = {}, {}
    for r in range(world):
        owned = [s for s in range(n_sets) if owner[s] == r]
        wanted_by = {s: sum(1 for q in range(world) if any(s in p for p in mine[q])) for s in owned}
        if world == 1:
            own_order[r] = list(range(n_sets - 1, -1, -1))
        else:
            # blocking_first (simulated, NOT adopted: it helps or hurts by 3-5 % depending on the fitted costs): first the sets some rank can
            # do NOTHING without (they are in every pair of its run), then the most wanted
            blocks = {s: sum(1 for q in range(world) if mine[q] and all(s in p for p in mine[q])) for s in owned}
            own_order[r] = sorted((s for s in owned if s in needed[r] or s in needed_by_others[r]),
                                  key=lambda s: ((-blocks[s], -wanted_by[s], s) if blocking_first else (-wanted_by[s], s)))
        refs = sorted({p[0] for p in mine[r]}, reverse=True)
        order[r] = []
        for ref in refs:
            for s in [ref] + [i for (rr, i) in mine[r] if rr == ref]:
                if s not in order[r]:
                    order[r].append(s)
    parsed_at = {}
    for r in range(world):                                 # first guess: every rank parses its sets back to back
        t = 0.0
        for s in own_order[r]:
            t += parse_s * sizes[s]
            parsed_at[s] = t
    ready = {}
    for _ in range(8):                                     # (an import between two parses delays the second: to a fixed point)
        new_parsed = {}
        for r in range(world):
            t, rd = 0.0, {}
            todo = [s for s in order[r] if owner[s] != r]
            for s_own in own_order[r]:
                t += parse_s * sizes[s_own]
                new_parsed[s_own] = rd[s_own] = t
                if interleave and world > 1:
                    for s in list(todo):
                        if max(parsed_at.get(s, 1e9), canary_s) <= t:
                            t += import_s * sizes[s]
                            rd[s] = t
                            todo.remove(s)
            for s in todo:
                t = max(t, parsed_at.get(s, 0.0), canary_s) + import_s * sizes[s]
                rd[s] = t
            ready[r] = rd
        if all(abs(new_parsed[s] - parsed_at[s]) < 1e-9 for s in new_parsed):
            break
        parsed_at = new_parsed
    # ---- job threads --------------------------------------------------------------------------------------------------
    out = []
    for r in range(world):
        refs = sorted({p[0] for p in mine[r]}, reverse=True)
        t, wait, busy, j1_builds = 0.0, 0.0, 0.0, 0
        refs_left = {}
        for (_, i) in mine[r]:
            refs_left[i] = refs_left.get(i, 0) + 1
        first_job = None
        left = {ref: [i for (rr, i) in mine[r] if rr == ref] for ref in refs}      # reference set -> targets not yet through J1 / J2
        while left:
            def startable(ref):
                return ready[r][ref] <= t and any(ready[r][i] <= t for i in left[ref])
            cand = [ref for ref in refs if ref in left and startable(ref)]
            if not cand:
                nxt = min(max(ready[r][ref], min(ready[r][i] for i in left[ref])) for ref in left)
                if not ready_first:                        # the fixed order: wait for the first reference set of the list, whatever else is there
                    ref0 = next(ref for ref in refs if ref in left)
                    nxt = max(ready[r][ref0], min(ready[r][i] for i in left[ref0]))
                wait += max(0.0, nxt - t)
                t = max(t, nxt)
                continue
            ref = cand[0] if ready_first else next(x for x in refs if x in left)
            if not ready_first and not startable(ref):
                nxt = max(ready[r][ref], min(ready[r][i] for i in left[ref]))
                wait += max(0.0, nxt - t)
                t = max(t, nxt)
                continue
            targets = [i for i in left[ref] if ready[r][i] <= t]
            if first_job is None:
                first_job = t
            d = j1_s(len(targets)) + jx_s(len(targets))
            t += d
            busy += d
            j1_builds += 1
            left[ref] = [i for i in left[ref] if i not in targets]
            if not ready_first:                            # (the fixed order stays on the reference set until all its targets are through)
                while left[ref]:
                    nxt = min(ready[r][i] for i in left[ref])
                    if nxt > t:
                        wait += nxt - t
                        t = nxt
                    tg = [i for i in left[ref] if ready[r][i] <= t]
                    d = j1_s(len(tg)) + jx_s(len(tg))
                    t += d
                    busy += d
                    j1_builds += 1
                    left[ref] = [i for i in left[ref] if i not in tg]
                    targets += tg
            for i in targets:
                refs_left[i] -= 1
            for i in sorted(targets):
                if refs_left[i] == 0:
                    d = jx_s(sum(1 for (_, tt) in mine[r] if tt == i))
                    t += d
                    busy += d
            if not left[ref]:
                del left[ref]
        out.append(dict(rank=r, pairs=len(mine[r]), refs=len(refs), j1_builds=j1_builds, sets_parsed=len(own_order[r]),
                        sets_imported=len([s for s in ready[r] if owner[s] != r]), first_job_at_s=round(first_job or 0.0, 3),
                        set_wait_s=round(wait, 3), jobs_s=round(busy, 3), end_s=round(t, 3)))
    total = max(o["end_s"] for o in out)
    mean_busy = sum(o["jobs_s"] for o in out) / world
    return dict(world=world, total_s=round(total, 3), imbalance=round(max(o["jobs_s"] for o in out) / mean_busy, 4) if mean_busy else None,
                idle_s=[round(total - o["jobs_s"], 3) for o in out], per_rank=out)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("bench_json")
    ap.add_argument("--leg", default="matrix", help="detail.<leg> of the bench line (matrix = configs[3], matrix_configs2, ragged = detail.ragged.matrix)")
    ap.add_argument("--world", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--import-s", type=float, default=0.08)
    ap.add_argument("--canary-s", type=float, default=0.3, help="no import before the canary process's verdict (HIP start-up of a fresh process + the first real set)")
    ap.add_argument("--old-order", action="store_true", help="the job order of rounds 4-5 (reference sets in the fixed order of the rank's list)")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    line = json.load(open(a.bench_json))
    det = line.get("detail", line)
    leg = det["ragged"]["matrix"] if a.leg == "ragged" else det[a.leg]
    prof = leg["per_rank"][0]
    if "job_log" not in prof:
        raise SystemExit("the record has no job_log (taken before round 6?)")
    n_sets = 1 + max(max([r[1]] + r[2]) for r in prof["job_log"])
    model = fit(prof["job_log"])
    parse_s = sum(x[1] for x in prof.get("parse_log", [])) / max(1, len(prof.get("parse_log", []))) or prof.get("parse_s", 0.0) / max(1, prof.get("sets_parsed", 1))
    report = dict(source=a.bench_json, leg=a.leg, workload=leg.get("workload"), n_sets=n_sets, measured_one_gpu=dict(total_s=leg.get("total_s"), jobs_s=leg.get("jobs_s"), set_wait_s=leg.get("set_wait_s")),
                  model=dict(j1_ms="%.1f + %.1f n" % model["j1"], j2_j3_ms="%.1f + %.1f n + %.1f ceil(n / 4)" % model["jx"], worst_rel_residual=model["worst_rel_residual"],
                             parse_s=round(parse_s, 3), import_s=a.import_s, canary_s=a.canary_s, order="rounds 4-5" if a.old_order else "round 6"),
                  predictions=[])
    print(f"{leg.get('workload')}\nmeasured on one GPU: total {leg.get('total_s')} s (jobs {leg.get('jobs_s')} s, waiting for sets {leg.get('set_wait_s')} s)")
    print(f"model: J1(n targets) = {report['model']['j1_ms']} ms, J2 / J3 call of n jobs = {report['model']['j2_j3_ms']} ms (worst relative residual {model['worst_rel_residual']}), "
          f"parse {parse_s:.3f} s per set, import {a.import_s} s per set")
    for w in a.world:
        sim = simulate(n_sets, w, model, parse_s, a.import_s, canary_s=a.canary_s if w > 1 else 0.0, ready_first=not a.old_order, interleave=False, blocking_first=False)
        report["predictions"].append(sim)
        print(f"N = {w}: predicted total {sim['total_s']:.2f} s, imbalance {sim['imbalance']}, per rank (pairs / first job at / waits / jobs / end): "
              + "  ".join(f"[{o['pairs']} / {o['first_job_at_s']:.2f} / {o['set_wait_s']:.2f} / {o['jobs_s']:.2f} / {o['end_s']:.2f}]" for o in sim["per_rank"]))
    if a.json:
        json.dump(report, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
