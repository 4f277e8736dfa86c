"""How much of the step's run-to-run spread belongs to the ALLOCATION of the scatter workspaces, and what keeping the fastest of
several candidates (option ws_candidates, capi.hip alloc_fastest) does about it.  One process; for every value of ws_candidates
given, `contexts` fresh contexts one after the other (same resident input arrays), per context: the step time of configs[1]
(median of 5 steps, index lanes as shipped) and the times of the kernels that sweep the workspaces.
  python tools/s1_variance.py [contexts = 8] [ws_candidates values, comma separated = 1,4]"""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import commet_amd  # noqa: E402
from commet_amd import synth  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    cands = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,4").split(",")]
    n, L = 10_000_000, 100
    b0, o0 = synth.synth_set(0, n, L)
    b1, o1 = synth.synth_set(1, n, L)
    for cand in cands:
        steps_ms = []
        for rep in range(reps):
            with commet_amd.Context(k=32, t=2) as ctx:
                ctx.set_option("ws_candidates", cand)
                irs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
                qrs = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
                t0 = time.perf_counter()
                ctx.index_and_search(irs, [qrs])
                ctx.synchronize()
                cold_ms = (time.perf_counter() - t0) * 1e3
                ctx.index_and_search(irs, [qrs])
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    ctx.index_and_search(irs, [qrs])
                    ctx.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                ctx.set_option("kernel_timing", 1)
                ctx.set_option("index_lanes", 1)
                for _ in range(3):
                    ctx.index_and_search(irs, [qrs])
                kt = {name: ms / 3 for name, (_, ms) in ctx.kernel_times().items()}
                ctx.set_option("kernel_timing", 0)
                step = statistics.median(ts)
                steps_ms.append(step)
                print(json.dumps(dict(ws_candidates=cand, context=rep, step_ms=round(step, 3), cold_first_job_ms=round(cold_ms, 1),
                                      scatter1_ms=round(kt.get("part_scatter1_kernel", 0), 3), scatter2_ms=round(kt.get("part_scatter2_packed_kernel", 0), 3),
                                      build_ms=round(kt.get("part_build_kernel", 0), 3))), flush=True)
                irs.close()
                qrs.close()
        print(json.dumps(dict(ws_candidates=cand, contexts=reps, step_ms_min=round(min(steps_ms), 3), step_ms_max=round(max(steps_ms), 3),
                              step_ms_median=round(statistics.median(steps_ms), 3),
                              spread_pct=round(100.0 * (max(steps_ms) - min(steps_ms)) / statistics.median(steps_ms), 2))), flush=True)


if __name__ == "__main__":
    main()
