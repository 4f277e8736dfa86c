"""What the first job on a search set costs beyond a steady-state job (configs[1]): the kernels of the query-list build
(tq_count / tq_scan / tq_bounds / tq_fill), timed with the library's per-launch clock.
  python tools/list_build_times.py [reads = 10000000]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import commet_amd  # noqa: E402
from commet_amd import synth  # noqa: E402


def main():
    n, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000), 100
    b0, o0 = synth.synth_set(0, n, L)
    b1, o1 = synth.synth_set(1, n, L)
    with commet_amd.Context(k=32, t=2) as ctx:
        irs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
        qrs = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
        for _ in range(2):
            ctx.index_and_search(irs, [qrs])
        out = {}
        for rep in range(3):
            qrs.drop_cache()
            ctx.synchronize()
            ctx.set_option("kernel_timing", 1)
            t0 = time.perf_counter()
            ctx.index_and_search(irs, [qrs])
            ctx.synchronize()
            first_ms = (time.perf_counter() - t0) * 1e3
            kt = ctx.kernel_times()
            ctx.set_option("kernel_timing", 0)
            t0 = time.perf_counter()
            ctx.index_and_search(irs, [qrs])
            ctx.synchronize()
            steady_ms = (time.perf_counter() - t0) * 1e3
            out = dict(first_job_ms=round(first_ms, 3), steady_job_ms=round(steady_ms, 3),
                       list_kernels_ms={k: round(ms, 3) for k, (cnt, ms) in kt.items() if k.startswith("tq_") and "probe" not in k and "replay" not in k},
                       query_list_bytes=qrs.cache_bytes)
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
