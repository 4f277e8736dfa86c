"""Per-kernel times of the three jobs of one pair chain (Commet.py:186-240) on configs[1]-sized sets: J1 (whole index set),
J2 / J3 (index sets restricted to the previous job's result).  python tools/r03_j2_anatomy.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import commet_amd  # noqa: E402
from commet_amd import synth  # noqa: E402

n, L = 10_000_000, 100
b0, o0 = synth.synth_set(0, n, L)
b1, o1 = synth.synth_set(1, n, L)
with commet_amd.Context(k=32, t=2) as ctx:
    a = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
    b = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
    t1, _, _ = ctx.index_and_search(a, [b])
    t2, _, _ = ctx.index_and_search(b, [a], index_select=t1[0])
    t3, _, _ = ctx.index_and_search(a, [b], index_select=t2[0])
    for name, idx, srch, sel in (("J1", a, b, None), ("J2", b, a, t1[0]), ("J3", a, b, t2[0])):
        ctx.set_option("kernel_timing", 1)
        w0 = time.perf_counter()
        for _ in range(3):
            tags, stats, info = ctx.index_and_search(idx, [srch], index_select=sel)
        wall = (time.perf_counter() - w0) / 3
        kt = {k: round(ms / 3, 2) for k, (cnt, ms) in ctx.kernel_times().items() if ms / 3 > 0.05}
        ctx.set_option("kernel_timing", 0)
        w0 = time.perf_counter()
        for _ in range(3):
            ctx.index_and_search(idx, [srch], index_select=sel)
        wall2 = (time.perf_counter() - w0) / 3
        print(name, "indexed", stats[0]["indexed"], "chunks", info["n_chunks"], "wall ms (timed / untimed)", round(wall * 1e3, 2), round(wall2 * 1e3, 2),
              "device", round(sum(kt.values()), 2), kt, flush=True)
