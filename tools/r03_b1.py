"""coarse buckets of the bucketed index construction: 2^8 (default at k = 32) against 2^7 (runs of 128 keys = 512 bytes out of scatter1,
256 final buckets per coarse one in scatter2) — same context, same workspaces, so the allocation plays no part
  python tools/r03_b1.py [k] [reads]"""
import os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/commet_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import commet_amd
from commet_amd import synth
k = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n, L = (int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000), 100
b0, o0 = synth.synth_set(0, n, L)
b1, o1 = synth.synth_set(1, n, L)
with commet_amd.Context(k=k, t=2) as ctx:
    irs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
    qrs = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
    ref = None
    for rep in range(2):
        for b1bits in (8, 7):
            ctx.set_option("part_b1", b1bits)
            tags, stats, info = ctx.index_and_search(irs, [qrs])
            if ref is None:
                ref = tags[0].copy()
            assert np.array_equal(tags[0], ref), "tags differ"
            ctx.set_option("kernel_timing", 1)
            for _ in range(4):
                ctx.index_and_search(irs, [qrs])
            kt = ctx.kernel_times()
            ctx.set_option("kernel_timing", 0)
            idx = {k2: round(v[1] / 4, 3) for k2, v in kt.items() if k2.startswith("part_")}
            print("part_b1", b1bits, "index kernels ms/job:", idx, "sum", round(sum(idx.values()), 3), flush=True)
