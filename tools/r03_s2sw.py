"""scatter2's slab order (option s2_swizzle: number of interleaved slab ranges) with the 7 + 8 split, same context and workspaces
  python tools/r03_s2sw.py [reads]"""
import os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/commet_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import commet_amd
from commet_amd import synth
n, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000), 100
b0, o0 = synth.synth_set(0, n, L)
b1, o1 = synth.synth_set(1, n, L)
with commet_amd.Context(k=32, t=2) as ctx:
    irs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
    qrs = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
    ref = None
    for sw in (128, 0, 32, 64, 256, 512, 1024, 128):
        ctx.set_option("s2_swizzle", sw)
        tags, stats, info = ctx.index_and_search(irs, [qrs])
        if ref is None:
            ref = tags[0].copy()
        assert np.array_equal(tags[0], ref), "tags differ"
        ctx.set_option("kernel_timing", 1)
        for _ in range(4):
            ctx.index_and_search(irs, [qrs])
        kt = ctx.kernel_times()
        ctx.set_option("kernel_timing", 0)
        print("s2_swizzle", sw, "scatter2 ms/job", round(kt["part_scatter2_packed_kernel"][1] / 4, 3), "build", round(kt["part_build_kernel"][1] / 4, 3), flush=True)
