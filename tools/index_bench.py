"""Times the index path alone (one chunk of BASELINE configs[1]) in both modes and checks
that the bucketed construction finds the same reads as the atomic kernel.
Usage: python tools/index_bench.py [reads] [k]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import commet_amd  # noqa: E402
from commet_amd import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 7_000_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    L = 100
    b0, o0 = synth.synth_set(0, n, L)
    b1, o1 = synth.synth_set(1, 1_000_000, L)
    res = {}
    for mode in (2, 3, 1):          # 2 = bucketed, 3 = bucketed without the uniform-length fast path, 1 = atomic
        with commet_amd.Context(k=k, t=2) as ctx:
            ctx.set_option("index_mode", min(mode, 2))
            ctx.set_option("part_no_uni", int(mode == 3))
            rs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
            qs = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
            times = []
            for it in range(4 if mode != 1 else 2):
                ctx.filter_reset()
                fed = ctx.index_reads(rs)
                times.append(ctx.last_kernel_ms()[0])
            found, _, nf = ctx.search_reads(qs)
            res[mode] = (fed, nf, found)
            if k <= 28:      # small enough to pull the whole filter back: bit-exact comparison of the two constructions
                res[mode] = res[mode] + (ctx.export_filter_reference(),)
            print(f"mode={mode} kmers={fed} index_ms={['%.2f' % t for t in times]} -> {fed * 4 / min(times) / 1e6:.2f} G keys/s; found={nf}",
                  flush=True)
    for m in (2, 3):
        assert res[1][0] == res[m][0] and res[1][1] == res[m][1] and np.array_equal(res[1][2], res[m][2]), "MISMATCH"
        if len(res[1]) > 3:
            assert np.array_equal(res[1][3], res[m][3]), "FILTER MISMATCH"
    print("modes agree")


if __name__ == "__main__":
    main()
