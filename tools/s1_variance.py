"""Is scatter1's run-to-run spread (4.25 - 4.95 ms per configs[1] step between processes on one box) a property of the
process or of the allocation?  One process, several contexts one after the other, per-kernel times of each.
  python tools/s1_variance.py [contexts]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import commet_amd  # noqa: E402
from commet_amd import synth  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    n, L = 10_000_000, 100
    b0, o0 = synth.synth_set(0, n, L)
    b1, o1 = synth.synth_set(1, n, L)
    for rep in range(reps):
        with commet_amd.Context(k=32, t=2) as ctx:
            irs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
            qrs = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
            ctx.index_and_search(irs, [qrs])
            ctx.set_option("kernel_timing", 1)
            for _ in range(3):
                ctx.index_and_search(irs, [qrs])
            kt = ctx.kernel_times()
            ctx.set_option("kernel_timing", 0)
            per = {name: ms / 3 for name, (_, ms) in kt.items()}
            print(rep, {k: round(v, 2) for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:5]}, flush=True)
            irs.close()
            qrs.close()


if __name__ == "__main__":
    main()
