"""Timing ablations of part_scatter1_kernel (option part_debug, results become wrong on purpose): run under
  rocprofv3 --kernel-trace --output-format csv -d DIR -o abl -- python3 tools/s1_ablate.py
then  python tools/s1_ablate.py --report DIR/abl_kernel_trace.csv
Each debug value does REP index_reads calls of one BASELINE-configs[1] chunk; the report lists the scatter1 / hist /
scatter2 / build durations of the calls in launch order.  S1_DEBUGS picks the values (scatter1: 1 no write-out, 2 no
placement, 4 no counting atomics; scatter2: 32 no placement, 64 no write-out, 128 no cursor reservation, 256 no counting;
kernels downstream of an ablated one are not launched), S1_REP the repeats, S2_SWIZZLE=1 sweeps the slab order of
scatter2 with the same values instead."""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

DEBUGS = [int(x) for x in os.environ.get("S1_DEBUGS", "0,1,2,4,16,3,7,23").split(",")]
REP = int(os.environ.get("S1_REP", "3"))


def report(path):
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for kern in ("part_scatter1_kernel", "part_hist_kernel", "part_scatter2_kernel", "part_build_kernel"):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if kern in r["Kernel_Name"]]
        print(kern)
        for i, dbg in enumerate(DEBUGS):
            print(f"  debug={dbg:2d}: " + " ".join(f"{x:.3f}" for x in d[i * REP:(i + 1) * REP]))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        return report(sys.argv[2])
    import commet_amd
    from commet_amd import synth
    n = 7_246_377
    b0, o0 = synth.synth_set(0, n, 100)
    with commet_amd.Context(k=32, t=2) as ctx:
        ctx.set_option("index_mode", 2)
        rs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
        for dbg in DEBUGS:
            if "S2_SWIZZLE" in os.environ:      # sweep scatter2's slab order instead of scatter1's ablations
                ctx.set_option("s2_swizzle", dbg)
                dbg = 0
            ctx.set_option("part_debug", dbg)
            for _ in range(REP):
                ctx.filter_reset()
                ctx.index_reads(rs)
                print(dbg, ctx.last_kernel_ms()[0], flush=True)


if __name__ == "__main__":
    main()
