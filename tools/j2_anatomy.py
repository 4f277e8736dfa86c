"""Per-kernel times of one pair chain J1 -> J2 -> J3 of Commet.py's job sequence (Commet.py:186-240) on two synthetic sets: J2 / J3
index a selection-restricted set (the previous job's result), the jobs 90 of the 99 of a 10-set matrix are made of.
  python tools/j2_anatomy.py [reads = 50000000]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import commet_amd  # noqa: E402
from commet_amd import synth  # noqa: E402


def main():
    n, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000), 100
    b0, o0 = synth.synth_set(0, n, L)
    b1, o1 = synth.synth_set(1, n, L)
    with commet_amd.Context(k=32, t=2) as ctx:
        s0 = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
        s1 = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
        for rep in range(2):
            out = {}
            ctx.set_option("kernel_timing", 1)
            t1, st1, info1 = ctx.index_and_search(s0, [s1])                 # J1: S_1 in S_0
            kt1 = ctx.kernel_times()
            ctx.set_option("kernel_timing", 0)
            out["J1"] = dict(total_ms=round(info1["total_ms"], 2), index_ms=round(info1["index_ms"], 2), search_ms=round(info1["search_ms"], 2),
                             chunks=info1["n_chunks"], shared=st1[0]["shared"],
                             kernels={k: [c, round(ms, 3)] for k, (c, ms) in sorted(kt1.items(), key=lambda kv: -kv[1][1]) if ms > 0.5})
            for name, idx, srch, sel in (("J2", s1, s0, t1[0]), ("J3", s0, s1, None)):
                if name == "J3":
                    sel = out["_t2"]
                ctx.set_option("kernel_timing", 1)
                tags, st, info = ctx.index_and_search(idx, [srch], index_select=sel)
                kt = ctx.kernel_times()
                ctx.set_option("kernel_timing", 0)
                out[name] = dict(total_ms=round(info["total_ms"], 2), index_ms=round(info["index_ms"], 2), search_ms=round(info["search_ms"], 2),
                                 chunks=info["n_chunks"], indexed=st[0]["indexed"], shared=st[0]["shared"],
                                 kernels={k: [c, round(ms, 3)] for k, (c, ms) in sorted(kt.items(), key=lambda kv: -kv[1][1]) if ms > 0.05})
                if name == "J2":
                    out["_t2"] = tags[0]
            out.pop("_t2")
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
