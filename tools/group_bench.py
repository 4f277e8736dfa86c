"""group_bench.py — one index_and_search job at several chunk_group settings: does a pass over eight chunk filters (search_group8_kernel,
2 / 3 / 4 / 6 mask words by the set's longest read) beat two passes over four (search_group_kernel, masks in LDS)?
  python tools/group_bench.py [--index-reads 12000000] [--search-reads 4000000] [--read-len 250 | --ragged 100-250] [-k 32] [-t 2]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--index-reads", type=int, default=12_000_000)
    ap.add_argument("--search-reads", type=int, default=4_000_000)
    ap.add_argument("--read-len", type=int, default=250)
    ap.add_argument("--ragged", default=None)
    ap.add_argument("-k", type=int, default=32)
    ap.add_argument("-t", type=int, default=2)
    ap.add_argument("--groups", type=int, nargs="+", default=[8, 4])
    a = ap.parse_args()
    import commet_amd
    from commet_amd import synth
    if a.ragged:
        lo, hi = (int(x) for x in a.ragged.split("-"))
        mk = lambda s, n: synth.synth_set_ragged(s, n, lo, hi)   # noqa: E731
    else:
        mk = lambda s, n: synth.synth_set(s, n, a.read_len)      # noqa: E731
    out = {"workload": f"index {a.index_reads} reads, search {a.search_reads} reads of {a.ragged or a.read_len} bp, k={a.k} t={a.t}"}
    with commet_amd.Context(k=a.k, t=a.t) as ctx:
        irs = commet_amd.ReadSet.from_files(ctx, [mk(0, a.index_reads)])
        qrs = commet_amd.ReadSet.from_files(ctx, [mk(1, a.search_reads)])
        ref = None
        for g in a.groups:
            ctx.set_option("chunk_group", g)
            ctx.index_and_search(irs, [qrs])
            ctx.set_option("kernel_timing", 1)
            tags, stats, info = ctx.index_and_search(irs, [qrs])
            kt = {k_: round(ms, 3) for k_, (c, ms) in ctx.kernel_times().items() if k_.startswith(("search", "tq_", "interleave"))}
            ctx.set_option("kernel_timing", 0)
            same = ref is None or bool((tags[0] == ref).all())
            ref = tags[0] if ref is None else ref
            out[f"chunk_group_{g}"] = {"chunks": info["n_chunks"], "search_ms": round(info["search_ms"], 3), "index_ms": round(info["index_kernel_ms"], 3),
                                       "search_launches": info["search_launches"], "kernels_ms": kt, "shared": stats[0]["shared"], "same_bits_as_first": same}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
