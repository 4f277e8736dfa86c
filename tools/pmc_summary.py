"""Summarises rocprofv3 outputs of a bench.py run into per-kernel averages:
  python tools/pmc_summary.py profiles/<dir>
reads kernel_stats.csv (--kernel-trace --stats), pmc_fetch_size.csv / pmc_write_size.csv (--pmc FETCH_SIZE /
WRITE_SIZE, separate passes) and writes <dir>/traffic.json:
  {kernel: {calls, avg_ms, fetch_bytes_per_launch, write_bytes_per_launch, hbm_bytes_per_launch}}
FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies the 128-B
requests of wide coalesced streams at 64 B, so streaming kernels' reads are doubled; random 4-B gathers (one 64-B
request per probe) and atomics are taken as reported — for the search kernel this is confirmed by
FETCH_SIZE == P_ref x 64 B within 2 %."""
import collections
import csv
import json
import os
import sys

STREAMING = ("part_scatter2", "part_build", "part_hist", "part_scatter1", "interleave_a",   # wide coalesced readers
             "search_wide")   # rows of 1.3 KB fetched 16 bytes per lane: FETCH_SIZE 8.32 TB per launch against 13.86 TB of rows asked for (r03_c5)


def short(name):
    n = name.replace("void ", "").replace("commet::", "")
    base = n.split("(")[0].split("<")[0]
    # the probe-counting instantiations (COUNT = true) only run in bench.py's untimed P_ref step
    return base + "[count]" if base.startswith("search") and ", true>(" in n else base


def main(d):
    out = collections.OrderedDict()
    for r in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))):
        k = short(r["Name"])
        e = out.setdefault(k, dict(calls=0, total_ms=0.0))
        e["calls"] += int(r["Calls"])
        e["total_ms"] += float(r["TotalDurationNs"]) / 1e6
    for e in out.values():
        e["avg_ms"] = round(e["total_ms"] / max(e["calls"], 1), 4)
        e["total_ms"] = round(e["total_ms"], 3)
    for fname, field in (("pmc_fetch_size.csv", "fetch"), ("pmc_write_size.csv", "write")):
        p = os.path.join(d, fname)
        if not os.path.exists(p):
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
        for k, v in acc.items():
            if k in out:
                mean = sum(v) / len(v)
                if field == "fetch" and k.startswith(STREAMING):
                    mean *= 2.0
                out[k][field + "_bytes_per_launch"] = round(mean)
    for e in out.values():
        if "fetch_bytes_per_launch" in e or "write_bytes_per_launch" in e:
            e["hbm_bytes_per_launch"] = e.get("fetch_bytes_per_launch", 0) + e.get("write_bytes_per_launch", 0)
    # SQ counters (one more pass), per launch: VALU wave-instructions, the VALU's busy share (SQ_ACTIVE_INST_VALU counts
    # quad-cycles per SIMD: x 4 cycles / (1024 SIMDs x launch duration at 2.4 GHz)), share of wave-cycles spent waiting
    p = os.path.join(d, "pmc_sq.csv")
    if os.path.exists(p):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(p)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[short(r["Kernel_Name"])].add(r["Dispatch_Id"])
        for k, c in acc.items():
            n = max(len(disp[k]), 1)
            if k in out and c.get("SQ_WAVE_CYCLES"):
                cyc = out[k]["avg_ms"] * 1e-3 * 2.4e9
                out[k]["sq"] = {"launches_sampled": n,
                                "valu_insts_per_launch": round(c.get("SQ_INSTS_VALU", 0) / n),
                                "valu_busy_share": round(c.get("SQ_ACTIVE_INST_VALU", 0) / n * 4 / (1024 * cyc), 3) if cyc else None,
                                "wait_share_of_wave_cycles": round(c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 3),
                                "vmem_rd_insts_per_launch": round(c.get("SQ_INSTS_VMEM_RD", 0) / n),
                                "lds_insts_per_launch": round(c.get("SQ_INSTS_LDS", 0) / n), "waves_per_launch": round(c.get("SQ_WAVES", 0) / n)}
    meta = {"note": "per launch; FETCH_SIZE doubled for the wide streaming readers (gfx950, MI355X_MICROARCH.md HBM section)"}
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        meta["source_hash"] = bench.source_hash()
        meta["workload"] = json.load(open(os.path.join(d, "bench.json")))["config"]["workload"]
    except Exception as ex:      # a summary without the meta block is still usable (bench.py then marks it stale)
        meta["error"] = repr(ex)
    out["_meta"] = meta
    json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
    out.pop("_meta")
    for k, e in out.items():
        print(f"{k:28s} calls={e['calls']:4d} avg_ms={e['avg_ms']:9.3f} fetch={e.get('fetch_bytes_per_launch', 0) / 1e9:8.2f} GB "
              f"write={e.get('write_bytes_per_launch', 0) / 1e9:8.2f} GB")


if __name__ == "__main__":
    main(sys.argv[1])
