#!/usr/bin/env python3
"""bench.py — reads/sec searched by the MI355X index_and_search path.

Workload (BASELINE.json configs[1]): 2 synthetic sets x 10 M x 100 bp reads,
k=32, t=2.  One step = one whole index_and_search job on read sets already
resident in HBM: set 0 is indexed chunk by chunk (filter zeroing + index
kernels), set 1 is searched against every chunk (search kernels), tag bits come
back to the host.  value = query reads searched per second (whole job).

N > 1 (launched by torch.distributed.run, one rank per GPU): the path shards as
independent (i, j) jobs with no data-path collective, so every rank
runs its own job of the same size on its own GPU -> "weak" scaling; ranks only
meet at the barriers around the timed region (gloo; the GPU work never touches
torch) and value = all ranks' reads / the slowest rank's time.

The JSON line also carries
  roofline     — the dominant kernel (largest share of device time) priced in
                 ALGORITHMIC bytes against the 8 TB/s HBM peak,
                 durations from hipEvents on the stream the kernels run on
  cpu_baseline — the reference CPU tool (oracle/_ref, kind "reference") or our
                 C restatement (oracle/, kind "port") on a bounded sample of the
                 same synthetic sets, one core.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SECTOR = 64                    # bytes per random filter access


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per set (configs[1]: 10 M)")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("-k", type=int, default=32)
    ap.add_argument("-t", type=int, default=2)
    ap.add_argument("--cpu-sample", type=int, default=400_000, help="reads per set of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-probe-count", action="store_true", help="skip the extra (untimed) P_ref counting step")
    return ap.parse_args()


def cpu_baseline(args, b0, o0, b1, o1):
    """Times the reference CPU path on the first `cpu_sample` reads of both sets (rank 0, N=1 only)."""
    n = min(args.cpu_sample, args.reads)
    if n <= 0:
        return None
    from commet_amd import synth
    L = args.read_len
    work = tempfile.mkdtemp(prefix="commet_cpu_")
    try:
        synth.write_fasta(os.path.join(work, "s0.fa"), b0[: n * L], o0[: n + 1])
        synth.write_fasta(os.path.join(work, "s1.fa"), b1[: n * L], o1[: n + 1])
        open(os.path.join(work, "i.txt"), "w").write("s0:s0.fa\n")
        open(os.path.join(work, "s.txt"), "w").write("s1:s1.fa\n")
        ref = os.path.join(ROOT, "oracle", "_ref", "index_and_search")
        if os.path.exists(ref):
            kind, tool = "reference", ref
        else:
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_build/oracle_cli"], check=True)
            kind, tool = "port", os.path.join(ROOT, "oracle", "_build", "oracle_cli")
        t0 = time.time()
        subprocess.run([tool, "-i", "i.txt", "-s", "s.txt", "-o", "out", "-l", "log", "-k", str(args.k), "-t", str(args.t)],
                       cwd=work, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        wall = time.time() - t0
        hot = None
        if kind == "reference":      # its own clock() around index_reads / search_reads (index_and_search.cpp:252-300)
            lines = open(os.path.join(work, "log", "s1_in_s0.log")).read().split("\n")
            hot = float(lines[0].split()[2]) + float(lines[1].split()[2])
        secs = hot if hot else wall
        return {"value": round(n / secs, 1), "unit": "reads/s", "cores": 1, "kind": kind,
                "sample": f"first {n} reads of each of the 2 synthetic sets, k={args.k} t={args.t}; "
                          f"{'tool-reported index+search CPU time' if hot else 'wall time of the tool'} {secs:.2f} s "
                          f"(whole process wall {wall:.2f} s)"}
    finally:
        subprocess.run(["rm", "-rf", work])


def measured_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC profile of this very workload
    (profiles/<dir>/traffic.json written by tools/pmc_summary.py next to the bench.json it was taken with).
    PMC counters cannot be collected from inside the timed run; None when no matching profile exists."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    def natural(name):   # r01_v10 after r01_v9
        return [int(x) if x.isdigit() else x for x in re.split(r"(\d+)", name)]

    for d in sorted(os.listdir(pdir), key=natural) if os.path.isdir(pdir) else []:
        tj, bj = os.path.join(pdir, d, "traffic.json"), os.path.join(pdir, d, "bench.json")
        if not (os.path.exists(tj) and os.path.exists(bj)):
            continue
        try:
            if json.load(open(bj))["config"]["workload"] != workload:
                continue
            e = json.load(open(tj)).get(kernel)
            if e and "hbm_bytes_per_launch" in e:
                best = (e["hbm_bytes_per_launch"], f"profiles/{d}/traffic.json")
        except Exception:
            continue
    return best


def main():
    args = parse_args()
    from commet_amd import sharding
    ranks = sharding.Ranks(backend="gloo")   # host-side barrier / MAX only; N=1 needs no torch at all
    world, rank, local_rank = ranks.world, ranks.rank, ranks.local_rank

    import numpy as np  # noqa: F401
    import commet_amd
    from commet_amd import synth

    n, L, k, t = args.reads, args.read_len, args.k, args.t
    # every rank owns one (i, j) job of the N x N matrix: sets (2r, 2r+1)
    b0, o0 = synth.synth_set(2 * rank, n, L, base_set=2 * rank)
    b1, o1 = synth.synth_set(2 * rank + 1, n, L, base_set=2 * rank)

    # COMMET_FORCE_DEVICE: debugging aid to run several ranks on one GPU (never set by the driver)
    device = int(os.environ.get("COMMET_FORCE_DEVICE", local_rank))
    ctx = commet_amd.Context(k=k, t=t, device=device)
    t_up = time.perf_counter()
    irs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
    qrs = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
    ctx.synchronize()
    upload_s = time.perf_counter() - t_up      # PCIe + packing of both sets (reported, never part of `value`)

    probes = None
    if not args.no_probe_count:
        ctx.set_option("count_probes", 1)     # untimed: P_ref for the search kernel's algorithmic bytes
        _, _, inf = ctx.index_and_search(irs, [qrs])
        probes = inf["probes"]
        ctx.set_option("count_probes", 0)
    for _ in range(args.warmup):
        ctx.index_and_search(irs, [qrs])

    acc = dict(index_kernel_ms=0.0, search_ms=0.0, zero_ms=0.0, index_launches=0, search_launches=0)
    last = {}

    def step():
        tags, stats, info = ctx.index_and_search(irs, [qrs])
        for f in acc:
            acc[f] += info[f]
        last["stats"], last["info"] = stats, info

    elapsed = sharding.timed_region(ranks, ctx.synchronize, step, args.steps)
    stats, info = last["stats"], last["info"]

    if rank == 0:
        steps = args.steps
        ms_per_step = elapsed * 1000.0 / steps
        value = world * n * steps / elapsed
        # ---- roofline of the dominant kernel, algorithmic bytes per launch ----
        kmers = info["kmers_indexed"]
        idx_bytes_step = info["reads_indexed"] * (L / 4.0) + 4.0 * kmers * 2 * SECTOR
        srch_bytes_step = None
        if probes is not None:
            srch_bytes_step = info["reads_scanned"] * (L / 4.0 + 1 / 8.0) + probes * SECTOR
        idx_ms = acc["index_kernel_ms"] / steps
        srch_ms = acc["search_ms"] / steps
        # dominant KERNEL: the index phase of one chunk is a chain of >= 3 comparable streaming kernels (scatter1,
        # scatter2, build; see profiles/), the search phase is one kernel per launch
        idx_per_kernel = idx_ms / max(acc["index_launches"] / steps, 1) / 3.0
        srch_per_kernel = srch_ms / max(acc["search_launches"] / steps, 1)
        if srch_bytes_step is None or idx_per_kernel > srch_per_kernel:
            name, kms, kbytes, launches = "index_kernel", idx_ms, idx_bytes_step, acc["index_launches"] / steps
        else:
            name = "search_group_kernel" if acc["search_launches"] / steps < info["n_chunks"] else "search_kernel"
            kms, kbytes, launches = srch_ms, srch_bytes_step, acc["search_launches"] / steps
        achieved = kbytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        workload = (f"2 synthetic sets x {n} x {L} bp reads, k={k} t={t}, index set 0 + search set 1 "
                    f"per GPU (BASELINE configs[1]), inputs resident in HBM")
        tr = measured_traffic(workload, name)
        roofline = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": tr[0] if tr else None,
                    "traffic_source": tr[1] if tr else None,
                    # as executed: measured HBM bytes of the launch over its measured duration, against the same peak
                    "traffic_GBps": round(tr[0] / (kms / max(launches, 1) * 1e-3) / 1e9, 1) if tr and kms > 0 else None,
                    "traffic_frac": round(tr[0] / (kms / max(launches, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tr and kms > 0 else None,
                    "launches_per_step": launches, "avg_launch_ms": round(kms / max(launches, 1), 3),
                    "algorithmic_bytes_per_launch": round(kbytes / max(launches, 1)),
                    "note": "algorithmic bytes = reference probes x 64-B sectors (SURVEY 8d); frac > 1 or traffic < algorithmic "
                            "means one HBM request serves several reference probes (strand-paired, chunk-interleaved plane A)"}
        out = {
            "metric": "reads/sec searched (index_and_search, k=%d)" % k,
            "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32" if k <= 32 else "u64", "data": "synthetic",
            "config": {"workload": workload,
                       "reads_per_set": n, "read_len": L, "k": k, "t": t, "jobs": world,
                       "parallelism": f"{world} independent (i,j) jobs, no collective"},
            "roofline": roofline,
            "detail": {"chunks": info["n_chunks"], "kmers_indexed": kmers, "reads_scanned": info["reads_scanned"],
                       "shared": stats[0]["shared"], "searched_last_pass": stats[0]["searched"],
                       "index_kernel_ms": round(idx_ms, 3), "search_kernel_ms": round(srch_ms, 3),
                       "filter_zero_ms": round(acc["zero_ms"] / steps, 3),
                       "index_alg_GBps": round(idx_bytes_step / (idx_ms * 1e-3) / 1e9, 1) if idx_ms else None,
                       "search_alg_GBps": round(srch_bytes_step / (srch_ms * 1e-3) / 1e9, 1) if srch_bytes_step and srch_ms else None,
                       "p_ref_probes": probes, "upload_and_pack_s": round(upload_s, 3),
                       "end_to_end_reads_per_s_incl_pcie": round(n / (upload_s + elapsed / steps), 1)},
        }
        if world == 1:
            out["cpu_baseline"] = cpu_baseline(args, b0, o0, b1, o1)
        print(json.dumps(out), flush=True)

    irs.close()
    qrs.close()
    ctx.close()
    ranks.close()


if __name__ == "__main__":
    main()
