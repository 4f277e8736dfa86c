#!/usr/bin/env python3
"""bench.py — reads/sec searched by the MI355X index_and_search path.

Workload (BASELINE.json configs[1]): 2 synthetic sets x 10 M x 100 bp reads,
k=32, t=2.  One step = one whole index_and_search job on read sets already
resident in HBM: set 0 is indexed chunk by chunk (filter zeroing + index
kernels), set 1 is searched against every chunk (search kernels), tag bits come
back to the host.  value = query reads searched per second (whole job).

N > 1 (one rank per GPU; started by torch.distributed.run, or by this script
itself as plain child processes when no launcher is around it): the path shards
as independent (i, j) jobs with no data-path collective, so every
rank runs its own job of the same size on its own GPU -> "weak" scaling; ranks
only meet at the barriers around the timed region (a TCP store of rank 0; no
torch in the rank processes) and value = all ranks' reads / the slowest rank's time.

The JSON line also carries
  roofline     — the kernel with the largest measured share of the step's device
                 time (a hipEvent pair around every launch, commet_kernel_times,
                 taken in untimed extra steps), priced in the HBM bytes it really
                 moves: rocprofv3 FETCH_SIZE + WRITE_SIZE per launch from the
                 committed profile of this workload (profiles/*/traffic.json,
                 marked stale when the kernel sources changed since) over the
                 live launch duration, against the 8 TB/s peak: frac <= 1.
                 request_rate prices the same kernel in 64-byte memory requests
                 against the random-gather ceiling measured in this run
                 (commet_membench); whole_step sums every kernel.  The figure in
                 reference probes (P_ref x 64 B) stays in `detail`.
  cpu_baseline — the reference CPU tool (oracle/_ref, kind "reference") or our
                 C restatement (oracle/, kind "port") on a bounded sample of the
                 same synthetic sets: one copy, and one copy per host core.
  matrix       — BASELINE configs[3] (10 sets x 50 M reads, the 10 x 10 matrix)
                 through commet_amd.matrix split over the N ranks, filter and
                 load times included — the SAME workload at every N, one GPU
                 included, so the per-N values are one curve (detail.matrix has
                 the per-rank profile; detail.matrix_configs2 is configs[2],
                 10 x 10 M reads, on one GPU; --no-matrix skips both).
"""
import argparse
import hashlib
import json
import os
import re
import shutil
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
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="reads per set of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-full", action="store_true",
                    help="also run the reference CPU tool ONCE on the whole job (both full sets, one core; 10-15 minutes at configs[1]) and "
                         "compare its .bv bytes and log line with the GPU's: cpu_baseline.full_job (off by default: the driver's run stays short)")
    ap.add_argument("--no-probe-count", action="store_true", help="skip the extra (untimed) P_ref counting step")
    ap.add_argument("--no-kernel-times", action="store_true", help="skip the extra (untimed) per-kernel timing steps")
    ap.add_argument("--kt-steps", type=int, default=3, help="untimed steps of the per-kernel timing leg")
    ap.add_argument("--traffic", default=None, help="traffic.json to price the roofline with (default: the newest matching one under profiles/)")
    ap.add_argument("--no-matrix", action="store_true", help="skip the configs[2] matrix leg (detail.matrix)")
    ap.add_argument("--matrix-sets", type=int, default=10)
    ap.add_argument("--matrix-reads", type=int, default=None,
                    help="reads per set of the matrix leg (default: 50 M = configs[3] at every N when the host holds the files, "
                         "plus 10 M = configs[2] on one GPU)")
    ap.add_argument("--skew", type=float, default=0.0,
                    help="fraction of every set's reads replaced by low-complexity / repeated reads (poly-A, tandem repeats, a shared "
                         "1000-read library): the non-uniform data leg, synth.skew_set")
    ap.add_argument("--ragged", default="50-150",
                    help="LO-HI: a second, untimed-from-the-headline leg on RAGGED sets (read lengths uniform in [LO, HI], same copy / mutation rules; "
                         "50-150 has the bases, k-mers and first-hit windows per read of the 100-bp sets) — one GPU only: the configs[1] step and the "
                         "10-set matrices of configs[2] and configs[3], reported in detail.ragged; 'none' skips it")
    ap.add_argument("--ragged-only", action="store_true", help="the headline step itself on ragged sets (A/B runs, profiles): value is then the ragged rate")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, meet at the barriers, print the line's launch fields and leave (no GPU work: the CPU test of the launch path)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): this process, which has made no HIP
    call and never will, starts the N ranks as plain `python bench.py` CHILD processes (one rank per GPU; RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR = 127.0.0.1 and a free MASTER_PORT in their environment; no torch anywhere: the ranks meet
    over commet_amd.sharding's TCP store), passes their output through and leaves with the first non-zero exit code."""
    from commet_amd import sharding
    return sharding.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                env=dict(os.environ, COMMET_SELF_LAUNCHED="1"))


def host_cores():
    """CPUs this process may really use: the cgroup quota when there is one, else the affinity mask."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except Exception:
        pass
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def host_memory_free(scratch_root):
    """Bytes of host memory this process can still take: MemAvailable, the cgroup's limit minus its use, and (the matrix inputs are
    files in it) the free space of the scratch root — the smallest of the three."""
    free = []
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                free.append(int(line.split()[1]) * 1024)
    except Exception:
        pass
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            v = open(lim).read().strip()
            if v != "max" and int(v) < (1 << 60):
                free.append(int(v) - int(open(cur).read()))
        except Exception:
            pass
    try:
        st = os.statvfs(scratch_root)
        free.append(st.f_bavail * st.f_frsize)
    except Exception:
        pass
    return min(free) if free else None


def matrix_memory_needed(n_sets, n_reads, read_len, workers):
    """host bytes of the matrix leg: the FASTA files (in /dev/shm they are memory) + the sets' packed images + the generator state of the worker processes
    (~2.5 bytes per base each) + the driver's own parsing of two sets at a time (mapped files: no copy) and its filter .bv files"""
    fasta = n_sets * n_reads * (read_len + 12)
    images = n_sets * n_reads * (read_len * 0.4 + 8)      # several ranks: every set's packed image in the scratch root, too (counted always)
    return int(fasta + images + workers * n_reads * read_len * 2.5 + (2 << 30))


def cpu_full_job(args, b0, b1, gpu_tags, gpu_stats):
    """--cpu-full: the reference tool (oracle/_ref, else our C restatement) on the WHOLE job, one copy on one core
    (index_and_search.cpp:252-300: its own clock() around index_reads / search_reads), its .bv bytes and its
    [indexed, searched, shared] line compared with the GPU's."""
    import numpy as np
    from commet_amd import synth
    n, L = args.reads, args.read_len
    work = tempfile.mkdtemp(prefix="commet_cpufull_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None)
    try:
        synth.write_fasta_fast(os.path.join(work, "s0.fa"), b0, n, L)
        synth.write_fasta_fast(os.path.join(work, "s1.fa"), b1, n, L)
        open(os.path.join(work, "i.txt"), "w").write(f"s0:{work}/s0.fa\n")
        open(os.path.join(work, "s.txt"), "w").write(f"s1:{work}/s1.fa\n")
        ref = os.path.join(ROOT, "oracle", "_ref", "index_and_search")
        if os.path.exists(ref):
            kind, tool = "reference", ref
        else:
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_build/oracle_cli"], check=True)
            kind, tool = "port", os.path.join(ROOT, "oracle", "_build", "oracle_cli")
        print(f"bench.py [{time.perf_counter() - _T0:7.1f} s] --cpu-full: {kind} tool on 2 x {n} reads, one core (minutes)", file=sys.stderr, flush=True)
        t0 = time.time()
        pr = subprocess.Popen([tool, "-i", os.path.join(work, "i.txt"), "-s", os.path.join(work, "s.txt"), "-o", "out", "-l", "log",
                               "-k", str(args.k), "-t", str(args.t)], cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        while pr.poll() is None:                      # a line a minute: a long silent run looks hung to whoever watches
            try:
                pr.wait(timeout=60)
            except subprocess.TimeoutExpired:
                print(f"bench.py [{time.perf_counter() - _T0:7.1f} s] --cpu-full: running, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
        wall = time.time() - t0
        if pr.returncode != 0:
            return {"error": f"the CPU tool left with code {pr.returncode}"}
        lines = open(os.path.join(work, "log", "s1_in_s0.log")).read().split("\n")
        hot = float(lines[0].split()[2]) + float(lines[1].split()[2]) if kind == "reference" else wall
        nums = [int(x) for x in re.findall(r"\d+", lines[3])]
        data = open(os.path.join(work, "out", "s1.fa_in_s0.bv"), "rb").read()
        body = data[data.index(b"\n", data.index(b"#")) + 1:]
        cpu_bits = np.frombuffer(body[: n // 8 + 1], dtype=np.uint8)
        same = bool(np.array_equal(cpu_bits, np.asarray(gpu_tags[: n // 8 + 1], dtype=np.uint8)))
        return {"kind": kind, "cores": 1, "reads_per_s": round(n / hot, 1), "index_s": float(lines[0].split()[2]) if kind == "reference" else None,
                "search_s": float(lines[1].split()[2]) if kind == "reference" else None, "wall_s": round(wall, 1),
                "log_line": lines[3], "bv_bytes_equal_gpu": same,
                "log_numbers_equal_gpu": nums == [gpu_stats["indexed"], gpu_stats["searched"], gpu_stats["shared"]],
                "what": f"the whole job (2 x {n} reads, k={args.k} t={args.t}) on one host core; .bv of the search set byte-compared with the GPU's tags"}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def cpu_baseline(args, b0, b1, gpu_info=None):
    """Times the reference CPU path (rank 0, N=1 only) on two bounded samples of the job, one independent copy per host core
    (P stated):
      heavy (the line's `value`): the index set's FIRST CHUNK fed whole (70 % of the set: as many reads as stay under the reference's
        max_kmer, index_and_search.cpp:73,146 — the filter is then as full as the real job's first filter, 11 % at configs[1]) and
        every 50th read of the search set searched against it (a quarter of them copies, as in the whole set); the tool's own
        index / search clocks are scaled to the whole job: index x (reads of the set / reads indexed), search x 50 x (read scans of
        the whole job, from the GPU's run, / reads of the set);
      light (secondary, what rounds 1-5 reported): the first `cpu_sample` reads of both sets — filters 1.6 % full and every sampled
        search read a copy of an indexed one, which flatters the CPU by ~2x."""
    n = min(args.cpu_sample, args.reads)
    if n <= 0:
        return None
    if gpu_info and gpu_info.get("n_chunks", 0) > 64:
        n = min(n, 50_000)       # (a job of thousands of chunk filters — configs[4] — searches the sample against every one of them: minutes per 1 M reads on a core)
    import numpy as np
    from commet_amd import synth
    L = args.read_len
    work = tempfile.mkdtemp(prefix="commet_cpu_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None)
    try:
        ref = os.path.join(ROOT, "oracle", "_ref", "index_and_search")
        if os.path.exists(ref):
            kind, tool = "reference", ref
        else:
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_build/oracle_cli"], check=True)
            kind, tool = "port", os.path.join(ROOT, "oracle", "_build", "oracle_cli")

        def copies(p, tag):
            """p concurrent copies on <work>/<tag>_{i,s}.txt; returns [(index s, search s, wall s, log numbers)] per copy"""
            procs, t0 = [], time.time()
            for c in range(p):
                d = os.path.join(work, f"run_{tag}_{p}_{c}")
                os.makedirs(d)
                procs.append((d, subprocess.Popen([tool, "-i", os.path.join(work, tag + "_i.txt"), "-s", os.path.join(work, tag + "_s.txt"), "-o", "out",
                                                   "-l", "log", "-k", str(args.k), "-t", str(args.t)], cwd=d,
                                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
            out = []
            for d, pr in procs:
                if pr.wait() != 0:
                    raise RuntimeError("CPU baseline tool failed")
                wall = time.time() - t0
                ti = ts = None
                nums = None
                if kind == "reference":      # its own clock() around index_reads / search_reads (index_and_search.cpp:252-300)
                    lines = open(os.path.join(d, "log", "s1_in_s0.log")).read().split("\n")
                    ti, ts = float(lines[0].split()[2]), float(lines[1].split()[2])
                    nums = [int(x) for x in re.findall(r"\d+", lines[3])]
                out.append((ti, ts, wall, nums))
                shutil.rmtree(d, ignore_errors=True)
            return out

        def write_sets(tag, idx_reads, srch_ids):
            """<tag>_i.txt / <tag>_s.txt: the first idx_reads reads of set 0, the reads srch_ids of set 1"""
            v0 = np.asarray(b0[: idx_reads * L]).reshape(idx_reads, L)
            synth.write_fasta_fast(os.path.join(work, tag + "0.fa"), v0, idx_reads, L)
            v1 = np.asarray(b1).reshape(-1, L)[srch_ids]
            synth.write_fasta_fast(os.path.join(work, tag + "1.fa"), np.ascontiguousarray(v1), len(srch_ids), L)
            open(os.path.join(work, tag + "_i.txt"), "w").write(f"s0:{work}/{tag}0.fa\n")
            open(os.path.join(work, tag + "_s.txt"), "w").write(f"s1:{work}/{tag}1.fa\n")

        P = host_cores()
        # ---- light sample: the first n reads of both sets ----
        write_sets("light", n, np.arange(n))
        one = copies(1, "light")[0]
        hot1 = (one[0] + one[1]) if one[0] is not None else one[2]
        many = copies(P, "light") if P > 1 else [one]
        light_all = sum(n / ((a + b) if a is not None else w) for a, b, w, _ in many)
        light = {"value": round(light_all, 1), "value_1core": round(n / hot1, 1),
                 "sample": f"first {n} reads of each set ({100.0 * n / args.reads:.0f} % of the job): filters 1.6 % full, every sampled search read a copy of an "
                           f"indexed one; 1 copy {hot1:.2f} s, {P} copies at once: slowest {max((a + b) if a is not None else w for a, b, w, _ in many):.2f} s"}
        res = {"value": light["value"], "unit": "reads/s", "cores": P, "kind": kind, "value_1core": light["value_1core"], "sample": light["sample"]}
        # ---- heavy sample: the first chunk fed whole, a strided sample of the search set ----
        max_kmer = int(1e9 / 2 ** (33 - args.k)) if args.k <= 33 else int(1e9)
        per_read = max(1, L - args.k + 1)
        idx_reads = min(args.reads, int(0.97 * max_kmer / per_read))      # (under max_kmer: ONE chunk, no dropped look-ahead read)
        if kind == "reference" and gpu_info and idx_reads >= args.reads // 4 and args.reads >= 1_000_000:
            stride = 50
            srch = np.arange(0, args.reads, stride)
            write_sets("heavy", idx_reads, srch)
            hv = copies(P, "heavy")
            scans = gpu_info["reads_scanned"]
            full_s = [a * (args.reads / max(1, nums[0])) + b * stride * (scans / args.reads) for a, b, w, nums in hv]
            res.update(value=round(sum(args.reads / x for x in full_s), 1), value_1core=None,
                       sample=(f"{P} independent copies at once (one per host core of this box's cgroup), each: the index set's first chunk fed whole "
                               f"({idx_reads} reads = {100.0 * idx_reads / args.reads:.0f} % of the set, one filter as full as the real job's first) and every {stride}th read of the "
                               f"search set ({len(srch)} reads) searched against it; the tool's own clocks (index {min(a for a, *_ in hv):.1f}-{max(a for a, *_ in hv):.1f} s, "
                               f"search {min(b for _, b, *_ in hv):.2f}-{max(b for _, b, *_ in hv):.2f} s per copy, wall {max(w for _, _, w, _ in hv):.1f} s) scaled to the WHOLE job: "
                               f"index x {args.reads}/{hv[0][3][0]}, search x {stride} x {scans}/{args.reads} read scans (the GPU run's count) = "
                               f"{min(full_s):.0f}-{max(full_s):.0f} s per copy; value = sum of the copies' whole-job rates"),
                       light_sample=light)
        return res
    finally:
        shutil.rmtree(work, ignore_errors=True)


def source_hash():
    """hash of the device-code sources: a traffic.json taken from another version of the kernels is marked stale"""
    h = hashlib.sha256()
    for sub in ("", "capi"):                      # the kernels and the launch / dispatch code (csrc/host is the HIP-free host side)
        d = os.path.join(ROOT, "commet_amd", "csrc", sub)
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".hpp")):
                h.update((sub + "/" + f).encode())
                h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def explicit_traffic(path):
    """a traffic.json given on the command line (tools/profile_bench.sh: the one just collected)"""
    table = json.load(open(path))
    rel = os.path.relpath(os.path.abspath(path), ROOT)
    return table, rel, table.get("_meta", {}).get("source_hash") != source_hash()


def measured_traffic(workload):
    """Per-kernel HBM bytes per launch from the newest committed rocprofv3 PMC profile of this very workload
    (profiles/<dir>/traffic.json, written by tools/pmc_summary.py next to the bench.json it was taken with); PMC
    counters cannot be collected from inside the timed run.  Returns (table, path, stale) or None."""
    pdir = os.path.join(ROOT, "profiles")

    def natural(name):   # r01_v10 after r01_v9
        return [int(x) if x.isdigit() else x for x in re.split(r"(\d+)", name)]

    best, cur = None, source_hash()
    for d in sorted(os.listdir(pdir), key=natural) if os.path.isdir(pdir) else []:
        tj, bj = os.path.join(pdir, d, "traffic.json"), os.path.join(pdir, d, "bench.json")
        if not (os.path.exists(tj) and os.path.exists(bj)):
            continue
        try:
            if json.load(open(bj))["config"]["workload"] != workload:
                continue
            table = json.load(open(tj))
            meta = table.get("_meta", {})
            cand = (table, f"profiles/{d}/traffic.json", meta.get("source_hash") != cur)
            if best is None or not cand[2] or best[2]:      # prefer a profile of these very sources, else the newest
                best = cand
        except Exception:
            continue
    return best


_T0 = time.perf_counter()
_WORK_DIRS = []        # scratch directories of the matrix leg (removed by whoever gives the leg up)


def progress(ranks, msg):
    """a line on stderr (rank 0): a long leg must not look hung to whoever watches the run; stdout stays the one JSON line"""
    if ranks is None or ranks.rank == 0:
        print(f"bench.py [{time.perf_counter() - _T0:7.1f} s] {msg}", file=sys.stderr, flush=True)


def parse_ragged(spec):
    """'LO-HI' -> (lo, hi), None for 'none' / ''"""
    if not spec or spec.lower() in ("none", "off", "0"):
        return None
    lo, hi = (int(x) for x in spec.split("-"))
    if not 1 <= lo <= hi:
        raise SystemExit(f"--ragged {spec}: need 1 <= LO <= HI")
    return lo, hi


def ragged_leg(args, device, fixed):
    """The configs[1] step on RAGGED sets (one GPU, after the headline; nothing of it is part of `value`): 2 synthetic sets of
    args.reads reads whose lengths are uniform in [lo, hi] — the same copy / substitution / reverse-complement / N rules as the
    fixed-length sets — k and t as the headline.  Real inputs are trimmed .fq.gz runs (the reference's README:105-107,
    fastq_file.h:139-190): what they cost per read and per base beside the fixed-length figure.  `fixed` = the headline's
    {reads_per_s, bases_per_s, index_kernel_ms, search_kernel_ms}."""
    import commet_amd
    from commet_amd import synth
    lo, hi = parse_ragged(args.ragged)
    n, k, t = args.reads, args.k, args.t
    b0, o0 = synth.synth_set_ragged(0, n, lo, hi, base_set=0)
    b1, o1 = synth.synth_set_ragged(1, n, lo, hi, base_set=0)
    bases_q = int(o1[-1])
    ctx = commet_amd.Context(k=k, t=t, device=device)
    try:
        irs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
        qrs = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
        ctx.synchronize()
        t_c = time.perf_counter()
        ctx.index_and_search(irs, [qrs])                       # (the set's first job: builds what is cached with the sets)
        ctx.synchronize()
        first_s = time.perf_counter() - t_c
        for _ in range(max(0, args.warmup - 1)):
            ctx.index_and_search(irs, [qrs])
        acc = dict(index_kernel_ms=0.0, search_ms=0.0)
        ctx.synchronize()
        t_c = time.perf_counter()
        for _ in range(args.steps):
            _, stats, info = ctx.index_and_search(irs, [qrs])
            for f in acc:
                acc[f] += info[f]
        ctx.synchronize()
        el = time.perf_counter() - t_c
        ktimes = None
        if not args.no_kernel_times:
            ctx.set_option("kernel_timing", 1)
            for _ in range(2):
                ctx.index_and_search(irs, [qrs])
            ktimes = {name: round(ms / 2, 4) for name, (cnt, ms) in sorted(ctx.kernel_times().items(), key=lambda kv: -kv[1][1])}
            ctx.set_option("kernel_timing", 0)
        out = {"workload": f"2 synthetic sets x {n} reads of {lo}-{hi} bp (uniform; {bases_q / n:.1f} bp on average), k={k} t={t}, index set 0 + search set 1, inputs resident in HBM",
               "reads_per_s": round(n * args.steps / el, 1), "bases_per_s": round(bases_q * args.steps / el, 1),
               "ms_per_step": round(el * 1e3 / args.steps, 3), "steps": args.steps,
               "index_kernel_ms": round(acc["index_kernel_ms"] / args.steps, 3), "search_kernel_ms": round(acc["search_ms"] / args.steps, 3),
               "chunks": info["n_chunks"], "kmers_indexed": info["kmers_indexed"], "reads_scanned": info["reads_scanned"], "shared": stats[0]["shared"],
               "first_job_ms": round(first_s * 1e3, 3), "query_list_bytes": qrs.cache_bytes,
               "kernel_ms_per_step": ktimes}
        if fixed:
            out["vs_fixed_length"] = {"reads_per_s": round(out["reads_per_s"] / fixed["reads_per_s"], 4),
                                      "bases_per_s": round(out["bases_per_s"] / fixed["bases_per_s"], 4),
                                      "index_kernel_ms": round(out["index_kernel_ms"] / fixed["index_kernel_ms"], 4) if fixed.get("index_kernel_ms") else None,
                                      "search_kernel_ms": round(out["search_kernel_ms"] / fixed["search_kernel_ms"], 4) if fixed.get("search_kernel_ms") else None,
                                      "fixed": fixed}
        irs.close()
        qrs.close()
        return out
    finally:
        ctx.close()


def matrix_size(args, ranks, root):
    """reads per set of the like-for-like matrix leg: BASELINE configs[3] (10 x 50 M reads) at EVERY N, one GPU included, so that the
    per-N values are one curve; what the host cannot hold as FASTA in the scratch root is cut down (and said so).  Returns (n, note)."""
    S, L = args.matrix_sets, args.read_len
    if args.matrix_reads is not None:
        return args.matrix_reads, None
    n, note = 50_000_000, None
    workers_all = min(S, max(1, host_cores() // 2))
    free = ranks.broadcast_object(host_memory_free(root) if ranks.rank == 0 else None)
    while free is not None and n > 1_000_000 and matrix_memory_needed(S, n, L, workers_all) * 1.25 > free:
        note = f"host memory ({free / 2**30:.0f} GiB free) does not hold {S} x {n} reads as FASTA in {root}"
        n = 10_000_000 if n > 10_000_000 else n // 2
    if note:
        note += f": {n} reads per set instead"
        if ranks.rank == 0:
            print("bench.py matrix leg: " + note, file=sys.stderr)
    return n, note


def matrix_leg(args, ranks, n, note=None, fatal_hook=None, ragged=None):
    """The full S x S matrix of S synthetic sets of n reads through the resident N x N driver, split over the ranks; sets written
    as FASTA to scratch (each rank generates its share), filter + load + jobs all timed by the driver.
    ragged = (lo, hi): read lengths uniform in [lo, hi] instead of args.read_len."""
    from commet_amd import matrix, synth
    root = os.environ.get("COMMET_SCRATCH") or ("/dev/shm" if os.access("/dev/shm", os.W_OK) else tempfile.gettempdir())
    # rank 0 makes the work directory (mkdtemp: a fresh name, mode 0700 — /dev/shm is shared with other users)
    work = ranks.broadcast_object(tempfile.mkdtemp(prefix="commet_bench_", dir=root) if ranks.rank == 0 else None)
    _WORK_DIRS.append(work)
    S, L = args.matrix_sets, args.read_len
    which = {(10, 10_000_000): "BASELINE configs[2]", (10, 50_000_000): "BASELINE configs[3]"}.get((S, n), "custom size")
    if ragged:
        which += f", RAGGED reads of {ragged[0]}-{ragged[1]} bp"
    progress(ranks, f"matrix leg: {S} sets x {n} reads ({which}) over {ranks.world} rank(s): writing the FASTA files under {work}")
    saved_scratch = os.environ.get("COMMET_SCRATCH")
    os.environ["COMMET_SCRATCH"] = work          # the driver's own scratch (descriptors, packed images) inside the work directory:
    try:                                          # whoever removes `work` removes everything this leg ever wrote
        t0 = time.perf_counter()
        if ragged:
            mine = [(s, n, ragged[0], ragged[1], os.path.join(work, f"set{s}.fa")) for s in range(S) if s % ranks.world == ranks.rank]
        else:
            mine = [(s, n, L, os.path.join(work, f"set{s}.fa")) for s in range(S) if s % ranks.world == ranks.rank]
        if mine:
            import multiprocessing as mp
            # (a 50 M-read set is ~12 GB of generator state per worker process)
            with mp.get_context("spawn").Pool(min(len(mine), max(1, host_cores() // (2 * ranks.world) if ranks.world > 1 else host_cores() // 2))) as pool:
                pool.map(synth.write_set_fasta_ragged if ragged else synth.write_set_fasta, mine, chunksize=1)
        if ranks.rank == 0:
            with open(os.path.join(work, "sets.txt"), "w") as fh:
                for s in range(S):
                    fh.write(f"S{s}: {work}/set{s}.fa\n")
        ranks.barrier()
        gen_s = time.perf_counter() - t0
        progress(ranks, f"matrix leg: files written in {gen_s:.1f} s; filter + load + jobs")
        res = matrix.run(os.path.join(work, "sets.txt"), os.path.join(work, "out") + "/", k=args.k, t=args.t, ranks=ranks, verbose=False,
                         fatal_hook=fatal_hook,
                         progress=lambda msg: print(f"bench.py [{time.perf_counter() - _T0:7.1f} s] matrix leg, rank {ranks.rank}: {msg}",
                                                    file=sys.stderr, flush=True))
    except BaseException as ex:
        ranks.abort(f"{type(ex).__name__}: {ex}")      # (the other ranks' waits end with an error naming this one; no-op on one rank)
        raise
    finally:
        if saved_scratch is None:
            os.environ.pop("COMMET_SCRATCH", None)
        else:
            os.environ["COMMET_SCRATCH"] = saved_scratch
        if not ranks.failed:
            ranks.barrier()
        if ranks.failed and ranks.world > 1:
            time.sleep(3.0)                             # (the other ranks notice within a second and are out of their jobs by then)
        t_rm = time.perf_counter()
        if ranks.rank == 0 or ranks.failed:             # (a failed job: whoever gets here removes what is left)
            shutil.rmtree(work, ignore_errors=True)
        _WORK_DIRS.remove(work)
        cleanup_s = time.perf_counter() - t_rm
    # (the seconds between the last job and this line are the removal of the leg's FASTA files — 56 GB at configs[3] — from the scratch
    # root, outside total_s: the files are the bench's, not the driver's)
    progress(ranks, f"matrix leg: done (work directory removed in {cleanup_s:.1f} s)")
    if res is None:
        return None
    keep = ("filter_s", "load_s", "jobs_s", "set_wait_s", "total_s", "reads_searched", "reads_per_s", "reads_per_s_incl_load_and_filter", "world",
            "filter_overlaps_load", "load_overlaps_jobs")
    out = {f: (round(res[f], 4) if isinstance(res[f], float) else res[f]) for f in keep}
    per_rank = [{f: (round(v, 4) if isinstance(v, float) else v) for f, v in p.items()} for p in res["per_rank"]]
    busy = [p["jobs_s"] + p.get("set_wait_s", 0.0) for p in per_rank]
    out.update(workload=f"{S} synthetic sets x {n} x {f'{ragged[0]}-{ragged[1]}' if ragged else L} bp reads, full {S} x {S} matrix ({which}) over {ranks.world} GPU(s): "
                        f"filter_reads + parse/upload + {S * S - 1} Commet.py jobs' worth of work",
               generate_s=round(gen_s, 2), size_note=note,
               handover=sorted({p.get("handover") for p in per_rank}) if ranks.world > 1 else None,
               # how evenly the static cut of the pairs loaded the ranks: slowest / mean of the ranks' job time (1.0 = even),
               # and what the cut predicted for every rank (its share of the pairs' cost) beside what it took
               imbalance=round(max(busy) / (sum(busy) / len(busy)), 4) if busy and sum(busy) > 0 else None,
               # what the DRIVER charged for device memory during the leg (commet_device_alloc_stats): host time inside hipMalloc on the
               # slowest rank, bytes asked fresh from the driver over all ranks — box tax, not kernel time
               alloc_wait_ms=max((p.get("alloc_wait_ms", 0.0) for p in per_rank), default=None),
               fresh_device_bytes=sum(p.get("fresh_device_bytes", 0) for p in per_rank),
               device_ms=max((p.get("device_ms", 0.0) for p in per_rank), default=None),
               remove_work_dir_s=round(cleanup_s, 2),
               predicted_vs_actual_share=[{"rank": p["rank"], "predicted": p.get("predicted_share"),
                                           "actual": round(b / sum(busy), 4) if sum(busy) > 0 else None} for p, b in zip(per_rank, busy)],
               per_rank=per_rank)
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if os.environ.get("BENCH_DUMP_STACKS_S"):          # debugging aid: every thread's Python stack on stderr after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_DUMP_STACKS_S"]), repeat=False, file=sys.stderr)
    from commet_amd import sharding
    ranks = sharding.Ranks()   # host-side barrier / gather / MAX over a TCP store of rank 0: no torch in the rank processes
    world, rank, local_rank = ranks.world, ranks.rank, ranks.local_rank
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); n_gpus reports {world}", file=sys.stderr)
    if args.rendezvous_only:
        elapsed = sharding.timed_region(ranks, lambda: None, lambda: time.sleep(0.01 * (rank + 1)), args.steps)
        # (no HIP call here: BENCH_FAKE_DEVICE_COUNT stands in for commet_device_count() in the CPU test of the launch path)
        devices = ranks.gather_objects(sharding.pick_device(local_rank, int(os.environ.get("BENCH_FAKE_DEVICE_COUNT", "0")) or None))
        torch_in = ranks.gather_objects("torch" in sys.modules)
        if rank == 0:
            print(json.dumps({"metric": "rendezvous only (no GPU work)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(elapsed * 1000.0 / args.steps, 3), "devices": devices,
                              "self_launched": os.environ.get("COMMET_SELF_LAUNCHED") == "1", "ranks_backend": ranks.backend,
                              "torch_in_ranks": any(torch_in)}), flush=True)
        ranks.close()
        return

    if "COMMET_FORCE_DEVICE" in os.environ and world > 1:
        # a rehearsal with several ranks on ONE device: what every rank's library keeps of the memory it frees (half the device by
        # default) must fit the device together — no rank can take memory back from another one's cache
        os.environ.setdefault("COMMET_DEVMEM_CACHE_GB", str(max(1, 96 // world)))

    import numpy as np  # noqa: F401
    import commet_amd
    from commet_amd import synth

    n, L, k, t = args.reads, args.read_len, args.k, args.t
    # every rank owns one (i, j) job of the N x N matrix: sets (2r, 2r+1)
    ragged_hl = parse_ragged(args.ragged) if args.ragged_only else None
    if ragged_hl:
        b0, o0 = synth.synth_set_ragged(2 * rank, n, ragged_hl[0], ragged_hl[1], base_set=2 * rank)
        b1, o1 = synth.synth_set_ragged(2 * rank + 1, n, ragged_hl[0], ragged_hl[1], base_set=2 * rank)
    elif args.skew > 0:
        b0, o0 = synth.synth_set_skewed(2 * rank, n, L, args.skew, base_set=2 * rank)
        b1, o1 = synth.synth_set_skewed(2 * rank + 1, n, L, args.skew, base_set=2 * rank)
    else:
        b0, o0 = synth.synth_set(2 * rank, n, L, base_set=2 * rank)
        b1, o1 = synth.synth_set(2 * rank + 1, n, L, base_set=2 * rank)

    # a launcher may give every rank ONE visible device (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank): LOCAL_RANK modulo
    # what this process sees.  COMMET_FORCE_DEVICE: debugging aid to run several ranks on one GPU (never set by the driver)
    device = sharding.pick_device(local_rank, commet_amd.device_count())
    ctx = commet_amd.Context(k=k, t=t, device=device)
    t_up = time.perf_counter()
    irs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])   # host threads 2-bit pack, planes cross PCIe (reported, never part of `value`)
    ctx.synchronize()
    upload_first_s = time.perf_counter() - t_up            # includes pinning the staging buffers of this context
    qrs = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
    ctx.synchronize()
    upload_s = time.perf_counter() - t_up

    # the very first job of a context also allocates its workspaces: reported (cold_context_first_job_ms), never timed
    cold_context_first_job_s = None
    for w in range(args.warmup):
        t_c = time.perf_counter()
        ctx.index_and_search(irs, [qrs])
        if w == 0:
            ctx.synchronize()
            cold_context_first_job_s = time.perf_counter() - t_c

    acc = dict(index_kernel_ms=0.0, search_ms=0.0, zero_ms=0.0, index_launches=0, search_launches=0)
    last = {}

    def step():
        tags, stats, info = ctx.index_and_search(irs, [qrs])
        for f in acc:
            acc[f] += info[f]
        last["stats"], last["info"], last["tags"] = stats, info, tags

    progress(ranks, f"sets resident ({n} reads each), {args.warmup} warm-up step(s) done; timing {args.steps} step(s)")
    elapsed = sharding.timed_region(ranks, ctx.synchronize, step, args.steps)
    progress(ranks, f"timed region: {elapsed * 1000.0 / args.steps:.3f} ms per step")
    stats, info = last["stats"], last["info"]

    # the first job on a SET also builds what is cached with the set (the tiled search's query list): not part of the
    # steady-state `value`, reported beside it (warm context) —
    # measured on FRESH copies of the search set (made outside the clock), the fastest counts.  Two copies that stay alive, as in rounds
    # 4-5 (each job asks the driver for its list's 2.3 GB), then — the first copy closed, its blocks filed by the library (dm_free) — a
    # third whose job asks the DRIVER for nothing: on a box that charges a process for its first use of device memory (35 ms per GiB
    # seen in round 6) the first two time that charge, the third the job
    query_list_bytes = qrs.cache_bytes
    first_job_s, first_job_all, fresh_sets = None, [], []
    for i in range(3):
        if i == 2:
            fresh_sets.pop(0).close()
        fresh_sets.append(commet_amd.ReadSet.from_files(ctx, [(b1, o1)]))
        ctx.synchronize()
        t_c = time.perf_counter()
        ctx.index_and_search(irs, [fresh_sets[-1]])
        ctx.synchronize()
        dt = time.perf_counter() - t_c
        first_job_all.append(round(dt * 1e3, 3))
        first_job_s = dt if first_job_s is None else min(first_job_s, dt)
    for fs in fresh_sets:
        fs.close()

    # ---- untimed extras (rank 0): P_ref, per-kernel times, the random-gather ceiling ---------------------------
    probes, ktimes, gather_ceiling = None, None, None
    if rank == 0:
        if not args.no_probe_count and info["n_chunks"] <= 256:   # (the counting builds replay every chunk one by one: minutes at configs[4]'s 10 421 chunks)
            ctx.set_option("count_probes", 1)     # P_ref of the reference's control flow, detail only
            _, _, inf = ctx.index_and_search(irs, [qrs])
            probes = inf["probes"]
            ctx.set_option("count_probes", 0)
        if not args.no_kernel_times:
            KT_STEPS = max(1, args.kt_steps)
            ctx.set_option("kernel_timing", 1)    # hipEvent pair around every launch, on the launching stream
            for _ in range(KT_STEPS):
                ctx.index_and_search(irs, [qrs])
            ktimes = {name: dict(launches_per_step=cnt / KT_STEPS, avg_launch_ms=ms / max(cnt, 1), ms_per_step=ms / KT_STEPS)
                      for name, (cnt, ms) in ctx.kernel_times().items()}
            ctx.set_option("kernel_timing", 0)
            acc_n = 1 << 31
            gather_ceiling = acc_n / (ctx.membench(0, ctx_filter_bytes(k), acc_n) * 1e-3)    # random 4-B gathers per second

    irs.close()
    qrs.close()
    ctx.close()

    # ---- the same step on ragged sets (one GPU; nothing of it is part of `value`) ----------------------------------
    ragged_detail = None
    if world == 1 and parse_ragged(args.ragged) and not args.ragged_only:
        progress(ranks, f"ragged leg: 2 sets x {n} reads of {args.ragged} bp")
        try:
            ragged_detail = ragged_leg(args, device, {"reads_per_s": round(n * args.steps / elapsed, 1), "bases_per_s": round(float(o1[-1]) * args.steps / elapsed, 1),
                                                      "index_kernel_ms": round(acc["index_kernel_ms"] / args.steps, 3),
                                                      "search_kernel_ms": round(acc["search_ms"] / args.steps, 3)})
            progress(ranks, f"ragged leg: {ragged_detail['ms_per_step']:.3f} ms per step, {ragged_detail['vs_fixed_length']['bases_per_s']:.3f} x the fixed-length bases/s")
        except Exception as ex:   # the headline measured above must not be lost with this extra leg
            import traceback
            traceback.print_exc()
            ragged_detail = {"error": f"{type(ex).__name__}: {ex}"}

    import threading
    emitted = threading.Event()

    def emit(matrix_detail, matrix_c2=None):
        """rank 0: the one JSON line (once)"""
        if rank != 0 or emitted.is_set():
            return
        emitted.set()
        if True:
            steps = args.steps
            ms_per_step = elapsed * 1000.0 / steps
            value = world * n * steps / elapsed
            which = {(10_000_000, 100, 32, 2): "BASELINE configs[1]", (20_000_000, 150, 21, 5): "BASELINE configs[4]"}.get((n, L, k, t), "custom size")
            workload = (f"2 synthetic sets x {n} x {L} bp reads, k={k} t={t}, index set 0 + search set 1 "
                        f"per GPU ({which}), inputs resident in HBM")
            if args.skew > 0:
                workload += f"; {100 * args.skew:g} % of every set's reads low-complexity / repeated (poly-A, tandem repeats, shared 1000-read library)"
            if ragged_hl:
                workload = (f"2 synthetic sets x {n} RAGGED reads of {ragged_hl[0]}-{ragged_hl[1]} bp (uniform), k={k} t={t}, index set 0 + search set 1 "
                            f"per GPU (custom: --ragged-only), inputs resident in HBM")
            kmers = info["kmers_indexed"]
            idx_ms = acc["index_kernel_ms"] / steps
            srch_ms = acc["search_ms"] / steps
            roofline = None
            if ktimes:
                tr = explicit_traffic(args.traffic) if args.traffic else measured_traffic(workload)
                table, tr_path, stale = tr if tr else ({}, None, None)
                step_dev_ms = sum(e["ms_per_step"] for e in ktimes.values())
                dom = max(ktimes, key=lambda nme: ktimes[nme]["ms_per_step"])
                e = ktimes[dom]

                def hbm_bytes(nme):
                    x = table.get(nme)
                    return x.get("hbm_bytes_per_launch") if x else None

                tb = hbm_bytes(dom)
                achieved = tb / (e["avg_launch_ms"] * 1e-3) / 1e9 if tb else None
                step_bytes, covered = 0.0, True
                for nme, ee in ktimes.items():
                    x = hbm_bytes(nme)
                    if x is None:
                        covered = covered and ee["ms_per_step"] < 0.01 * step_dev_ms     # tiny kernels may be missing from the profile
                    else:
                        step_bytes += x * ee["launches_per_step"]
                fetch = (table.get(dom) or {}).get("fetch_bytes_per_launch")
                roofline = {
                    "bound": "hbm", "kernel": dom, "unit": "GB/s", "peak": HBM_PEAK_GBS,
                    # as executed: HBM bytes of one launch (rocprofv3 FETCH_SIZE + WRITE_SIZE, profiles/) over its live duration
                    "achieved": round(achieved, 1) if achieved else None,
                    "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                    "traffic": tb, "traffic_source": tr_path, "traffic_stale": stale,
                    "avg_launch_ms": round(e["avg_launch_ms"], 4), "launches_per_step": e["launches_per_step"],
                    "time_share": round(e["ms_per_step"] / step_dev_ms, 4),
                    "request_rate": None, "whole_step": None,
                    "kernels": {nme: {"ms_per_step": round(ee["ms_per_step"], 4), "launches_per_step": ee["launches_per_step"],
                                      "hbm_bytes_per_launch": hbm_bytes(nme),
                                      "frac": round(hbm_bytes(nme) / (ee["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if hbm_bytes(nme) and ee["avg_launch_ms"] > 0 else None}
                                for nme, ee in sorted(ktimes.items(), key=lambda kv: -kv[1]["ms_per_step"])},
                    "note": "durations: hipEvents around every launch in untimed extra steps (one index lane, so they add up); "
                            "bytes: rocprofv3 PMC passes of this workload (gfx950 corrections of MI355X_MICROARCH.md applied by tools/pmc_summary.py)",
                }
                if dom == "search_wide_kernel":
                    # ALGORITHMIC bytes of the row pass (SURVEY 8d's "bytes per unit x units", for this kernel's own algorithm):
                    # per read and window that can hold hit J = min(t, 3) of a strand's scan, six rows (planes A, B, C, both
                    # strands) of one bit per chunk filter; rows are filled in groups of 256 chunks
                    J = min(t, 3)
                    windows = max(0, (L - 1 - (t - J) * k) - (k - 1) + 1)
                    groups = -(-info["n_chunks"] // 256)
                    passes = -(-groups * 8 // 512)
                    row_bytes = -(-groups // passes) * 8 * 4
                    alg = n * windows * 6 * row_bytes * passes
                    # this kernel has an algorithmic byte count of its own, so the line's `achieved` / `frac` are that (the
                    # contract's definition); what the counters saw (rows start on 128-byte lines: 1312 of every 1408 bytes are
                    # asked for; the replay's single-word probes; FETCH_SIZE doubled as for every 16-byte-per-lane stream) is kept
                    roofline["as_executed"] = {"achieved": roofline["achieved"], "frac": roofline["frac"], "traffic": tb}
                    roofline["achieved"] = round(alg / (e["ms_per_step"] * 1e-3) / 1e9, 1)
                    roofline["frac"] = round(alg / (e["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                    roofline["algorithmic_bytes_per_launch"] = alg // max(1, int(e["launches_per_step"]))
                    roofline["algorithmic_model"] = (f"{n} reads x {windows} windows x 6 rows x {row_bytes} B (one bit per chunk filter, "
                                                     f"{info['n_chunks']} chunks in groups of 256) x {passes} pass(es)")
                def request_rate(nme):
                    """a gather kernel priced in 64-byte memory requests that miss L2 (FETCH_SIZE / 64) against the random-gather ceiling of this run"""
                    fx, ee = (table.get(nme) or {}).get("fetch_bytes_per_launch"), ktimes.get(nme)
                    if not (fx and ee and gather_ceiling):
                        return None
                    rps = fx / SECTOR / (ee["avg_launch_ms"] * 1e-3)
                    return {"requests_per_s": round(rps), "ceiling_per_s": round(gather_ceiling), "frac": round(rps / gather_ceiling, 4)}

                if fetch and gather_ceiling and dom.startswith(("search", "tq_")) and dom != "search_wide_kernel":   # (gather kernels: one 64-byte sector per request)
                    roofline["request_rate"] = dict(request_rate(dom), what="64-byte memory requests of the kernel (FETCH_SIZE / 64) per second against "
                                                    "commet_membench's random 4-byte gathers over a filter-sized table, measured in this run")
                # both kernels of a tiled scan: the replay's requests miss L2 (that ceiling applies); the probe's gathers are served
                # FROM L2 slice by slice, so it is priced in query-list records per second as well (what its HBM requests are: streams)
                sk = {nme: request_rate(nme) for nme in ktimes if nme.startswith(("search_", "tq_probe", "tq_replay")) and nme != "search_wide_kernel"}
                if "tq_probe_kernel" in ktimes and query_list_bytes:
                    recs = query_list_bytes // 6                                    # 4-byte address + 2-byte owner per first-hit window
                    pk = ktimes["tq_probe_kernel"]
                    sk["tq_probe_kernel"] = dict(sk.get("tq_probe_kernel") or {}, l2_gathers_per_s=round(recs * pk["launches_per_step"] / (pk["ms_per_step"] * 1e-3)),
                                                 records=recs)
                roofline["search_kernels_request_rate"] = {nme: v for nme, v in sk.items() if v}
                # what a step cannot avoid moving: both packed sets read once (12 bytes per 32 bases + a triple per read), every
                # chunk's filter written once and read once (2^(k-1) bytes each way), the tag bits
                triple_bytes = 12 * (L // 32 + 1)
                compulsory = (info["reads_indexed"] + n) * triple_bytes + info["n_chunks"] * 2 * (1 << (k - 1)) + n // 8
                roofline["compulsory_bytes"] = compulsory
                if step_bytes and covered:
                    roofline["whole_step"] = {"traffic": round(step_bytes), "device_ms": round(step_dev_ms, 3),
                                              "GBps": round(step_bytes / (step_dev_ms * 1e-3) / 1e9, 1),
                                              "frac": round(step_bytes / (step_dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                    roofline["traffic_over_compulsory"] = round(step_bytes / compulsory, 2)
            # the contract's figure in REFERENCE probes, kept as detail: one request of ours answers several of them
            idx_bytes_step = info["reads_indexed"] * (L / 4.0) + 4.0 * kmers * 2 * SECTOR
            srch_bytes_step = info["reads_scanned"] * (L / 4.0 + 1 / 8.0) + probes * SECTOR if probes is not None else None
            out = {
                "metric": "reads/sec searched (index_and_search, k=%d)" % k,
                "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                # what `value` is, and the figure for ONE pair met for the first time, beside it
                "value_note": "steady state of the N x N use (a set is searched N-1 times): the search set's query list — derived from the set and (k, t) alone — "
                              "is cached with the resident set; the index set's filters are rebuilt in every step.  first_job_reads_per_s = one job on a search "
                              "set never scanned before (list build included, warm context)",
                "first_job_reads_per_s": round(world * n / first_job_s, 1),
                "dtype": "u32" if k <= 32 else "u64", "data": "synthetic",
                "config": {"workload": workload,
                           "reads_per_set": n, "read_len": L, "k": k, "t": t, "jobs": world,
                           "parallelism": f"{world} independent (i,j) jobs, no collective"},
                "roofline": roofline,
                "detail": {"chunks": info["n_chunks"], "kmers_indexed": kmers, "reads_scanned": info["reads_scanned"],
                           "shared": stats[0]["shared"], "searched_last_pass": stats[0]["searched"],
                           "index_kernel_ms": round(idx_ms, 3), "search_kernel_ms": round(srch_ms, 3),
                           "filter_zero_ms": round(acc["zero_ms"] / steps, 3),
                           "p_ref_probes": probes,
                           "reference_model_bytes_per_step": {"index": round(idx_bytes_step), "search": round(srch_bytes_step) if srch_bytes_step else None,
                                                              "note": "SURVEY 8d's sector model of the REFERENCE's accesses; not what this implementation moves"},
                           # a job on a search set that was never scanned (warm context): builds the set's query list as well
                           "first_job_ms": round(first_job_s * 1e3, 3), "first_job_reads_per_s": round(n / first_job_s, 1),
                           "first_job_ms_all": first_job_all,
                           "query_list_bytes": query_list_bytes,
                           "cold_context_first_job_ms": round(cold_context_first_job_s * 1e3, 3) if cold_context_first_job_s else None,
                           "upload_and_pack_s": round(upload_s, 3), "upload_second_set_s": round(upload_s - upload_first_s, 3),
                           "end_to_end_reads_per_s_incl_pcie": round(n / (upload_s + elapsed / steps), 1),
                           "end_to_end_reads_per_s_incl_pcie_warm_staging": round(n / (2 * (upload_s - upload_first_s) + elapsed / steps), 1),
                           "ragged": ragged_detail,
                           "matrix": matrix_detail, "matrix_configs2": matrix_c2},
            }
            if matrix_detail and "error" in matrix_detail:
                # the leg failed: say so where a reader of the per-N lines looks (a linear weak-scaling `value` with no matrix object
                # beside it would read as a clean run); detail.matrix carries the same
                out["matrix"] = {"error": matrix_detail["error"], "world": world}
            elif matrix_detail:
                # the N x N matrix through the resident driver, everything included — the figure the 1 -> 8 GPU curve is about
                # (the same workload at every N — BASELINE configs[3] when the host holds it — so the per-N values are one curve)
                out["matrix"] = {f: matrix_detail[f] for f in ("workload", "world", "reads_per_s_incl_load_and_filter", "reads_per_s", "total_s",
                                                               "jobs_s", "set_wait_s", "alloc_wait_ms", "fresh_device_bytes", "handover", "imbalance", "predicted_vs_actual_share", "size_note")}
            if world == 1:
                out["cpu_baseline"] = cpu_baseline(args, b0, b1, info) if not ragged_hl else None
                if out["cpu_baseline"] is not None and not args.cpu_full:
                    # the whole job was run once on one core (--cpu-full, minutes): quoted from the committed record of that run
                    for rec_dir in ("r06_cpu_full", "r04_cpu_full"):       # (the newest committed record of this very workload)
                        try:
                            rec = json.load(open(os.path.join(ROOT, "profiles", rec_dir, "bench.json")))
                            fj = rec["cpu_baseline"]["full_job"]
                            if rec["config"]["workload"] == workload and "error" not in fj:
                                out["cpu_baseline"]["full_job_recorded"] = dict(fj, source=f"profiles/{rec_dir}/bench.json")
                                out["cpu_baseline"]["sample"] += (f"; the WHOLE job on one core, measured once with --cpu-full (profiles/{rec_dir}): "
                                                                  f"{fj['reads_per_s']:.0f} reads/s (index {fj['index_s']} s + search {fj['search_s']} s), "
                                                                  f".bv bytes equal the GPU's: {fj['bv_bytes_equal_gpu']}")
                                break
                        except Exception:
                            pass
                if args.cpu_full and out["cpu_baseline"] is not None:
                    out["cpu_baseline"]["full_job"] = cpu_full_job(args, b0, b1, last["tags"][0], stats[0])
                    fj = out["cpu_baseline"]["full_job"]
                    if "error" not in fj:
                        out["cpu_baseline"]["sample"] += (f"; the WHOLE job once on one core (--cpu-full): {fj['reads_per_s']:.0f} reads/s "
                                                          f"(index {fj['index_s']} s + search {fj['search_s']} s), .bv bytes equal the GPU's: {fj['bv_bytes_equal_gpu']}")
            print(json.dumps(out), flush=True)

    matrix_detail, matrix_c2, failed_here = None, None, False
    if not args.no_matrix and args.matrix_sets >= 2:
        # The extra legs must never cost the headline: if they are not done after BENCH_MATRIX_LIMIT_S (default 900 s; configs[3] takes
        # ~60 s on one GPU, most of it writing the FASTA files) every rank gives up — rank 0 prints the line without them — ends its
        # child processes, removes what it wrote and leaves NON-ZERO: a GPU process that had to be abandoned is not a success.
        limit = float(os.environ.get("BENCH_MATRIX_LIMIT_S", "900"))

        def bail():
            progress(ranks, f"matrix leg: not done after {limit:.0f} s; the line goes out without it")
            try:
                emit({"error": f"matrix leg not done after {limit:.0f} s (abandoned)"}, matrix_c2)
            finally:
                sys.stdout.flush()
                try:                                  # filter_reads, generator and canary processes of THIS rank, by their pids
                    import psutil
                    for ch in psutil.Process().children(recursive=True):
                        try:
                            ch.kill()
                        except psutil.Error:
                            pass
                except Exception:
                    pass
                for w in list(_WORK_DIRS):            # (every rank: the directory holds this rank's packed images, too)
                    shutil.rmtree(w, ignore_errors=True)
                os._exit(3)

        watch = threading.Timer(limit, bail)
        watch.daemon = True
        watch.start()
        try:
            root = os.environ.get("COMMET_SCRATCH") or ("/dev/shm" if os.access("/dev/shm", os.W_OK) else tempfile.gettempdir())
            n_m, note = matrix_size(args, ranks, root)
            if world == 1 and args.matrix_reads is None and n_m > 10_000_000:
                # one GPU: BASELINE configs[2] (10 x 10 M reads) as well, a second object beside the like-for-like one
                try:
                    matrix_c2 = matrix_leg(args, ranks, 10_000_000)
                except Exception as ex:
                    matrix_c2 = {"error": f"{type(ex).__name__}: {ex}"}
            if world == 1 and parse_ragged(args.ragged) and isinstance(ragged_detail, dict) and "error" not in ragged_detail:
                # ... and the same 10-set matrix on ragged sets (detail.ragged.matrix), beside the fixed-length leg it is to be held against
                n_r = 10_000_000 if args.matrix_reads is None else args.matrix_reads
                try:
                    mr = matrix_leg(args, ranks, n_r, ragged=parse_ragged(args.ragged))
                    if mr and matrix_c2 and "error" not in matrix_c2 and args.matrix_reads is None:
                        mr["vs_fixed_length_total_s"] = round(mr["total_s"] / matrix_c2["total_s"], 4)
                    ragged_detail["matrix"] = mr
                except Exception as ex:
                    ragged_detail["matrix"] = {"error": f"{type(ex).__name__}: {ex}"}
            # (a rank whose import of another rank's set hangs ends itself from a watchdog thread: rank 0 prints the line first)
            matrix_detail = matrix_leg(args, ranks, n_m, note, fatal_hook=lambda msg: (emit({"error": msg}, matrix_c2), sys.stdout.flush()))
            if (world == 1 and args.matrix_reads is None and n_m > 10_000_000 and parse_ragged(args.ragged) and isinstance(ragged_detail, dict)
                    and "error" not in ragged_detail and matrix_detail and "error" not in matrix_detail):
                # ... and the like-for-like matrix on ragged sets, too (detail.ragged.matrix_configs3): sets of this size have no query lists,
                # their passes are the gather kernels' — where reads of many lengths cost more than their windows (MEASUREMENTS.md)
                try:
                    mr3 = matrix_leg(args, ranks, n_m, ragged=parse_ragged(args.ragged))
                    if mr3:
                        mr3["vs_fixed_length_total_s"] = round(mr3["total_s"] / matrix_detail["total_s"], 4)
                    ragged_detail["matrix_configs3"] = mr3
                except Exception as ex:
                    ragged_detail["matrix_configs3"] = {"error": f"{type(ex).__name__}: {ex}"}
        except Exception as ex:   # the headline measured above must not be lost with this extra leg
            import traceback
            traceback.print_exc()
            matrix_detail = {"error": f"{type(ex).__name__}: {ex}"}
            # did the failure begin HERE, or is this the wake of another rank's (its waits ended with "rendezvous: rank X gave up ...")?
            failed_here = not (isinstance(ex, RuntimeError) and "rendezvous" in str(ex))
        finally:
            watch.cancel()

    emit(matrix_detail, matrix_c2)

    if world > 1 and ranks.failed:
        # The matrix leg lost a rank: its error is in the line, top level ("matrix": {"error": ...}) and detail.  A thread of this
        # process may still sit in a HIP call that never returns (an import from a rank that has left), so no teardown.  Exit codes:
        # rank 0 has printed the intact headline and leaves with 0; the rank the failure began in leaves NON-ZERO (3), a few seconds
        # later so that a launcher that ends the group on the first failure (torch.distributed.run) does not end rank 0 in
        # the middle of its line; the ranks that only saw another one fail leave with 0.
        sys.stdout.flush()
        sys.stderr.flush()
        if rank != 0 and failed_here:
            time.sleep(float(os.environ.get("BENCH_FAILED_RANK_LINGER_S", "5")))
            os._exit(3)
        os._exit(0)
    ranks.close()


def ctx_filter_bytes(k):
    """table size of the gather microbenchmark: the filter's own size, at least 128 MiB (well past L2)"""
    return max(1 << (k - 1), 128 << 20)


if __name__ == "__main__":
    main()
