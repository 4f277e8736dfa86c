"""N x N comparison driver with set residency — the MI355X counterpart of Commet.py's
local mode (reference: Commet.py:438-598), SURVEY 8f-2 / 8e.

Commet.py runs N^2-1 `index_and_search` processes one after the other; every one re-parses
its FASTA files and re-creates its filter, which on a GPU means a HIP start-up, a re-upload
and a 16 GB scratch allocation per job.  Here the job DAG runs in-process through the C ABI
on sets that stay packed in HBM:

    for ref < i :   J1  index S_ref                          search S_i      -> T1
                    J2  index S_i   restricted to T1         search S_ref    -> <G>_in_<S_i>.bv
                    J3  index S_ref restricted to J2's bits  search S_i      -> <F>_in_<S_ref>.bv

Order on a rank: per reference set J1 (one call: its index built once for all its targets), then the J2 jobs of those targets in ONE
call — they all search S_ref, and commet_index_many_and_search lets the chunk filters of up to four of them share a pass over it —
and, once every reference set is through, the J3 jobs target by target (they all search S_i) the same way.  What a job writes does
not depend on when it runs.

One process per GPU.  The path has no exchange step, so ranks share nothing but small files
and three host-side gathers (sharding.Ranks: a TCP store of rank 0, no torch in the ranks):
  * every set is PARSED ONCE on the node (set s by rank s % world; left-over sets by the ranks with
    the cheapest pairs), by a rank which exports the set's
    device buffers (commet_readset_export: HIP IPC handles, a 264-byte descriptor in a scratch
    directory); the other ranks that need the set copy it device to device
    (commet_readset_import: xGMI between GPUs) — no file, no parsing.  Where that is not to be
    had (the probe or the canary below fail, COMMET_MATRIX_IPC=0) the set travels as a packed
    image in /dev/shm (commet_readset_save / _load);
  * the row-major list of (ref, i) pairs is cut into contiguous runs of equal cost, one per
    rank: a rank works on few reference sets, builds J1's index of S_ref once for all its
    targets (as Commet.py's J1 does), and loads only the sets its pairs touch;
  * a rank that fails exits non-zero at once (the launcher then ends the group).

Same inputs and outputs as Commet.py: the set file `name: file[,bv]; file…`, the filter
step (`filter_reads`, skipped when bvs are given), `OUT/<file>_in_<set>.bv`,
`OUT/<s>_in_<i>.log`, `matrix_plain.csv`, `matrix_percentage.csv`, `matrix_normalized.csv`.

  python -m commet_amd.matrix sets.txt -k 32 -t 2 -o out/            (1 GPU)
  python -m commet_amd.matrix sets.txt -k 32 -t 2 -o out/ --gpus 8   (8 GPUs: starts its own ranks)
  python -m torch.distributed.run --nproc-per-node 8 -m commet_amd.matrix sets.txt …   (any launcher that sets RANK / WORLD_SIZE / MASTER_*)
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time
import traceback

import numpy as np

from . import sharding

HERE = os.path.dirname(os.path.abspath(__file__))


# ---- the set file, as Commet.py reads it (Commet.py:42-95) -------------------------------------
def parse_set_file(path):
    names, files, bvs = [], [], []
    with open(path) as fh:
        lines = [ln for ln in fh.read().split("\n") if ln.strip()]
    has_bv = bool(lines) and "," in lines[0]                      # only the first line is inspected (Commet.py:72)
    for ln in lines:
        names.append(ln.split(":")[0].strip())
        items = ln.split(":")[1].split(";")
        files.append([it.strip().split(",")[0] for it in items])
        if has_bv:
            bvs.append([it.strip().split(",")[1] for it in items])
    return names, files, (bvs if has_bv else None)


# ---- .bv files (boolean_vector.h:302-414) -----------------------------------------------------------
def read_bv(path):
    data = open(path, "rb").read()
    h = data.index(b"#")
    nl = data.index(b"\n", h)
    n = int(data[h + 1:nl])
    raw = np.frombuffer(data[nl + 1:nl + 1 + n // 8 + 1], dtype=np.uint8)
    bits = np.zeros(n // 8 + 1, dtype=np.uint8)
    bits[:raw.size] = raw
    return n, bits


def write_bv(path, comment, n, bits):
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o600)
    with os.fdopen(fd, "wb") as fh:
        fh.write(comment.encode() + b"\n#%d\n" % n)
        fh.write(np.ascontiguousarray(bits[:n // 8 + 1], dtype=np.uint8).tobytes())


def default_filter_bv(path, read_file, n):
    """What `filter_reads <read_file> -l 0 -e 0 -o <path>` writes (filter_reads.cpp:160-176, boolean_vector.h:148-164, 302-346): with the
    default options no read can be removed, so the vector is all ones over the file's n reads (padding bits cleared) behind the tool's
    comment block — written from the parser's record count instead of a second pass over the file.  Returns the bits."""
    bits = np.full(n // 8 + 1, 0xFF, dtype=np.uint8)
    bits[-1] = (1 << (n & 7)) - 1                                   # bits n .. of the last byte (all of it when n % 8 == 0) are padding
    i = read_file.rfind("/")
    comment = ("----------------\nReference file\n  " + (read_file[i + 1:] if i > 0 else read_file) + "\nFilter Options\n"
               "  min read size     : 0\n  max number of N   : infinite\n  min shannon index : 0\n")
    if path is not None:
        write_bv(path + ".part", comment, n, bits)
        os.rename(path + ".part", path)
    return bits


def popcount(bits, n):
    return int(np.unpackbits(bits[:n // 8 + 1], bitorder="little")[:n].sum())


def concat_bits(parts):
    """[(n, bits)] of the files of a set -> set-wide (N, bits)"""
    if len(parts) == 1:
        return parts[0]
    bools = np.concatenate([np.unpackbits(b[:n // 8 + 1], bitorder="little")[:n] for n, b in parts])
    out = np.zeros(bools.size // 8 + 1, dtype=np.uint8)
    pk = np.packbits(bools, bitorder="little")
    out[:pk.size] = pk
    return bools.size, out


def split_bits(bits, counts):
    """set-wide bits -> per-file bit arrays (each n/8+1 bytes)"""
    if len(counts) == 1:
        return [np.ascontiguousarray(bits[:counts[0] // 8 + 1])]
    total = sum(counts)
    bools = np.unpackbits(bits[:total // 8 + 1], bitorder="little")[:total]
    out, pos = [], 0
    for c in counts:
        b = np.zeros(c // 8 + 1, dtype=np.uint8)
        pk = np.packbits(bools[pos:pos + c], bitorder="little")
        b[:pk.size] = pk
        out.append(b)
        pos += c
    return out


# ---- the three matrices, formatted like Commet.py:276-317 ------------------------------------------
def write_matrices(out_dir, names, considered, shared):
    n = len(names)
    head = "".join(";" + s for s in names) + "\n"
    with open(out_dir + "matrix_plain.csv", "w") as fh:
        fh.write(head)
        for i in range(n):
            fh.write(names[i] + "".join(";" + str(shared[i][j]) for j in range(n)) + "\n")
    with open(out_dir + "matrix_percentage.csv", "w") as fh:
        fh.write(head)
        for i in range(n):
            fh.write(names[i] + "".join(";" + str(100 * shared[i][j] / float(considered[i])) for j in range(n)) + "\n")
    with open(out_dir + "matrix_normalized.csv", "w") as fh:
        fh.write(head)
        for i in range(n):
            fh.write(names[i] + "".join(
                ";" + str(100 * (shared[i][j] + shared[j][i]) / float(considered[i] + considered[j])) for j in range(n)) + "\n")


def _log(out_dir, search_name, index_name, st, index_ms, wall_s):
    with open(f"{out_dir}{search_name}_in_{index_name}.log", "w") as fh:
        fh.write(f"Index  time: {index_ms / 1000.0:g} s\nSearch time: {st['search_ms'] / 1000.0:g} s\n"
                 f"Total  time: {wall_s:g} s\n[indexed {st['indexed']}, searched {st['searched']}, shared {st['shared']}]\n")


# ---- the engine: where the reads live and the jobs run ------------------------------------------------
class HipEngine:
    """The product engine: read sets packed in HBM, jobs through libcommet_hip.so (commet_amd.api).
    There is no other engine in the package; tests inject a CPU checker through `engine_factory` to run
    the multi-rank host logic without a GPU."""

    def __init__(self, k, t, local_rank):
        import commet_amd
        self._alloc0 = commet_amd.device_alloc_stats(-1)          # (before the context: its filter and workspaces count as this run's)
        # LOCAL_RANK modulo the devices this process sees (a launcher may give every rank one visible device);
        # COMMET_FORCE_DEVICE: debugging aid to run several ranks on one GPU (never set by the driver)
        self._api = commet_amd
        self.ctx = commet_amd.Context(k=k, t=t, device=sharding.pick_device(local_rank, commet_amd.device_count()))
        self._kernel_times = os.environ.get("COMMET_MATRIX_KERNEL_TIMES", "0") == "1"
        if self._kernel_times:
            self.ctx.set_option("kernel_timing", 1)
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if "COMMET_FORCE_DEVICE" in os.environ and world > 1:
            # several ranks on ONE device (a rehearsal): the cached query lists of all of them must fit it together — no rank can
            # take memory back from another one's cache
            self.ctx.set_option("query_list_budget_mb", (64 << 10) // world)

    def parse(self, files):
        return self._api.ReadSet.from_fasta(self.ctx, files)

    def save(self, rs, path):
        rs.save(path)

    def load(self, path):
        return self._api.ReadSet.load(self.ctx, path)

    def parse_probe(self):
        """a tiny set of this rank's own, for the hand-over probe"""
        b = np.frombuffer(b"ACGTTGCAACGTACGTTTGACCAGTACGATCGATCGGCTA" * 4, dtype=np.uint8)
        o = np.arange(5, dtype=np.uint64) * np.uint64(40)
        return self._api.ReadSet.from_files(self.ctx, [(b, o)])

    def export_set(self, rs):
        """bytes another rank imports the set from, device to device (HIP IPC); rs must outlive every import"""
        return rs.export()

    def import_set(self, blob):
        return self._api.ReadSet.import_(self.ctx, blob)

    def canary_argv(self, scratch, candidates):
        """the command of the fresh child process that imports the first real set before this process does (ipc_canary.py)"""
        # (by path, not `-m`: the child's working directory need not be one from which the package can be imported)
        return [sys.executable, os.path.join(HERE, "ipc_canary.py"), str(self.ctx.device), str(self.ctx.k), str(self.ctx.t), scratch,
                ",".join(str(c) for c in candidates)]

    def same_set(self, a, b):
        """the packed images of two resident sets are the same bytes (the hand-over probe: what came over is what was sent)"""
        d = tempfile.mkdtemp(prefix="commet_probe_", dir=_scratch_root())
        try:
            a.save(os.path.join(d, "a.pk"))
            b.save(os.path.join(d, "b.pk"))
            with open(os.path.join(d, "a.pk"), "rb") as fa, open(os.path.join(d, "b.pk"), "rb") as fb:
                return fa.read() == fb.read()
        finally:
            shutil.rmtree(d, ignore_errors=True)

    def file_reads(self, rs):
        return rs.file_reads()

    def release(self, rs):
        rs.close()

    def index_and_search(self, index, searches, isel, ssels):
        return self.ctx.index_and_search(index, searches, isel, ssels)

    def index_many_and_search(self, indexes, search, isels, ssel):
        return self.ctx.index_many_and_search(indexes, search, isels, ssel)

    def list_estimate(self, rs):
        return rs.cache_estimate()

    def reserve_list(self, rs):
        rs.reserve_cache()

    def device_total(self):
        return self.ctx.device_memory()[1]

    def alloc_stats(self):
        """what this run has asked the DRIVER for so far (commet_device_alloc_stats; blocks reused from the library's own cache do
        not count): host ms inside hipMalloc, bytes, calls — a box that charges a process for its first use of device memory
        (15-30 ms per GiB on some of the pool's) shows here, between the kernels, not in any kernel's time"""
        now = self._api.device_alloc_stats(-1)
        return {f: now[f] - self._alloc0[f] for f in now}

    def kernel_times(self):
        """COMMET_MATRIX_KERNEL_TIMES=1: {kernel: [launches, ms]} of this rank's jobs (a hipEvent pair around every launch; the chunks
        of a group are then built one after the other, so the figures add up but the run is a little slower) — else None"""
        return {k: [c, round(ms, 3)] for k, (c, ms) in self.ctx.kernel_times().items()} if self._kernel_times else None

    def synchronize(self):
        self.ctx.synchronize()

    def close(self):
        self.ctx.close()

    def mismatch_error(self, msg):
        return self._api.CommetError(msg)


def _scratch_root():
    root = os.environ.get("COMMET_SCRATCH")
    if root:
        return root
    return "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()


def run(input_file, out_dir, k=33, t=2, l=0, n=-1, e=0.0, m=-1, bin_dir=None, ranks=None, verbose=True,
        engine_factory=None, progress=None, fatal_hook=None):
    """progress: optional callable(str), called on every rank at the stages of the run (a caller that keeps stdout for itself —
    bench.py — shows a long run is alive with it).
    fatal_hook: optional callable(str), called from a watchdog thread right before this process is ended with os._exit because a
    HIP call of it does not return (an import of another rank's set): the caller's last chance to say what it has to say"""
    t_start = time.perf_counter()
    own_ranks = ranks is None
    if ranks is None:
        ranks = sharding.Ranks()
    world, rank = ranks.world, ranks.rank
    if out_dir[-1] != "/":
        out_dir += "/"
    bin_dir = bin_dir or os.path.join(HERE, "bin")
    os.makedirs(out_dir, exist_ok=True)
    names, files, bvs = parse_set_file(input_file)
    N = len(names)
    say = print if (verbose and rank == 0) else (lambda *a, **kw: None)
    note = progress if progress is not None else (lambda msg: None)

    if l < k * t and l != 0:                                      # Commet.py:509-513 (l stays 0 by default)
        l = k * t
    # ---- who does what: pairs in contiguous runs of equal cost, every set parsed by one rank (sharding.assign_owners) ----
    pairs = [(ref, i) for ref in range(N - 1) for i in range(ref + 1, N)]
    size = [float(sum(os.path.getsize(f) for f in fl)) for fl in files]   # cost proxy known before any parsing
    pair_cost = [size[a] + size[b] for a, b in pairs]
    runs = sharding.assign_pairs_contiguous(pair_cost, world)
    mine = [pairs[c] for c in runs[rank]]
    needed = sorted({s for p in mine for s in p})
    owner = sharding.assign_owners(N, world, [sum(pair_cost[c] for c in runs[r]) for r in range(world)])
    owned = [s for s in range(N) if owner[s] == rank]
    needed_by_others = {s for r in range(world) if r != rank for c in runs[r] for s in pairs[c]}

    # ---- filter step (Commet.py:103-121): one filter_reads per file, run by the rank that parses the set ---------
    t_filter = time.perf_counter()
    filter_err = []
    # Commet.py's default options (-l 0 -e 0, no -n, no -m) remove no read: the filter vectors are then all ones over each file's
    # reads and are written from the parser's record counts by the set's owner (default_filter_bv: the tool's bytes, tested), every
    # rank derives the same selection from the counts of the set it holds — no second pass over the files, nothing to wait for.
    # COMMET_MATRIX_FILTER_TOOL=1 runs filter_reads all the same.
    synth_filters = bvs is None and l == 0 and e == 0 and n < 0 and m < 0 and os.environ.get("COMMET_MATRIX_FILTER_TOOL", "0") != "1"
    if synth_filters:
        bvs = [[out_dir + os.path.basename(f) + ".bv" for f in fl] for fl in files]
        filter_pool, filter_jobs, filtered_here = None, [], False
    elif bvs is None:
        bvs = [[out_dir + os.path.basename(f) + ".bv" for f in fl] for fl in files]
        todo = [(s, j) for s in range(N) for j in range(len(files[s]))]
        cmds = []
        for s, j in todo:
            if owner[s] != rank:                                   # (the set's parser filters its files, too: one producer per set)
                continue
            cmd = [os.path.join(bin_dir, "filter_reads"), files[s][j], "-l", str(l), "-e", str(e)]
            if n >= 0:
                cmd += ["-n", str(n)]
            if m >= 0:
                cmd += ["-m", str(m / len(files[s]))]
            cmd += ["-o", bvs[s][j]]
            say("Filtering command: " + " ".join(cmd))
            cmds.append(cmd)
        # independent processes (each one multi-threaded over its file), a few at a time — and beside the parsing of
        # the sets below.  A filter's .bv is written under another name and renamed into place, so the file appears complete
        # or not at all: whoever needs set s (any rank) waits for ITS files only (`prepare`), not for every filter of the node.
        from concurrent.futures import ThreadPoolExecutor
        for c in cmds:                                             # what an earlier run left in this directory must not be read as this run's
            for stale in (c[-1], c[-1] + ".part"):
                try:
                    os.remove(stale)
                except OSError:
                    pass
        ranks.barrier()                                            # (every rank's stale files are gone before anybody looks for new ones)

        def run_filter(cmd):
            subprocess.run(cmd[:-1] + [cmd[-1] + ".part"], check=True, stdout=subprocess.DEVNULL)
            os.rename(cmd[-1] + ".part", cmd[-1])

        filter_pool = ThreadPoolExecutor(max_workers=int(os.environ.get("COMMET_FILTER_JOBS", "3")))
        # (the sets are loaded last set first, see below: so are their filters)
        filter_jobs = [filter_pool.submit(run_filter, c) for c in reversed(cmds)]
        filter_end = [t_filter]

        def filter_over(f):
            filter_end[0] = max(filter_end[0], time.perf_counter())   # when the last of them was done
            if not f.cancelled() and f.exception() is not None:
                filter_err.append(f.exception())                   # whoever waits for a set (here) is told at once

        for j in filter_jobs:
            j.add_done_callback(filter_over)
        filtered_here = True
    else:
        filter_pool, filter_jobs = None, []
        filtered_here = False
    filter_s = 0.0

    def filters_done():
        """this rank's filter processes are through (raises what a filter_reads process raised)"""
        nonlocal filter_s
        if filter_pool is not None:
            try:
                for j in filter_jobs:
                    j.result()
            finally:
                filter_pool.shutdown(wait=True)
            filter_s = (filter_end[0] if filter_jobs else time.perf_counter()) - t_filter

    scratch = None
    if world > 1:
        # rank 0 makes the directory (mkdtemp: a fresh name, mode 0700 — the scratch root is shared with other users)
        scratch = ranks.broadcast_object(tempfile.mkdtemp(prefix="commet_pk_", dir=_scratch_root()) if rank == 0 else None)
    # sets are made resident by a second thread while the jobs run (COMMET_MATRIX_PIPELINE=0: everything first)
    pipelined = N >= 2 and os.environ.get("COMMET_MATRIX_PIPELINE", "1") != "0"
    eng = (engine_factory or HipEngine)(k, t, ranks.local_rank)
    # How a set parsed by one rank reaches the others: device to device over HIP IPC (the owner exports its buffers, the
    # others copy them over xGMI: no file, tens of ms for a 50 M-read set) when every rank can import a probe set of its
    # neighbour and of rank 0; else as a packed image in the scratch directory (0.5 s to write, 0.2 s to read).
    use_ipc = False

    def import_guarded(blob, limit_s=None):
        """eng.import_set with a deadline: a HIP call that hangs cannot be cancelled from inside the process, so a rank whose
        import does not return leaves (non-zero; the launcher ends the job) rather than keep its peers waiting for good.
        COMMET_IPC_LOCK=1 also takes a lock file of the node around the import (one import at a time on the node: a round-3
        precaution against two processes attaching to each other's buffers at the same moment, never needed without torch)."""
        def give_up():
            msg = f"commet_amd.matrix, rank {rank}: commet_readset_import did not return within {limit:.0f} s; leaving"
            print(msg, file=sys.stderr, flush=True)
            if fatal_hook is not None:
                try:
                    fatal_hook(msg)
                except Exception:
                    pass
            for s_ in owned:
                for ext in ("pk", "ipc"):
                    try:
                        os.remove(os.path.join(scratch, f"set{s_}.{ext}"))
                    except OSError:
                        pass
            os._exit(4)

        limit = float(limit_s if limit_s is not None else os.environ.get("COMMET_IPC_IMPORT_LIMIT_S", "120"))
        watch = threading.Timer(limit, give_up)
        watch.daemon = True
        watch.start()
        try:
            if os.environ.get("COMMET_IPC_LOCK", "0") == "1":
                import fcntl
                with open(os.path.join(scratch, "import.lock"), "a+") as lf:
                    fcntl.flock(lf, fcntl.LOCK_EX)
                    try:
                        return eng.import_set(blob)
                    finally:
                        fcntl.flock(lf, fcntl.LOCK_UN)
            return eng.import_set(blob)
        finally:
            watch.cancel()

    # The default since round 4 (COMMET_MATRIX_IPC=0: packed images).  Round 3 had to make it opt-in: an import of a 50 M-read set
    # did not return when the rank process had imported torch (for the gloo barrier) — two ROCm runtimes in one process.  The ranks
    # meet over sharding's TCP store now and hold one runtime.  Two nets stay under the large imports, which the probe below (a
    # four-read set) says nothing about: the first REAL set is imported by a fresh child process first (the canary: killed when it
    # does not come back, and every rank then asks the owners for packed images), and an import of this process that does not
    # return within COMMET_IPC_IMPORT_LIMIT_S ends the rank non-zero instead of leaving the job hung.
    if world > 1 and os.environ.get("COMMET_MATRIX_IPC", "1") != "0" and hasattr(eng, "export_set"):
        probe = blob = None
        try:
            probe = eng.parse_probe()
            blob = eng.export_set(probe)
        except Exception as ex:
            say(f"device-to-device hand-over of sets not available ({ex}): packed images instead")
        blobs = ranks.gather_objects(blob)                        # (every rank, whatever happened above)
        ok = int(blob is not None)
        if ok:
            try:
                for src in sorted({0, (rank + 1) % world} - {rank}):
                    if blobs[src] is None:
                        ok = 0
                    else:
                        got = import_guarded(blobs[src], os.environ.get("COMMET_IPC_PROBE_LIMIT_S", "30"))   # (four reads: seconds are generous) every rank's probe set holds the same reads:
                        if hasattr(eng, "same_set") and not eng.same_set(got, probe):   # a copy that arrives damaged counts as no hand-over
                            say("device-to-device hand-over of sets: the probe set did not arrive intact: packed images instead")
                            ok = 0
                        eng.release(got)
            except Exception as ex:
                say(f"device-to-device hand-over of sets not available ({ex}): packed images instead")
                ok = 0
        use_ipc = ranks.sum_int(ok) == world                      # (also: every import of the probes is done)
        if probe is not None:
            eng.release(probe)
    note(f"{N} sets, {len(mine)} of {len(pairs)} pairs on this rank; sets are handed over " + ("device to device" if use_ipc else "as packed images" if world > 1 else "nowhere (one rank)"))
    exported = {}                                                 # sets this rank keeps alive for the others' imports
    loader, loader_stop = None, None
    prof = dict(rank=rank, pairs=len(mine), sets_parsed=0, sets_loaded=0, j1_builds=0, parse_s=0.0, save_s=0.0, load_s=0.0,
                jobs=0, call_ms=0.0, device_ms=0.0, handover="ipc" if use_ipc else "image", backend=getattr(ranks, "backend", None),
                torch_loaded="torch" in sys.modules,
                predicted_share=round(sum(pair_cost[c] for c in runs[rank]) / max(sum(pair_cost), 1e-9), 4))   # what the static cut expects of this rank
    # the canary: of the ranks that take sets from others, the first one starts a fresh child process that imports the first
    # real set to appear (tests/engines without a child command: no canary)
    def foreign(r):
        return sorted({s_ for c in runs[r] for s_ in pairs[c] if owner[s_] != r})

    canary_rank = next((r for r in range(world) if foreign(r)), None) if use_ipc else None
    canary = None
    if use_ipc and rank == canary_rank and hasattr(eng, "canary_argv") and os.environ.get("COMMET_IPC_CANARY", "1") != "0":
        canary = subprocess.Popen(eng.canary_argv(scratch, foreign(rank)), stdout=subprocess.DEVNULL)
    server, serve_stop, stop_ev = None, None, None
    try:
        # ---- residency: parse my sets once, publish their packed images, load the others I need ----------------
        t0 = time.perf_counter()
        sets = {}
        solo = pipelined and world == 1                           # (then the loading thread parses, too)

        def leave_filters(s, rs):
            """default options: the filter files of a set this rank has just parsed, from its record counts"""
            if synth_filters:
                for c_, f_, b_ in zip(eng.file_reads(rs), files[s], bvs[s]):
                    default_filter_bv(b_, f_, c_)

        def parse_own(s):
            """one of this rank's sets: parsed here and nowhere else; its packed image published for the ranks that need it
            (commet_readset_save writes a .tmp and renames it: the file appears complete or not at all)"""
            w0 = time.perf_counter()
            rs = eng.parse(files[s])
            prof["parse_s"] += time.perf_counter() - w0
            prof.setdefault("parse_log", []).append([s, round(time.perf_counter() - w0, 4)])
            prof["sets_parsed"] += 1
            leave_filters(s, rs)
            if s in needed_by_others:
                w0 = time.perf_counter()
                if use_ipc:                                       # a small descriptor file; the set stays alive for the importers
                    path = os.path.join(scratch, f"set{s}.ipc")
                    with open(path + ".tmp", "wb") as fh:
                        fh.write(eng.export_set(rs))
                    os.rename(path + ".tmp", path)
                    exported[s] = rs
                else:
                    eng.save(rs, os.path.join(scratch, f"set{s}.pk"))
                prof["save_s"] += time.perf_counter() - w0
            if s in needed:
                sets[s] = rs
            elif s not in exported:
                eng.release(rs)

        if not pipelined:
            for s in owned:
                if s in needed or s in needed_by_others:
                    parse_own(s)
            ranks.barrier()                                      # every image is in place
        counts, sel, considered = {}, {}, {}

        def prepare(s):
            """set s is resident; once its filter files are there (written by whichever rank filtered them): its per-file read
            counts, its input selection, the number of reads it was asked about (the matrix's diagonal)"""
            counts[s] = eng.file_reads(sets[s])
            if filtered_here:
                for b in bvs[s]:
                    if not wait_file(b, f"the filter of set {s}", where=os.path.dirname(b)):
                        return False
            if synth_filters:                                     # all ones (the set's owner has left the files: leave_filters)
                parts = [(c, default_filter_bv(None, f, c)) for c, f in zip(counts[s], files[s])]
            else:
                parts = [read_bv(b) for b in bvs[s]]
            considered[s] = sum(popcount(b, nb) for nb, b in parts)
            for (nb, _), c, f in zip(parts, counts[s], files[s]):
                if nb != c:
                    raise eng.mismatch_error(f"Number of reads in {f} and boolean vector size are not equal -> quit")
            _, sel[s] = concat_bits(parts)
            if considered[s] == sum(counts[s]) and os.environ.get("COMMET_MATRIX_KEEP_SEL", "0") != "1":
                sel[s] = None                                     # every read selected (the default filters): no bitmap to upload, dense plans
            return True

        stop_ev = threading.Event()                               # set when this rank is through (or has failed): ends every wait below
        hand = dict(ipc=use_ipc, canary=None)

        def wait_file(path, what, where=None):
            """a file another rank publishes (renamed into place: complete or absent): there once its owner has got that far
            (or never, if that rank died: the launcher then ends this process; the deadline only bounds a stray wait)"""
            deadline = time.perf_counter() + float(os.environ.get("COMMET_DIST_TIMEOUT_S", "600"))
            w0 = time.perf_counter()
            polls = 0
            while not os.path.exists(path):
                if filter_err:
                    raise filter_err[0]
                if stop_ev.is_set():
                    return False
                polls += 1
                if polls % 128 == 0 and hasattr(ranks, "check"):   # (every quarter of a second: has the rank that is to publish it given up?)
                    ranks.check()
                if time.perf_counter() > deadline:
                    raise RuntimeError(f"{what} did not appear in {where or scratch}")
                time.sleep(0.002)
            prof["image_wait_s"] = prof.get("image_wait_s", 0.0) + time.perf_counter() - w0
            return True

        def canary_verdict():
            """Did the fresh child process of `canary_rank` get the first real set across?  That rank waits for its child (and
            kills it by its pid when it does not answer in COMMET_IPC_CANARY_S), says so in the scratch directory, the others
            read it there.  True: this process imports, too."""
            if hand["canary"] is None:
                ok_path, fail_path = os.path.join(scratch, "canary.ok"), os.path.join(scratch, "canary.fail")
                if rank == canary_rank:
                    verdict = "passed"
                    if canary is not None:
                        limit = float(os.environ.get("COMMET_IPC_CANARY_S", "30"))
                        try:
                            rc = canary.wait(timeout=limit)
                            verdict = "passed" if rc == 0 else f"failed (exit code {rc})"
                        except subprocess.TimeoutExpired:
                            canary.kill()
                            try:
                                canary.wait(timeout=5)
                            except subprocess.TimeoutExpired:
                                pass
                            verdict = f"failed (no answer within {limit:.0f} s: killed)"
                    path = ok_path if verdict == "passed" else fail_path
                    with open(path + ".tmp", "w") as fh:
                        fh.write(verdict)
                    os.rename(path + ".tmp", path)
                else:
                    deadline = time.perf_counter() + float(os.environ.get("COMMET_DIST_TIMEOUT_S", "600"))
                    while not (os.path.exists(ok_path) or os.path.exists(fail_path)):
                        if stop_ev.is_set() or time.perf_counter() > deadline:
                            break
                        time.sleep(0.002)
                    verdict = "passed" if os.path.exists(ok_path) else (open(fail_path).read() if os.path.exists(fail_path) else "failed (no verdict)")
                hand["canary"] = verdict
                prof["ipc_canary"] = verdict
            return hand["canary"] == "passed"

        def fetch(s):
            """another rank's set: from its owner's device buffers, or from its packed image; False: this rank is stopping"""
            ipc_path, pk_path = os.path.join(scratch, f"set{s}.ipc"), os.path.join(scratch, f"set{s}.pk")
            if hasattr(ranks, "check"):
                ranks.check()                                     # (no import from a job that has lost a rank: its owner may be leaving)
            if hand["ipc"]:
                if not wait_file(ipc_path, f"the descriptor of set {s}"):
                    return False
                if not canary_verdict():                          # (the first set only)
                    hand["ipc"] = False
                    prof["handover"] = "image"
                    note(f"device-to-device hand-over given up (canary {hand['canary']}): packed images from here on")
            w0 = time.perf_counter()
            if hand["ipc"]:
                with open(ipc_path, "rb") as fh:
                    sets[s] = import_guarded(fh.read())
            else:
                if use_ipc:                                       # the owners published descriptors only: ask for the image
                    open(os.path.join(scratch, f"set{s}.want.{rank}"), "w").close()
                w0 = time.perf_counter()
                if not wait_file(pk_path, f"the packed image of set {s}"):
                    return False
                w0 = time.perf_counter()
                sets[s] = eng.load(pk_path)
            prof["load_s"] += time.perf_counter() - w0
            prof["sets_loaded"] += 1
            return True

        def serve_images():
            """the way back: a rank that gave the device-to-device hand-over up asks for `set<s>.pk`; its owner, which keeps
            every exported set alive, writes it"""
            served = set()
            while not serve_stop.wait(0.005):
                for s_ in list(exported):
                    if s_ not in served and any(f.startswith(f"set{s_}.want.") for f in os.listdir(scratch)):
                        w0 = time.perf_counter()
                        eng.save(exported[s_], os.path.join(scratch, f"set{s_}.pk"))
                        prof["save_s"] += time.perf_counter() - w0
                        served.add(s_)

        server, serve_stop = None, threading.Event()
        if use_ipc and any(s_ in needed_by_others for s_ in owned):
            server = threading.Thread(target=serve_images, name="commet-image-server", daemon=True)
            server.start()

        refs = sorted({p[0] for p in mine}, reverse=pipelined)    # pipelined: last reference set first
        if not pipelined:
            for s in needed:
                if s not in sets:
                    fetch(s)                                      # (its owner's descriptor / image is in place behind the barrier)
            for s in needed:
                prepare(s)
            load_s = time.perf_counter() - t0
            say(f"loaded {N} sets in {load_s:.2f} s (rank 0: {prof['sets_parsed']} parsed, {prof['sets_loaded']} from packed images)")
        else:
            # A second host thread makes the sets resident in the order the jobs want them (read sets are made on a stream
            # of their own, include/commet_hip.h) while this one runs the jobs of a reference set as soon as it and its
            # targets are there: the host-bound loading hides behind the device-bound jobs.  One rank: the thread parses
            # the files, last set first, and ref = N-2, N-3, ... need the sets ref .. N-1.  Several ranks: the thread parses
            # this rank's own sets and publishes them, then takes the others' as they appear (no barrier in between).
            ready = [threading.Event() for _ in range(N)]
            jobs_done = threading.Event()
            loader_stop = stop_ev
            load_err = []
            load_end = [t0]
            if solo:
                order = list(range(N - 1, -1, -1))
                own_first = []
            else:
                order = []
                for ref in refs:
                    for s in [ref] + [i for (r, i) in mine if r == ref]:
                        if s not in order:
                            order.append(s)
                # Several ranks: nobody waits at a barrier for every set of the node to be parsed.  This rank parses its own
                # sets first — the ones most ranks wait for first — and publishes their images; then it takes the other
                # ranks' images, in the order its jobs want them, as soon as each file appears.  Its first job starts when
                # the two sets of that job are there, whatever the other ranks are still parsing.
                wanted_by = {s_: sum(1 for r in range(world) if any(s_ in pairs[c] for c in runs[r])) for s_ in owned}
                own_first = sorted((s_ for s_ in owned if s_ in needed or s_ in needed_by_others), key=lambda s_: (-wanted_by[s_], s_))
                # (simulated and NOT adopted in round 6: parsing first the sets some rank cannot start without — in every pair of its run —
                # helps the ranks that wait for those and delays the one with the longest run: configs[3] at eight ranks 2.50 against 2.58 s
                # with one run's fitted costs, 2.65 against 2.53 s with another's: tools/schedule_sim.py, blocking_first)

            loaded_all = [False]

            def reserve_lists():
                """COMMET_MATRIX_LARGE_LISTS=1 (off by default): query lists above the library's cap (a 50 M-read set's is 11 GB; it saves
                ~12 ms of every J2 / J3 job that searches the set) for the sets this rank searches three times or more; their memory is asked
                from the driver HERE, by the loader thread once every set is resident, and a set whose memory waits in the library's device
                cache gets its list at its next eligible scan but one.  Measured on configs[3] (profiles/r05_large_lists): 11.0 s against
                11.8 s on a box whose device memory had been used before (the driver's 110 GiB take no time there), 13.4 s on a fresh box —
                there hipMalloc costs 15-30 ms per GiB (3.4 s), and while one thread is inside hipMalloc the HIP calls of every other thread
                of the process wait, so the job thread stands still with it.  Hence opt-in: for long-lived hosts."""
                if os.environ.get("COMMET_MATRIX_LARGE_LISTS", "0") != "1" or not hasattr(eng, "list_estimate"):
                    return
                scans = {}
                for (r_, i_) in mine:                            # J2 searches the reference set, J3 the target (Commet.py:220, 233)
                    scans[r_] = scans.get(r_, 0) + 1
                    scans[i_] = scans.get(i_, 0) + 1
                want = [s_ for s_ in sorted(scans, key=lambda s_: -scans[s_]) if scans[s_] >= 3 and s_ in sets]
                est = {s_: eng.list_estimate(sets[s_]) for s_ in want}
                want = [s_ for s_ in want if est[s_] > (4 << 30)]          # (smaller lists are the library's default already)
                budget = 0.4 * eng.device_total()
                got = 0
                for s_ in want:
                    if loader_stop.is_set() or jobs_done.is_set() or sum(est[x] for x in want[:want.index(s_) + 1]) > budget:
                        break
                    eng.reserve_list(sets[s_])
                    got += 1
                prof["lists_reserved"] = got
                if got:
                    note(f"memory of {got} large query lists set aside ({sum(est[x] for x in want[:got]) / 2**30:.0f} GiB)")

            def load_all():
                try:
                    for s in own_first:
                        if loader_stop.is_set():
                            return
                        parse_own(s)
                    for s in order:
                        if loader_stop.is_set():                 # the job thread has failed
                            break
                        if solo:
                            w0 = time.perf_counter()
                            sets[s] = eng.parse(files[s])
                            prof["parse_s"] += time.perf_counter() - w0
                            prof.setdefault("parse_log", []).append([s, round(time.perf_counter() - w0, 4)])
                            prof["sets_parsed"] += 1
                            leave_filters(s, sets[s])
                        elif s not in sets and not fetch(s):
                            break
                        if not prepare(s):
                            break
                        ready[s].set()
                        note(f"set {s} resident")
                    load_end[0] = time.perf_counter()
                    loaded_all[0] = True
                    reserve_lists()
                except BaseException as ex:          # handed to the job thread, which is waiting for a set
                    load_err.append(ex)
                    for ev in ready:
                        ev.set()
                finally:
                    if not loaded_all[0]:
                        load_end[0] = time.perf_counter()

            loader = threading.Thread(target=load_all, name="commet-set-loader", daemon=True)
            loader.start()
        set_wait = [0.0]

        def wait_for(s):
            if loader is not None:
                w0 = time.perf_counter()
                polls = 0
                while not ready[s].wait(0.05):
                    if filter_err:                               # a filter_reads process of this rank failed
                        raise filter_err[0]
                    polls += 1
                    if world > 1 and polls % 5 == 0 and hasattr(ranks, "check"):
                        ranks.check()                            # (has a rank given up?  Its sets will never come)
                set_wait[0] += time.perf_counter() - w0
                if load_err:
                    raise load_err[0]

        # ---- my pairs, grouped by ref ------------------------------------------------------------------------
        shared = {}                    # (from set, in set) -> reads of `from` found in `in`
        reads_searched = 0

        call_log = os.environ.get("COMMET_MATRIX_CALL_LOG")   # one line per library call: jobs, wall, event-timed device time, python clock

        job_log = prof.setdefault("job_log", [])   # one row per library call: [kind, search set or reference, [the other sets], index ms, search ms, call ms]
                                                   # (what tools/schedule_sim.py replays on the pair cut of N ranks)

        def _acc(inf, n=1, what=None):
            prof["jobs"] += n
            prof["call_ms"] += inf["total_ms"]
            prof["device_ms"] += inf["index_ms"] + inf["search_ms"]
            if what is not None:
                job_log.append([what[0], what[1], list(what[2]), round(inf["index_ms"], 3), round(inf["search_ms"], 3), round(inf["total_ms"], 3)])
            if call_log:
                with open(call_log, "a") as fh:
                    fh.write(f"{rank} {n} {inf['total_ms']:.3f} {inf['index_ms']:.3f} {inf['search_ms']:.3f} {time.perf_counter():.6f}\n")

        # the .bv and .log files of a job are written by two helper threads while the next job runs (6 MB per 50 M-read file: 3-4 ms
        # of a job's ~6 ms of host time at configs[3]); all of them are on disk before the jobs' clock stops
        from concurrent.futures import ThreadPoolExecutor
        writer, written = ThreadPoolExecutor(2), []

        def out_bv(path, comment, c, b):
            written.append(writer.submit(write_bv, path, comment, c, b))

        def out_log(*a):
            written.append(writer.submit(_log, *a))

        def jobs_on_one_search_set(index_ids, search_id, selections, kind="J2"):
            """Jobs that search the SAME set — the J2 jobs of a reference set, the J3 jobs of a target (Commet.py:220, 233) — in one call
            where the engine has one (commet_index_many_and_search: their chunk filters share passes over the search set: the lane-a
            gathers of its reads, two thirds of such a job's memory requests, are made once per pass instead of once per job);
            -> [(tags, stats, index_ms)] in the jobs' order, bit for bit what the jobs give one by one."""
            if not index_ids:
                return []
            if hasattr(eng, "index_many_and_search") and len(index_ids) > 1:
                tags, st, inf = eng.index_many_and_search([sets[x] for x in index_ids], sets[search_id], selections, sel[search_id])
                _acc(inf, len(index_ids), (kind, search_id, index_ids))
                return [(tags[j], st[j], inf["index_ms"] / len(index_ids)) for j in range(len(index_ids))]
            out = []
            for x, sl in zip(index_ids, selections):
                tags, st, inf = eng.index_and_search(sets[x], [sets[search_id]], sl, [sel[search_id]])
                _acc(inf, 1, (kind, search_id, [x]))
                out.append((tags[0], st[0], inf["index_ms"]))
            return out

        # Order of a rank's jobs: per reference set J1 (its index built once for all its targets), then the J2 jobs of its targets
        # together (they all search S_ref); the J3 jobs — (ref, i) searches S_i — are kept back and run target by target at the end, so
        # that the J3 jobs of a target share passes as well.  The files a job writes do not depend on when it runs.
        t_jobs = time.perf_counter()
        kept_T2 = {}                   # (ref, i) -> J2's result, the index selection of J3(ref, i); freed as J3 consumes it
        refs_left = {}                 # target -> reference sets of this rank's pairs that have not been through J2 yet
        for (r_, i_) in mine:
            refs_left[i_] = refs_left.get(i_, 0) + 1

        def j3_of(i):
            """J3 of every pair of target i: S_i in (S_ref restricted to J2's result) — overwrites J1's <F>_in_<S_ref>.bv (Commet.py:233)"""
            nonlocal reads_searched
            w0 = time.perf_counter()
            wait_for(i)
            of_i = [r for (r, t_) in mine if t_ == i]
            for ref, (T3, st3, index_ms) in zip(of_i, jobs_on_one_search_set(of_i, i, [kept_T2.pop((r, i)) for r in of_i], "J3")):
                for f, c, b in zip(files[i], counts[i], split_bits(T3, counts[i])):
                    out_bv(out_dir + os.path.basename(f) + "_in_" + names[ref] + ".bv", f + " in " + names[ref], c, b)
                out_log(out_dir, names[i], names[ref], st3, index_ms, time.perf_counter() - w0)
                shared[(i, ref)] = st3["shared"]
                reads_searched += considered[i]
            note(f"J3 jobs of set {i} done ({prof['jobs']} so far)")

        try:
            # Which reference set next (round 6): the first of the rank's list that is resident TOGETHER with one of its targets — a rank of a
            # node starts on whatever pair has arrived instead of waiting for the first reference set of its list (tools/schedule_sim.py on
            # configs[3]: 2.69 -> 2.53 s at eight ranks, 4.24 -> 3.97 s at four).  A reference set some of whose targets are still on their way
            # is taken up again later (one more index build of S_ref instead of an idle GPU, as before).  One rank, or everything loaded
            # first: the list's own order.
            def there(s_):
                return loader is None or ready[s_].is_set()

            left = {ref: [i for (r, i) in mine if r == ref] for ref in refs}
            while left:
                ref = next((r_ for r_ in refs if r_ in left and there(r_) and any(there(i) for i in left[r_])), None)
                if ref is None:                                      # nothing can start: until some set arrives (errors of the loader / the filters / another rank end the wait)
                    w0 = time.perf_counter()
                    polls = 0
                    while not any(there(r_) and any(there(i) for i in left[r_]) for r_ in left):
                        if filter_err:
                            raise filter_err[0]
                        if load_err:
                            raise load_err[0]
                        polls += 1
                        if world > 1 and polls % 125 == 0 and hasattr(ranks, "check"):
                            ranks.check()
                        time.sleep(0.002)
                    set_wait[0] += time.perf_counter() - w0
                    continue
                targets = [i for i in left[ref] if there(i)]
                left[ref] = [i for i in left[ref] if i not in targets]
                for s_need in [ref] + targets:
                    wait_for(s_need)                                 # (resident: raises what the loader raised, if it did)
                w0 = time.perf_counter()
                tags1, st1, inf1 = eng.index_and_search(sets[ref], [sets[i] for i in targets], sel[ref], [sel[i] for i in targets])
                prof["j1_builds"] += 1
                reads_searched += sum(considered[i] for i in targets)
                _acc(inf1, 1, ("J1", ref, targets))
                # J2 of every target: X_i = S_i restricted to (S_i in S_ref); S_ref in X_i
                for i, (T2, st2, index_ms) in zip(targets, jobs_on_one_search_set(targets, ref, list(tags1))):
                    for f, c, b in zip(files[ref], counts[ref], split_bits(T2, counts[ref])):
                        out_bv(out_dir + os.path.basename(f) + "_in_" + names[i] + ".bv", f + " in " + names[i], c, b)
                    out_log(out_dir, names[ref], names[i], st2, index_ms, time.perf_counter() - w0)
                    shared[(ref, i)] = st2["shared"]
                    kept_T2[(ref, i)] = T2
                    reads_searched += considered[ref]
                if not left[ref]:
                    del left[ref]
                    note(f"J1 and J2 jobs of set {ref} done ({prof['jobs']} so far)")
                # The J3 jobs — (ref, i) searches S_i — are kept back so that the J3 jobs of a target share passes as well, but no longer than
                # needed: a target's batch runs as soon as the last of its reference sets on this rank has been through J2 (its J2 bitmaps are
                # freed with it, its files are on disk: a late failure loses little).  The files a job writes do not depend on when it runs.
                for i in targets:
                    refs_left[i] -= 1
                for i in sorted(targets):
                    if refs_left[i] == 0:
                        j3_of(i)
            for i in sorted(i_ for i_, n_ in refs_left.items() if n_ > 0):     # (never: every target's references are in `refs`)
                j3_of(i)
        except BaseException:
            writer.shutdown(wait=False, cancel_futures=True)
            raise
        eng.synchronize()
        if loader is not None:
            jobs_done.set()                                      # (no list memory is set aside for jobs that are over)
        for f in written:                                        # (what a writer raised is raised here)
            f.result()
        writer.shutdown()
        jobs_s = time.perf_counter() - t_jobs - set_wait[0]      # (pipelined: without the waits for sets still being loaded)
        prof["jobs_s"] = jobs_s
        prof["set_wait_s"] = set_wait[0]
        if hasattr(eng, "alloc_stats"):
            a_ = eng.alloc_stats()
            prof["alloc_wait_ms"], prof["fresh_device_bytes"], prof["alloc_calls"] = round(a_["wait_ms"], 1), a_["fresh_bytes"], a_["calls"]
        if hasattr(eng, "kernel_times") and eng.kernel_times() is not None:
            prof["kernel_ms"] = eng.kernel_times()
        if loader is not None:
            for s in order:                                      # (one rank: a set no pair needs is still loaded and counted)
                wait_for(s)
            loader.join()
            load_s = load_end[0] - t0
        filters_done()                                           # (this rank's filter processes: what one of them raised is raised here)
        # ---- matrices on rank 0 -----------------------------------------------------------------------------
        everyone = ranks.gather_objects((shared, prof, considered))   # (every rank is through its jobs: nobody asks for a set any more)
        if server is not None:
            serve_stop.set()
            server.join()
        result = None
        if rank == 0:
            mat = [[0] * N for _ in range(N)]
            diag = {}
            for d, _, cons in everyone:
                diag.update(cons)                                # (every set is in some rank's pairs)
                for (a, b), v in d.items():
                    mat[a][b] = v
            considered_all = [diag[s] for s in range(N)]
            for s in range(N):
                mat[s][s] = considered_all[s]
            write_matrices(out_dir, names, considered_all, mat)
            result = dict(names=names, considered=considered_all, matrix=mat)
            say("All Commet work is done")
            say("\t Output csv matrices are in:")
            for f in ("matrix_plain.csv", "matrix_percentage.csv", "matrix_normalized.csv"):
                say("\t\t" + out_dir + f)
        slowest = ranks.max_seconds(jobs_s)
        slowest_load = ranks.max_seconds(load_s)
        slowest_filter = ranks.max_seconds(filter_s)
        total_searched = ranks.sum_int(reads_searched)
        total_s = ranks.max_seconds(time.perf_counter() - t_start)
        if result is not None:
            # (the filter processes run beside the parsing: filter_s and load_s overlap, total_s is the wall time of it all)
            result.update(filter_s=slowest_filter, load_s=slowest_load, filter_overlaps_load=filter_pool is not None,
                          load_overlaps_jobs=pipelined, set_wait_s=prof.get("set_wait_s", 0.0), jobs_s=slowest, total_s=total_s,
                          reads_searched=total_searched, world=world, rank0_profile=prof,
                          per_rank=[p for _, p, _c in everyone],
                          reads_per_s=total_searched / slowest if slowest > 0 else 0.0,
                          reads_per_s_incl_load_and_filter=total_searched / total_s if total_s > 0 else 0.0)
            say(f"{total_searched} reads searched in {slowest:.3f} s of jobs on {world} GPU(s): {result['reads_per_s'] / 1e6:.1f} M reads/s "
                f"({result['reads_per_s_incl_load_and_filter'] / 1e6:.1f} M reads/s with filter {slowest_filter:.2f} s + load {slowest_load:.2f} s)")
        # (the gathers above come after every rank's loading: no import of an exported set is still under way)
        for s_, rs in exported.items():
            if s_ not in sets:
                eng.release(rs)
        for rs in sets.values():
            eng.release(rs)
        return result
    except BaseException as ex:
        # with several ranks: tell the others at once (their waits end with an error naming this rank) instead of leaving them in a
        # gather until the timeout
        if world > 1 and hasattr(ranks, "abort"):
            if not isinstance(ex, RuntimeError) or "rendezvous" not in str(ex):
                try:                                              # what this rank ran into may only be the wake of another rank's failure
                    ranks.check()                                 # (scratch gone under its feet): then THAT is the error to report
                except RuntimeError as first:
                    raise first from ex
            ranks.abort(f"{type(ex).__name__}: {ex}")
        raise
    finally:
        failed = sys.exc_info()[0] is not None
        if stop_ev is not None:
            stop_ev.set()
        if failed and world > 1 and exported:
            # the other ranks may be in the middle of importing a set of this one: they notice the abort within a quarter of a second
            # and start no new import; what is under way takes tens of ms.  An exporter that left at once would leave them in a HIP
            # call that never returns (seen: 120 s until their own watchdog).
            time.sleep(float(os.environ.get("COMMET_ABORT_LINGER_S", "2")))
        if loader is not None and loader.is_alive():             # (an error in the job thread)
            loader.join(timeout=5.0 if failed else None)
        stuck = loader is not None and loader.is_alive()         # in a HIP call that does not return: the process is on its way out
        if server is not None and server.is_alive():
            serve_stop.set()
            server.join()
        if canary is not None and canary.poll() is None:         # (never asked: this rank failed first)
            canary.kill()
        if filter_pool is not None:
            filter_pool.shutdown(wait=True, cancel_futures=True)
        if not stuck:
            eng.close()                                          # (never under a thread that is still inside the library)
        if scratch is not None:
            # rank 0 removes the scratch directory once everybody is through; a failing rank removes its own images
            if sys.exc_info()[0] is None and not getattr(ranks, "failed", False):
                ranks.barrier()
                if rank == 0:
                    shutil.rmtree(scratch, ignore_errors=True)
            else:
                for s in owned:
                    for ext in ("pk", "ipc"):
                        try:
                            os.remove(os.path.join(scratch, f"set{s}.{ext}"))
                        except OSError:
                            pass
        if own_ranks and sys.exc_info()[0] is None:
            ranks.close()


def main(argv=None):
    ap = argparse.ArgumentParser(description="Filtering and full N x N intersections of read sets on MI355X GPUs")
    ap.add_argument("input_file")
    ap.add_argument("-b", "--binaries_directory", dest="bin_dir", default=None)
    ap.add_argument("-o", "--output_directory", dest="directory", default="output_commet/")
    ap.add_argument("-k", type=int, default=33)
    ap.add_argument("-t", type=int, default=2)
    ap.add_argument("-l", type=int, default=0)
    ap.add_argument("-n", type=int, default=-1)
    ap.add_argument("-e", type=float, default=0)
    ap.add_argument("-m", type=int, default=-1)
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks to start on this node, one per GPU (ignored under a launcher that has set WORLD_SIZE)")
    a = ap.parse_args(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # this process never touches the GPU: it starts the ranks as plain child processes and leaves with their exit code
        return sharding.spawn_ranks(a.gpus, [sys.executable, "-m", "commet_amd.matrix"] + list(sys.argv[1:] if argv is None else argv))
    try:
        res = run(a.input_file, a.directory, k=a.k, t=a.t, l=a.l, n=a.n, e=a.e, m=a.m, bin_dir=a.bin_dir)
        if res is not None and os.environ.get("COMMET_MATRIX_REPORT"):   # rank 0: times and per-rank profile, as JSON
            import json
            with open(os.environ["COMMET_MATRIX_REPORT"], "w") as fh:
                json.dump({f: v for f, v in res.items() if f != "rank0_profile"}, fh)
    except BaseException:
        # a rank that fails must not leave its peers in a barrier: report and leave at once, skipping the process
        # group's shutdown handshake; torch.distributed.run then terminates the other ranks and exits non-zero
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
