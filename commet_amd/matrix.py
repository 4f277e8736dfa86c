"""N x N comparison driver with set residency — the MI355X counterpart of Commet.py's
local mode (reference: Commet.py:438-598), SURVEY 8f-2.

Commet.py runs N^2-1 `index_and_search` processes one after the other; every one re-parses
its FASTA files and re-creates its filter, which on a GPU means a HIP start-up, a re-upload
and a 16 GB scratch allocation per job.  Here every rank parses and uploads each set ONCE,
keeps it packed in HBM, and runs its share of the job DAG in-process through the C ABI:

    for ref < i :   J1  index S_ref                          search S_i      -> T1
                    J2  index S_i   restricted to T1         search S_ref    -> <G>_in_<S_i>.bv
                    J3  index S_ref restricted to J2's bits  search S_i      -> <F>_in_<S_ref>.bv

(J1 of one `ref` is run once per rank for all of the rank's `i`, as Commet.py does.)
Ranks are independent: pairs (ref, i) are dealt out, no collective touches read data; only
the per-pair counts are gathered (host, gloo) for the three CSV matrices.

Same inputs and outputs as Commet.py: the set file `name: file[,bv]; file…`, the filter
step (`filter_reads`, skipped when bvs are given), `OUT/<file>_in_<set>.bv`,
`OUT/<s>_in_<i>.log`, `matrix_plain.csv`, `matrix_percentage.csv`, `matrix_normalized.csv`.

  python -m commet_amd.matrix sets.txt -k 32 -t 2 -o out/            (1 GPU)
  python -m torch.distributed.run --nproc-per-node 8 -m commet_amd.matrix sets.txt …   (8 GPUs)
"""
import argparse
import os
import subprocess
import sys
import time

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


def run(input_file, out_dir, k=33, t=2, l=0, n=-1, e=0.0, m=-1, bin_dir=None, ranks=None, verbose=True):
    import commet_amd
    own_ranks = ranks is None
    if ranks is None:
        ranks = sharding.Ranks(backend="gloo")
    if out_dir[-1] != "/":
        out_dir += "/"
    bin_dir = bin_dir or os.path.join(HERE, "bin")
    os.makedirs(out_dir, exist_ok=True)
    names, files, bvs = parse_set_file(input_file)
    N = len(names)
    say = print if (verbose and ranks.rank == 0) else (lambda *a, **kw: None)

    if l < k * t and l != 0:                                      # Commet.py:509-513 (l stays 0 by default)
        l = k * t
    # ---- filter step (Commet.py:103-121): one filter_reads per file, dealt over the ranks -------------
    t_filter = time.perf_counter()
    if bvs is None:
        bvs = [[out_dir + os.path.basename(f) + ".bv" for f in fl] for fl in files]
        todo = [(s, j) for s in range(N) for j in range(len(files[s]))]
        cmds = []
        for q, (s, j) in enumerate(todo):
            if q % ranks.world != ranks.rank:
                continue
            cmd = [os.path.join(bin_dir, "filter_reads"), files[s][j], "-l", str(l), "-e", str(e)]
            if n >= 0:
                cmd += ["-n", str(n)]
            if m >= 0:
                cmd += ["-m", str(m / len(files[s]))]
            cmd += ["-o", bvs[s][j]]
            say("Filtering command: " + " ".join(cmd))
            cmds.append(cmd)
        # independent processes (each one multi-threaded over its file): a few at a time
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=int(os.environ.get("COMMET_FILTER_JOBS", "3"))) as pool:
            list(pool.map(lambda c: subprocess.run(c, check=True, stdout=subprocess.DEVNULL), cmds))
        ranks.barrier()

    filter_s = time.perf_counter() - t_filter
    # ---- residency: every rank holds every set (packed: 12 B per 32 bases) ----------------------------------
    t0 = time.perf_counter()
    # COMMET_FORCE_DEVICE: debugging aid to run several ranks on one GPU
    ctx = commet_amd.Context(k=k, t=t, device=int(os.environ.get("COMMET_FORCE_DEVICE", ranks.local_rank)))
    sets = [commet_amd.ReadSet.from_fasta(ctx, fl) for fl in files]
    counts = [rs.file_reads() for rs in sets]
    sel = []
    considered = []
    for s in range(N):
        parts = [read_bv(b) for b in bvs[s]]
        for (nb, _), c, f in zip(parts, counts[s], files[s]):
            if nb != c:
                raise commet_amd.CommetError(f"Number of reads in {f} and boolean vector size are not equal -> quit")
        _, bits = concat_bits(parts)
        sel.append(bits)
        considered.append(sum(popcount(b, nb) for nb, b in parts))
    load_s = time.perf_counter() - t0
    say(f"loaded {N} sets ({sum(rs.num_reads for rs in sets)} reads) in {load_s:.2f} s")

    # ---- my pairs, grouped by ref ------------------------------------------------------------------------
    pairs = [(ref, i) for ref in range(N - 1) for i in range(ref + 1, N)]
    cost = [float(sets[a].num_reads + sets[b].num_reads) for a, b in pairs]
    mine = [pairs[c] for c in sharding.assign_chains(pairs, ranks.world, ranks.rank, cost)]
    shared = {}                    # (from set, in set) -> reads of `from` found in `in`
    reads_searched = 0
    prof = dict(jobs=0, call_ms=0.0, device_ms=0.0)      # library calls: wall time vs device time

    def _acc(inf):
        prof["jobs"] += 1
        prof["call_ms"] += inf["total_ms"]
        prof["device_ms"] += inf["index_ms"] + inf["search_ms"]

    t_jobs = time.perf_counter()
    for ref in sorted({p[0] for p in mine}):
        targets = [i for (r, i) in mine if r == ref]
        w0 = time.perf_counter()
        tags1, st1, inf1 = ctx.index_and_search(sets[ref], [sets[i] for i in targets], sel[ref], [sel[i] for i in targets])
        reads_searched += sum(considered[i] for i in targets)
        _acc(inf1)
        for i, T1 in zip(targets, tags1):
            # J2: X = S_i restricted to (S_i in S_ref); S_ref in X
            tags2, st2, inf2 = ctx.index_and_search(sets[i], [sets[ref]], T1, [sel[ref]])
            T2 = tags2[0]
            _acc(inf2)
            for f, c, b in zip(files[ref], counts[ref], split_bits(T2, counts[ref])):
                write_bv(out_dir + os.path.basename(f) + "_in_" + names[i] + ".bv", f + " in " + names[i], c, b)
            _log(out_dir, names[ref], names[i], st2[0], inf2["index_ms"], time.perf_counter() - w0)
            shared[(ref, i)] = st2[0]["shared"]
            # J3: S_i in (S_ref restricted to J2's result)  — overwrites J1's <F>_in_<S_ref>.bv (Commet.py:233)
            tags3, st3, inf3 = ctx.index_and_search(sets[ref], [sets[i]], T2, [sel[i]])
            _acc(inf3)
            for f, c, b in zip(files[i], counts[i], split_bits(tags3[0], counts[i])):
                write_bv(out_dir + os.path.basename(f) + "_in_" + names[ref] + ".bv", f + " in " + names[ref], c, b)
            _log(out_dir, names[i], names[ref], st3[0], inf3["index_ms"], time.perf_counter() - w0)
            shared[(i, ref)] = st3[0]["shared"]
            reads_searched += considered[ref] + considered[i]
    ctx.synchronize()
    jobs_s = time.perf_counter() - t_jobs
    # ---- matrices on rank 0 -----------------------------------------------------------------------------
    everyone = ranks.gather_objects(shared)
    result = None
    if ranks.rank == 0:
        mat = [[0] * N for _ in range(N)]
        for d in everyone:
            for (a, b), v in d.items():
                mat[a][b] = v
        for s in range(N):
            mat[s][s] = considered[s]
        write_matrices(out_dir, names, considered, mat)
        result = dict(names=names, considered=considered, matrix=mat)
        say("All Commet work is done")
        say("\t Output csv matrices are in:")
        for f in ("matrix_plain.csv", "matrix_percentage.csv", "matrix_normalized.csv"):
            say("\t\t" + out_dir + f)
    slowest = ranks.max_seconds(jobs_s)
    total_searched = ranks.sum_int(reads_searched)
    if result is not None:
        result.update(filter_s=filter_s, load_s=load_s, jobs_s=slowest, reads_searched=total_searched, world=ranks.world, rank0_profile=prof,
                      reads_per_s=total_searched / slowest if slowest > 0 else 0.0)
        say(f"{total_searched} reads searched in {slowest:.3f} s on {ranks.world} GPU(s): {result['reads_per_s'] / 1e6:.1f} M reads/s")
    for rs in sets:
        rs.close()
    ctx.close()
    if own_ranks:
        ranks.close()
    return result


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
    a = ap.parse_args(argv)
    run(a.input_file, a.directory, k=a.k, t=a.t, l=a.l, n=a.n, e=a.e, m=a.m, bin_dir=a.bin_dir)
    return 0


if __name__ == "__main__":
    sys.exit(main())
