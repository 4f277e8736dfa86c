"""Randomised job-level parity cases shared by tests/test_gpu_fuzz.py (a few hundred seeds, in the suite) and tools/fuzz_gpu.py
(tens of thousands, outside it): one scenario of tests/scenarios.py per seed — index modes, chunk-group sizes, input formats and,
forced on, the bit-sliced regime (narrow tables and wide rows, one and several passes), the tiled search (32- and 64-bit keys) and
the list form of sparse search passes,
a fifth of those with many small chunks (the library's `max_kmer` test hook; the CPU checker is chunked with the same constant).
Every case compares the GPU job with the CPU checker: .bv bits, [indexed, searched, shared], chunk and k-mer counts, and (probe-
counting builds) the reference's probe count.  TEST INFRASTRUCTURE ONLY."""
import os
import shutil
import tempfile

import numpy as np

import util
from scenarios import Scenario, run_oracle


def load_set(commet_amd, ctx, files, sdir):
    rs = commet_amd.ReadSet.from_fasta(ctx, [os.path.join(sdir, fa) for fa, _, _, _ in files])
    sel = np.concatenate([s for _, _, _, s in files]) if files else np.zeros(0, bool)
    return rs, (util.bits_from_bools(sel) if any(bv for _, bv, _, _ in files) else None)


def fuzz_one(seed):
    """Returns (ok, used the chunk-size hook, description)"""
    import commet_amd
    d = tempfile.mkdtemp(prefix="fuzz")
    try:
        mode = seed % 3
        k = None if mode != 2 else [20, 21, 24, 25, 28, 32, 31][seed % 7]
        forced = seed % 11 >= 7             # the bit-sliced regime needs k >= 12, the tiled search k >= 25 (33, 34: 64-bit keys)
        if forced:
            k = [12, 16, 21, 24, 26, 30, 33, 34, 32][seed % 9]
        fmts = ("fa", "fq", "fa.gz", "fq.gz") if seed % 4 == 0 else ("fa",)
        scn = Scenario(os.path.join(d, "s"), seed, k=k, n_scale=1.0 + (seed % 7), formats=fmts,
                       crlf=False if len(fmts) > 1 else None)
        hook = forced and seed % 5 == 0
        max_kmer = [40, 200, 1500][seed % 3] if hook else 0       # many small chunks from small sets; both sides chunk alike
        rc, res, chunks, kmers = run_oracle(scn, os.path.join(d, "o"), os.path.join(d, "l"), max_kmer=max_kmer)
        assert rc == 0
        with commet_amd.Context(k=scn.k, t=scn.t) as ctx:
            counting = seed % 2 == 0           # the probe-counting builds keep groups <= 4 and the full replay
            ctx.set_option("count_probes", int(counting))
            ctx.set_option("index_mode", mode)
            ctx.set_option("chunk_group", 1 + seed % 8)
            # (round 6) ragged sets: hist / scatter1 on the chunk's item list, or — one seed in five — on the round planner
            ctx.set_option("part_list", int(seed % 5 == 3))
            ctx.set_option("ordered_scan", [2, 0, 1][seed % 3])      # (round 6) ragged search sets walked in order of their window counts on a job's first pass
            ctx.set_option("mask_split", int(seed % 7 == 5))         # (round 6) ... segment by segment with the narrowest masks (0) or in one launch (1)
            if not counting and seed % 3 == 1:
                ctx.set_option("sparse_search", 2)     # passes over a selection walk the list of their reads (kernels.hpp, ActiveList)
            if forced:
                counting = False
                ctx.set_option("count_probes", 0)
                ctx.set_option("slice_mode", 2 if seed % 2 else 1)
                ctx.set_option("slice_words", [0, 1, 2, 4, 8][seed % 5])
                ctx.set_option("slice_wide", [0, 2, 2][seed % 3])            # wide rows (search_wide_kernel), ...
                ctx.set_option("slice_wide_words", [0, 8][(seed // 3) % 2])  # ... in one pass or in passes of 256 chunks
                ctx.set_option("max_kmer", max_kmer)
                ctx.set_option("tiled_search", 2 if seed % 2 == 0 else 1)
                ctx.set_option("tq_hit_cap", [1024, 0, 2][(seed // 2) % 3])  # (round 6) the replay's bounded hit list, and pieces that overflow it
                ctx.set_option("chunk_group", 1 + seed % 3)
            irs, isel = load_set(commet_amd, ctx, scn.sets[scn.index_name], scn.dir)
            names = sorted(scn.search_names)
            loaded = [load_set(commet_amd, ctx, scn.sets[nme], scn.dir) for nme in names]
            try:
                tags, stats, info = ctx.index_and_search(irs, [x[0] for x in loaded], isel, [x[1] for x in loaded])
            except commet_amd.CommetError as ex:
                if "bucketed index construction needs" not in str(ex):
                    raise
                ctx.set_option("index_mode", 0)   # forced on a set it does not take (k, or a read of more than 4096 k-mers)
                tags, stats, info = ctx.index_and_search(irs, [x[0] for x in loaded], isel, [x[1] for x in loaded])
            ok = info["n_chunks"] == chunks and info["kmers_indexed"] == kmers and \
                (not counting or info["probes"] == sum(r["probes"] for r in res))
            by = {r["name"]: r for r in res}
            for nme, tg, st in zip(names, tags, stats):
                o = by[nme]
                ok &= (st["indexed"], st["searched"], st["shared"]) == (o["indexed"], o["searched"], o["shared"])
                pos = 0
                for fa, _, reads, _ in scn.sets[nme]:
                    _, n, bits = util.read_bv(os.path.join(d, "o", os.path.basename(fa) + "_in_" + scn.index_name + ".bv"))
                    ok &= bool(np.array_equal(util.bools_from_bits(tg, pos + n)[pos:pos + n], util.bools_from_bits(bits, n)))
                    pos += n
            if seed % 5 == 2 and len(loaded) >= 2 and irs.num_reads:
                # the roles turned round: every search set of the scenario as an index set of its own, the index set searched by all of
                # them in ONE call (commet_index_many_and_search: chunk filters of several jobs in one pass where the sets allow it) — against
                # the same jobs one by one, which the lines above (and every other suite) pin to the CPU checker
                ctx.set_option("count_probes", 0)
                idx = [x[0] for x in loaded if x[0].num_reads]
                sels = [x[1] for x in loaded if x[0].num_reads]
                if len(idx) >= 2:
                    many = ctx.index_many_and_search(idx, irs, sels, isel)
                    for j, (rs_j, sel_j) in enumerate(zip(idx, sels)):
                        one = ctx.index_and_search(rs_j, [irs], sel_j, [isel])
                        ok &= bool(np.array_equal(many[0][j], one[0][0]))
                        ok &= all(many[1][j][f] == one[1][0][f] for f in ("indexed", "searched", "shared"))
        return bool(ok), hook, f"seed {seed} k {scn.k} t {scn.t} index_mode {mode}" + (f" max_kmer {max_kmer}" if hook else "")
    finally:
        shutil.rmtree(d, ignore_errors=True)
