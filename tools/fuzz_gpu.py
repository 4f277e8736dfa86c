"""Extended randomised parity run (not part of the test-suite): job-level GPU results against the CPU checker
on many scenarios of tests/scenarios.py, cycling through index modes, chunk-group sizes, input formats and (forced on)
the bit-sliced regime (narrow tables and wide rows, one and several passes) and the tiled search (32- and 64-bit keys).
  python tools/fuzz_gpu.py [first_seed] [count]"""
import os
import sys
import tempfile
import shutil

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import commet_amd  # noqa: E402
import util  # noqa: E402
from scenarios import Scenario, run_oracle  # noqa: E402


def load_set(ctx, files, sdir):
    rs = commet_amd.ReadSet.from_fasta(ctx, [os.path.join(sdir, fa) for fa, _, _, _ in files])
    sel = np.concatenate([s for _, _, _, s in files]) if files else np.zeros(0, bool)
    return rs, (util.bits_from_bools(sel) if any(bv for _, bv, _, _ in files) else None)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    bad = n_hook = 0
    for seed in range(first, first + count):
        d = tempfile.mkdtemp(prefix="fuzz")
        try:
            mode = seed % 3
            k = None if mode != 2 else [20, 21, 24, 25, 28, 32, 31][seed % 7]
            if seed % 11 >= 7:                  # the round-2 paths need k >= 12 (bit-sliced) / k >= 25 (tiled search; 33, 34: 64-bit keys)
                k = [12, 16, 21, 24, 26, 30, 33, 34, 32][seed % 9]
            fmts = ("fa", "fq", "fa.gz", "fq.gz") if seed % 4 == 0 else ("fa",)
            scn = Scenario(os.path.join(d, "s"), seed, k=k, n_scale=1.0 + (seed % 7), formats=fmts,
                           crlf=False if len(fmts) > 1 else None)
            rc, res, chunks, kmers = run_oracle(scn, os.path.join(d, "o"), os.path.join(d, "l"))
            assert rc == 0
            with commet_amd.Context(k=scn.k, t=scn.t) as ctx:
                hook = False
                counting = seed % 2 == 0           # the probe-counting builds keep groups <= 4 and the full replay
                ctx.set_option("count_probes", int(counting))
                ctx.set_option("index_mode", mode)
                ctx.set_option("chunk_group", 1 + seed % 8)
                if seed % 11 >= 7:              # forced: bit-sliced regime (any number of chunks) or tiled search; small chunks
                    counting = False
                    ctx.set_option("count_probes", 0)
                    ctx.set_option("slice_mode", 2 if seed % 2 else 1)
                    ctx.set_option("slice_words", [0, 1, 2, 4, 8][seed % 5])
                    ctx.set_option("slice_wide", [0, 2, 2][seed % 3])            # round 3: wide rows (search_wide_kernel), ...
                    ctx.set_option("slice_wide_words", [0, 8][(seed // 3) % 2])  # ... in one pass or in passes of 256 chunks
                    hook = seed % 5 == 0
                    if hook:                    # many small chunks from small sets (test hook): the CPU checker chunks by the reference's
                        ctx.set_option("max_kmer", [40, 200, 1500][seed % 3])      # constant, so these are checked against the chunk-at-a-time path
                    ctx.set_option("tiled_search", 2 if seed % 2 == 0 else 1)
                    ctx.set_option("chunk_group", 1 + seed % 3)
                irs, isel = load_set(ctx, scn.sets[scn.index_name], scn.dir)
                names = sorted(scn.search_names)
                loaded = [load_set(ctx, scn.sets[nme], scn.dir) for nme in names]
                try:
                    tags, stats, info = ctx.index_and_search(irs, [x[0] for x in loaded], isel, [x[1] for x in loaded])
                except commet_amd.CommetError as ex:
                    if "bucketed index construction needs" not in str(ex):
                        raise
                    ctx.set_option("index_mode", 0)   # forced on a set it does not take (k, or a read of more than 4096 k-mers)
                    tags, stats, info = ctx.index_and_search(irs, [x[0] for x in loaded], isel, [x[1] for x in loaded])
                if hook:
                    for opt, v in (("slice_mode", 1), ("slice_wide", 1), ("tiled_search", 1), ("chunk_group", 1), ("index_mode", 0)):
                        ctx.set_option(opt, v)   # one chunk filter at a time, search_kernel: the reference's own order
                    t2, s2, i2 = ctx.index_and_search(irs, [x[0] for x in loaded], isel, [x[1] for x in loaded])
                    ok = info["n_chunks"] == i2["n_chunks"] and info["kmers_indexed"] == i2["kmers_indexed"]   # (other chunks drop other look-ahead reads than the checker's)
                    for a, b, sa, sb in zip(tags, t2, stats, s2):
                        ok &= bool(np.array_equal(a, b)) and (sa["indexed"], sa["searched"], sa["shared"]) == (sb["indexed"], sb["searched"], sb["shared"])
                    n_hook += 1
                else:
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
            if not ok:
                bad += 1
                print("MISMATCH seed", seed, "k", scn.k, "t", scn.t, "mode", mode, flush=True)
        finally:
            shutil.rmtree(d, ignore_errors=True)
        if (seed - first) % 100 == 99:
            print(f"... {seed - first + 1} scenarios, {bad} mismatches", flush=True)
    print(f"fuzz: {count} scenarios from seed {first} ({n_hook} of them with the chunk-size hook, checked against the chunk-at-a-time path), {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
