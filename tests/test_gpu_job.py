"""The chunk loop on resident sets (commet_index_and_search) against the CPU
checker run on the same FASTA files: .bv bytes and the [indexed, searched,
shared] numbers must be identical."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import util
from scenarios import Scenario, run_oracle

pytestmark = pytest.mark.gpu


def _load_set(commet, ctx, files, sdir):
    batches = [util.to_batch(util.parse_reads(os.path.join(sdir, fa))) for fa, _, _, _ in files]
    rs = commet.ReadSet.from_files(ctx, batches)
    sel = np.concatenate([s for _, _, _, s in files]) if files else np.zeros(0, bool)
    has_bv = any(bv for _, bv, _, _ in files)
    return rs, (util.bits_from_bools(sel) if has_bv else None)


@pytest.mark.parametrize("seed,index_mode", [(s, 0) for s in range(60)] + [(s, 2) for s in range(60, 80)])
def test_job_matches_oracle(tmp_path, seed, index_mode):
    import commet_amd as commet
    scn = Scenario(str(tmp_path / "scn"), seed, k=None if index_mode == 0 else [20, 21, 24, 25][seed % 4])
    out_o, log_o = str(tmp_path / "out"), str(tmp_path / "log")
    rc, res, chunks, kmers = run_oracle(scn, out_o, log_o)
    assert rc == 0
    with commet.Context(k=scn.k, t=scn.t) as ctx:
        irs, isel = _load_set(commet, ctx, scn.sets[scn.index_name], scn.dir)
        srs, ssel = [], []
        for nme in sorted(scn.search_names):          # std::map order
            r, s = _load_set(commet, ctx, scn.sets[nme], scn.dir)
            srs.append(r)
            ssel.append(s)
        ctx.set_option("count_probes", 1)
        ctx.set_option("index_mode", index_mode)
        ctx.set_option("chunk_group", [4, 1, 2, 3][seed % 4])     # chunk filters searched per pass (1 = reference order)
        tags, stats, info = ctx.index_and_search(irs, srs, isel, ssel)
        assert info["probes"] == sum(r["probes"] for r in res)        # P_ref: the reference's own probe count
        assert info["n_chunks"] == chunks
        assert info["kmers_indexed"] == kmers
        by_name = {r["name"]: r for r in res}
        for nme, tg, st in zip(sorted(scn.search_names), tags, stats):
            o = by_name[nme]
            assert (st["indexed"], st["searched"], st["shared"]) == (o["indexed"], o["searched"], o["shared"]), nme
            pos = 0
            for fa, _, reads, _ in scn.sets[nme]:
                _, n, bits = util.read_bv(os.path.join(out_o, os.path.basename(fa) + "_in_" + scn.index_name + ".bv"))
                exp = util.bools_from_bits(bits, n)
                got = util.bools_from_bits(tg, pos + n)[pos:pos + n]
                assert np.array_equal(got, exp), (nme, fa)
                pos += n
        for r in [irs] + srs:
            r.close()


@pytest.mark.parametrize("seed", range(48))
def test_job_sliced_regime_matches_oracle(tmp_path, seed):
    """the many-small-chunks regime (slice_search.hpp: chunk filters bit-sliced across 32..256-bit entries, one pass of a
    search set per group of chunks) forced on the randomised scenarios: multi-file sets, filter bvs incl. all-zero ones,
    ragged / short / N-rich reads, t = 1..4, 1..many chunks"""
    import commet_amd as commet
    scn = Scenario(str(tmp_path / "scn"), 500 + seed, k=[12, 13, 16, 20, 21, 24][seed % 6], n_scale=[1.0, 4.0, 12.0][seed % 3])
    out_o, log_o = str(tmp_path / "out"), str(tmp_path / "log")
    rc, res, chunks, kmers = run_oracle(scn, out_o, log_o)
    assert rc == 0
    with commet.Context(k=scn.k, t=scn.t) as ctx:
        irs, isel = _load_set(commet, ctx, scn.sets[scn.index_name], scn.dir)
        srs, ssel = [], []
        for nme in sorted(scn.search_names):
            r, s = _load_set(commet, ctx, scn.sets[nme], scn.dir)
            srs.append(r)
            ssel.append(s)
        ctx.set_option("slice_mode", 2)
        ctx.set_option("slice_words", [1, 2, 4, 8][(seed // 6) % 4])
        tags, stats, info = ctx.index_and_search(irs, srs, isel, ssel)
        assert info["n_chunks"] == chunks and info["kmers_indexed"] == kmers
        if chunks:
            assert info["index_launches"] == 2 * ((chunks + 32 * [1, 2, 4, 8][(seed // 6) % 4] - 1) // (32 * [1, 2, 4, 8][(seed // 6) % 4]))
        by_name = {r["name"]: r for r in res}
        for nme, tg, st in zip(sorted(scn.search_names), tags, stats):
            o = by_name[nme]
            assert (st["indexed"], st["searched"], st["shared"]) == (o["indexed"], o["searched"], o["shared"]), nme
            pos = 0
            for fa, _, reads, _ in scn.sets[nme]:
                _, n, bits = util.read_bv(os.path.join(out_o, os.path.basename(fa) + "_in_" + scn.index_name + ".bv"))
                assert np.array_equal(util.bools_from_bits(tg, pos + n)[pos:pos + n], util.bools_from_bits(bits, n)), (nme, fa)
                pos += n


@pytest.mark.parametrize("seed", range(30))
def test_job_wide_rows_match_oracle(tmp_path, seed):
    """the wide rows of the many-small-chunks regime (slice_search.hpp, search_wide_kernel: every chunk filter of a pass
    side by side in one table, a group of lanes per read) forced on the randomised scenarios: multi-file sets, filter
    bvs incl. all-zero ones, ragged / short / N-rich reads, t = 1..4; with the row capped at 8 words the jobs of more
    than 256 chunks take several passes"""
    import commet_amd as commet
    scn = Scenario(str(tmp_path / "scn"), 700 + seed, k=[12, 13, 16, 20, 21, 24][seed % 6], n_scale=[1.0, 4.0, 12.0][seed % 3])
    out_o, log_o = str(tmp_path / "out"), str(tmp_path / "log")
    rc, res, chunks, kmers = run_oracle(scn, out_o, log_o)
    assert rc == 0
    with commet.Context(k=scn.k, t=scn.t) as ctx:
        irs, isel = _load_set(commet, ctx, scn.sets[scn.index_name], scn.dir)
        srs, ssel = [], []
        for nme in sorted(scn.search_names):
            r, s = _load_set(commet, ctx, scn.sets[nme], scn.dir)
            srs.append(r)
            ssel.append(s)
        ctx.set_option("slice_mode", 2)
        ctx.set_option("slice_wide", 2)
        ctx.set_option("slice_wide_words", [0, 8][(seed // 6) % 2])
        ctx.set_option("kernel_timing", 1)
        tags, stats, info = ctx.index_and_search(irs, srs, isel, ssel)
        assert info["n_chunks"] == chunks and info["kmers_indexed"] == kmers
        if chunks and any(r.num_reads for r in srs):
            assert "search_wide_kernel" in ctx.kernel_times() and "search_sliced_kernel" not in ctx.kernel_times()
        by_name = {r["name"]: r for r in res}
        for nme, tg, st in zip(sorted(scn.search_names), tags, stats):
            o = by_name[nme]
            assert (st["indexed"], st["searched"], st["shared"]) == (o["indexed"], o["searched"], o["shared"]), nme
            pos = 0
            for fa, _, reads, _ in scn.sets[nme]:
                _, n, bits = util.read_bv(os.path.join(out_o, os.path.basename(fa) + "_in_" + scn.index_name + ".bv"))
                assert np.array_equal(util.bools_from_bits(tg, pos + n)[pos:pos + n], util.bools_from_bits(bits, n)), (nme, fa)
                pos += n


@pytest.mark.parametrize("seed", range(48))
def test_job_tiled_search_matches_oracle(tmp_path, seed):
    """the tiled search (tile_search.hpp: the set's lane-a addresses sorted by address slice once, probed slice by slice out
    of L2, replayed piece by piece) forced on the randomised scenarios, for groups of one and of two chunk filters (larger
    groups keep the gather kernels; the last group of a job may still be tiled)"""
    import commet_amd as commet
    k = [25, 26, 28, 30, 32, 33, 34, 32][seed % 8]          # 33, 34: 64-bit keys (tq_*<uint64_t>)
    scn = Scenario(str(tmp_path / "scn"), 900 + seed, k=k, n_scale=[1.0, 6.0, 20.0][seed % 3])
    max_kmer = [0, 3000, 900][(seed // 8) % 3]              # more chunks from small sets (test hook; the CPU checker is chunked alike)
    out_o, log_o = str(tmp_path / "out"), str(tmp_path / "log")
    rc, res, chunks, kmers = run_oracle(scn, out_o, log_o, max_kmer=max_kmer)
    assert rc == 0
    with commet.Context(k=scn.k, t=scn.t) as ctx:
        irs, isel = _load_set(commet, ctx, scn.sets[scn.index_name], scn.dir)
        srs, ssel = [], []
        for nme in sorted(scn.search_names):
            r, s = _load_set(commet, ctx, scn.sets[nme], scn.dir)
            srs.append(r)
            ssel.append(s)
        ctx.set_option("tiled_search", 2)
        ctx.set_option("max_kmer", max_kmer)
        ctx.set_option("chunk_group", [2, 1, 3][(seed // 3) % 3])
        # (round 6) the replay posts the light scans' full hits as a list of bounded length; a piece with more of them lets every scan
        # walk its own candidates: a third of the seeds run with a list of 0 or 3 entries, i.e. on that path
        ctx.set_option("tq_hit_cap", [1024, 0, 3][(seed // 2) % 3])
        got = ctx.index_and_search(irs, srs, isel, ssel)
        ctx.set_option("tiled_search", 1)
        ref = ctx.index_and_search(irs, srs, isel, ssel)                       # the gather kernels, same chunking
        for a, b in zip(got[0], ref[0]):
            assert np.array_equal(a, b)
        assert [(s["indexed"], s["searched"], s["shared"]) for s in got[1]] == [(s["indexed"], s["searched"], s["shared"]) for s in ref[1]]
        # every seed against the CPU checker's bits and log numbers (chunked with the same constant)
        tags, stats, info = got
        assert info["n_chunks"] == chunks and info["kmers_indexed"] == kmers
        by_name = {r["name"]: r for r in res}
        for nme, tg, st in zip(sorted(scn.search_names), tags, stats):
            o = by_name[nme]
            assert (st["indexed"], st["searched"], st["shared"]) == (o["indexed"], o["searched"], o["shared"]), nme
            pos = 0
            for fa, _, reads, _ in scn.sets[nme]:
                _, n, bits = util.read_bv(os.path.join(out_o, os.path.basename(fa) + "_in_" + scn.index_name + ".bv"))
                assert np.array_equal(util.bools_from_bits(tg, pos + n)[pos:pos + n], util.bools_from_bits(bits, n)), (nme, fa)
                pos += n


@pytest.mark.parametrize("seed", range(32))
def test_job_ordered_scan_matches_oracle(tmp_path, seed):
    """ragged sets walked in order of their first-hit window counts (round 6: the list form of the gather kernels fed with the set's
    length-ordered list — tile_search.hpp lo_*_kernel, capi ordered_pass — on a job's first pass over a set that is visited whole),
    forced on the randomised scenarios for the plain kernel and groups of 2, 4 and 5..8 chunk filters; every seed against the CPU
    checker's bits and log numbers, and against the same job in natural order"""
    import commet_amd as commet
    k = [20, 25, 28, 32, 33, 16, 31, 34][seed % 8]
    scn = Scenario(str(tmp_path / "scn"), 1300 + seed, k=k, n_scale=[1.0, 5.0, 12.0][seed % 3])
    max_kmer = [0, 2500, 700][(seed // 8) % 3]
    out_o, log_o = str(tmp_path / "out"), str(tmp_path / "log")
    rc, res, chunks, kmers = run_oracle(scn, out_o, log_o, max_kmer=max_kmer)
    assert rc == 0
    with commet.Context(k=scn.k, t=scn.t) as ctx:
        irs, isel = _load_set(commet, ctx, scn.sets[scn.index_name], scn.dir)
        srs, ssel = [], []
        for nme in sorted(scn.search_names):
            r, s = _load_set(commet, ctx, scn.sets[nme], scn.dir)
            srs.append(r)
            ssel.append(s)
        ctx.set_option("tiled_search", 1)
        ctx.set_option("slice_mode", 1)
        ctx.set_option("max_kmer", max_kmer)
        ctx.set_option("chunk_group", [8, 1, 4, 2][(seed // 2) % 4])
        ctx.set_option("ordered_scan", 2)
        ctx.set_option("kernel_timing", 1)
        got = ctx.index_and_search(irs, srs, isel, ssel)
        times = ctx.kernel_times()
        ctx.set_option("kernel_timing", 0)
        ctx.set_option("ordered_scan", 1)
        ref = ctx.index_and_search(irs, srs, isel, ssel)
        for a, b in zip(got[0], ref[0]):
            assert np.array_equal(a, b)
        assert [(s["indexed"], s["searched"], s["shared"]) for s in got[1]] == [(s["indexed"], s["searched"], s["shared"]) for s in ref[1]]
        tags, stats, info = got
        assert info["n_chunks"] == chunks and info["kmers_indexed"] == kmers
        by_name = {r["name"]: r for r in res}
        for nme, tg, st in zip(sorted(scn.search_names), tags, stats):
            o = by_name[nme]
            assert (st["indexed"], st["searched"], st["shared"]) == (o["indexed"], o["searched"], o["shared"]), nme
            pos = 0
            for fa, _, reads, _ in scn.sets[nme]:
                _, n, bits = util.read_bv(os.path.join(out_o, os.path.basename(fa) + "_in_" + scn.index_name + ".bv"))
                assert np.array_equal(util.bools_from_bits(tg, pos + n)[pos:pos + n], util.bools_from_bits(bits, n)), (nme, fa)
                pos += n



@pytest.mark.parametrize("seed", range(24))
def test_job_sparse_passes_match_oracle(tmp_path, seed):
    """a pass over a selection of a search set walks the list of its reads (kernels.hpp ActiveList: sel & ~tags, in order, re-made
    per pass) instead of the set's bitmap — forced on the randomised scenarios (multi-file sets, filter bvs incl. all-zero ones,
    ragged / short / N-rich reads, t = 1..4) for the plain kernel and groups of 2, 4 and 5..8 chunk filters, 32- and 64-bit keys"""
    import commet_amd as commet
    k = [16, 20, 25, 32, 33, 12, 28, 34][seed % 8]
    scn = Scenario(str(tmp_path / "scn"), 1300 + seed, k=k, n_scale=[1.0, 6.0, 20.0][seed % 3], allow_zero_bv=seed % 5 == 0)
    max_kmer = [0, 2500, 700][(seed // 8) % 3] if k >= 16 else 0
    out_o, log_o = str(tmp_path / "out"), str(tmp_path / "log")
    rc, res, chunks, kmers = run_oracle(scn, out_o, log_o, max_kmer=max_kmer)
    assert rc == 0
    with commet.Context(k=scn.k, t=scn.t) as ctx:
        irs, isel = _load_set(commet, ctx, scn.sets[scn.index_name], scn.dir)
        srs, ssel = [], []
        for nme in sorted(scn.search_names):
            r, s = _load_set(commet, ctx, scn.sets[nme], scn.dir)
            srs.append(r)
            ssel.append(s)
        ctx.set_option("sparse_search", 2)
        ctx.set_option("slice_mode", 1)                       # (the bit-sliced regime has no list form: the slot kernels)
        ctx.set_option("max_kmer", max_kmer)
        ctx.set_option("chunk_group", [8, 1, 2, 4][(seed // 2) % 4])
        ctx.set_option("kernel_timing", 1)
        tags, stats, info = ctx.index_and_search(irs, srs, isel, ssel)
        times = ctx.kernel_times()
        assert info["n_chunks"] == chunks and info["kmers_indexed"] == kmers
        # a set with a filter bv that does not select every read, searched against at least one chunk: its passes walked a list
        partial = [any(bv and not sel.all() for _, bv, _, sel in scn.sets[nme]) and sum(len(r) for _, _, r, _ in scn.sets[nme]) > 0
                   for nme in sorted(scn.search_names)]
        if chunks and any(partial):
            assert "active_list_kernels" in times
        by_name = {r["name"]: r for r in res}
        for nme, tg, st in zip(sorted(scn.search_names), tags, stats):
            o = by_name[nme]
            assert (st["indexed"], st["searched"], st["shared"]) == (o["indexed"], o["searched"], o["shared"]), nme
            pos = 0
            for fa, _, reads, _ in scn.sets[nme]:
                _, n, bits = util.read_bv(os.path.join(out_o, os.path.basename(fa) + "_in_" + scn.index_name + ".bv"))
                assert np.array_equal(util.bools_from_bits(tg, pos + n)[pos:pos + n], util.bools_from_bits(bits, n)), (nme, fa)
                pos += n


@pytest.mark.parametrize("k,t,chunk_group", [(20, 2, 4), (24, 3, 2), (33, 2, 4), (16, 2, 1)])
def test_long_and_ragged_reads(tmp_path, k, t, chunk_group):
    """reads far longer than the fast paths' limits (search masks cover 256 bases, chunk groups need <= 64 KiB of
    masks, the bucketed index needs <= 4096 k-mers per read): every fallback must still match the CPU checker"""
    import commet_amd as commet
    rng = np.random.default_rng(k * 7 + t)
    d = tmp_path / "long"
    os.makedirs(d)
    lens = [int(x) for x in rng.choice([40, 150, 257, 300, 1000, 4100, 6000], size=260)]
    idx_reads = [util.random_reads(rng, 1, L, L, n_rate=0.002)[0] for L in lens]
    q_src = idx_reads[:120]
    q_reads = []
    for i in range(400):
        r = q_src[int(rng.integers(0, len(q_src)))]
        a = int(rng.integers(0, max(1, len(r) - 30)))
        piece = r[a:a + int(rng.integers(30, 2500))]
        if i % 3 == 0:
            piece = util.revcomp(piece)
        if i % 5 == 0:
            piece = util.mutate(rng, piece, 0.01)
        q_reads.append(util.random_reads(rng, 1, 5, 400)[0] + piece if i % 2 else piece)
    q_reads += util.random_reads(rng, 100, 20, 700)
    util.write_fasta(str(d / "i.fa"), idx_reads, rng=rng, multiline=True)
    util.write_fasta(str(d / "q.fa"), q_reads, rng=rng, multiline=True)
    (d / "index.txt").write_text("I:i.fa\n")
    (d / "search.txt").write_text("Q:q.fa\n")

    class S:      # what run_oracle needs
        dir, index_cfg, search_cfg = str(d), "index.txt", "search.txt"
    S.k, S.t = k, t
    rc, res, chunks, kmers = run_oracle(S, str(tmp_path / "o"), str(tmp_path / "l"))
    assert rc == 0
    with commet.Context(k=k, t=t) as ctx:
        ctx.set_option("count_probes", 1)
        ctx.set_option("chunk_group", chunk_group)
        irs = commet.ReadSet.from_fasta(ctx, [str(d / "i.fa")])
        qrs = commet.ReadSet.from_fasta(ctx, [str(d / "q.fa")])
        tags, stats, info = ctx.index_and_search(irs, [qrs])
        assert info["n_chunks"] == chunks and info["kmers_indexed"] == kmers
        assert info["probes"] == res[0]["probes"]
        assert (stats[0]["indexed"], stats[0]["searched"], stats[0]["shared"]) == (res[0]["indexed"], res[0]["searched"], res[0]["shared"])
        _, n, bits = util.read_bv(str(tmp_path / "o" / "q.fa_in_I.bv"))
        assert np.array_equal(util.bools_from_bits(tags[0], n), util.bools_from_bits(bits, n))
        assert stats[0]["shared"] > 100


@pytest.mark.parametrize("k,L,uniform,n_idx", [(24, 100, True, 150000), (25, 90, True, 200000), (25, 120, False, 260000)])
def test_bucketed_multi_chunk_variants_match_oracle(tmp_path, k, L, uniform, n_idx):
    """several chunks, every one on the bucketed construction (part_min_kmers lowered): the two index lanes, the packed
    final level and the fixed-read-length fast path are each switched on and off — every combination must give the CPU
    checker's tags, chunk count and log numbers (groups of 8 — the register-mask kernel where the read length allows it —,
    4, 2 and 1 chunk filters per pass)"""
    import commet_amd as commet
    rng = np.random.default_rng(100 * k + L)
    lo = L if uniform else 40
    idx_reads = util.random_reads(rng, n_idx, lo, L, n_rate=0.003)
    idx_reads += [b"A" * L] * 300 + [(b"ACGT" * L)[:L]] * 200                     # hot buckets
    q_reads = util.related_reads(rng, idx_reads[:20000], 30000, lo, L, share=0.5, n_rate=0.003)
    d = tmp_path / "mc"
    os.makedirs(d)
    util.write_fasta(str(d / "i.fa"), idx_reads)
    util.write_fasta(str(d / "q.fa"), q_reads)
    (d / "index.txt").write_text("I:i.fa\n")
    (d / "search.txt").write_text("Q:q.fa\n")

    class S:
        dir, index_cfg, search_cfg = str(d), "index.txt", "search.txt"
    S.k, S.t = k, 2
    rc, res, chunks, kmers = run_oracle(S, str(tmp_path / "o"), str(tmp_path / "l"))
    assert rc == 0 and chunks >= 3, chunks
    _, n, bits = util.read_bv(str(tmp_path / "o" / "q.fa_in_I.bv"))
    want = util.bools_from_bits(bits, n)
    with commet.Context(k=k, t=2) as ctx:
        ctx.set_option("index_mode", 2)
        ctx.set_option("part_min_kmers", 1000)
        irs = commet.ReadSet.from_fasta(ctx, [str(d / "i.fa")])
        qrs = commet.ReadSet.from_fasta(ctx, [str(d / "q.fa")])
        for lanes, packed, no_uni, group in [(2, 1, 0, 8), (2, 1, 0, 4), (1, 1, 0, 4), (2, 0, 0, 2), (2, 1, 1, 8), (1, 0, 1, 1)]:
            ctx.set_option("index_lanes", lanes)
            ctx.set_option("part_packed", packed)
            ctx.set_option("part_no_uni", no_uni)
            ctx.set_option("chunk_group", group)
            tags, stats, info = ctx.index_and_search(irs, [qrs])
            assert info["n_chunks"] == chunks and info["kmers_indexed"] == kmers, (lanes, packed, no_uni, group)
            assert (stats[0]["indexed"], stats[0]["searched"], stats[0]["shared"]) == \
                (res[0]["indexed"], res[0]["searched"], res[0]["shared"]), (lanes, packed, no_uni, group)
            assert np.array_equal(util.bools_from_bits(tags[0], n), want), (lanes, packed, no_uni, group)
        assert stats[0]["shared"] > 1000


def test_record_longer_than_a_staging_buffer(tmp_path):
    """a 9 Mbase record (a contig used as a set: the reference accepts any length) is longer than one pinned staging buffer
    of the ingest (3 MiB of planes = 8.4 M bases): the packer uploads it in pieces; results must equal the CPU checker's"""
    import commet_amd as commet
    rng = np.random.default_rng(123)
    d = tmp_path / "long"
    os.makedirs(d)
    contig = util.random_reads(rng, 1, 9_000_000, 9_000_000, n_rate=0.0005, lower_rate=0, other_rate=0)[0]
    shorts = util.random_reads(rng, 50, 60, 200)
    q = [contig[a:a + 150] for a in rng.integers(0, len(contig) - 200, size=300)] + util.random_reads(rng, 300, 100, 150)
    q = [util.revcomp(r) if i % 3 == 0 else r for i, r in enumerate(q)]
    with open(d / "i.fa", "wb") as fh:
        fh.write(b">s0\n" + shorts[0] + b"\n>contig\n")
        for j in range(0, len(contig), 70):
            fh.write(contig[j:j + 70] + b"\n")
        for i, r in enumerate(shorts[1:]):
            fh.write(b">s%d\n" % (i + 1) + r + b"\n")
    util.write_fasta(str(d / "q.fa"), q)
    (d / "index.txt").write_text("I:i.fa\n")
    (d / "search.txt").write_text("Q:q.fa\n")

    class S:
        dir, index_cfg, search_cfg = str(d), "index.txt", "search.txt"
    S.k, S.t = 20, 2
    rc, res, chunks, kmers = run_oracle(S, str(tmp_path / "o"), str(tmp_path / "l"))
    assert rc == 0
    _, n, bits = util.read_bv(str(tmp_path / "o" / "q.fa_in_I.bv"))
    with commet.Context(k=20, t=2) as ctx:
        irs = commet.ReadSet.from_fasta(ctx, [str(d / "i.fa")])
        qrs = commet.ReadSet.from_fasta(ctx, [str(d / "q.fa")])
        assert irs.num_reads == 51 and int(irs.kmer_counts().max()) > 8_900_000
        tags, stats, info = ctx.index_and_search(irs, [qrs])
        assert info["n_chunks"] == chunks and info["kmers_indexed"] == kmers
        assert (stats[0]["indexed"], stats[0]["searched"], stats[0]["shared"]) == (res[0]["indexed"], res[0]["searched"], res[0]["shared"])
        assert np.array_equal(util.bools_from_bits(tags[0], n), util.bools_from_bits(bits, n))
        assert stats[0]["shared"] >= 250


@pytest.mark.parametrize("k,L,density,max_kmer", [(25, 100, 0.02, 0), (25, 100, 0.3, 300000), (32, 100, 0.25, 0), (33, 101, 0.5, 400000), (21, 60, 0.9, 150000)])
def test_selected_index_reads_as_a_list_match_the_round_planner(k, L, density, max_kmer):
    """an index selection on a fixed-length set (Commet.py's J2 / J3 jobs) is compacted into a list of read numbers that
    hist / scatter1 walk arithmetically (index_part.hpp, sel_ids_kernel): same bits as the round planner over the bitmap
    (part_no_uni), as the atomic kernel, and as the CPU checker fed the selected reads chunk by chunk"""
    import commet_amd as commet
    import oracle_pool
    rng = np.random.default_rng(1000 * k + int(100 * density))
    n = 40000
    idx_reads = util.random_reads(rng, n, L, L, n_rate=0.003)
    q_reads = util.related_reads(rng, idx_reads, 20000, L, L, share=0.6, n_rate=0.003)
    ib, io = util.to_batch(idx_reads)
    qb, qo = util.to_batch(q_reads)
    sel = rng.random(n) < density
    sel[:3] = [True, False, True]
    sb = util.bits_from_bools(sel)
    res = {}
    with commet.Context(k=k, t=2) as ctx:
        if max_kmer:
            ctx.set_option("max_kmer", max_kmer)
        ctx.set_option("index_mode", 2)
        irs = commet.ReadSet.from_files(ctx, [(ib, io)])
        qrs = commet.ReadSet.from_files(ctx, [(qb, qo)])
        kc = irs.kmer_counts()
        ctx.set_option("kernel_timing", 1)
        res["list"] = ctx.index_and_search(irs, [qrs], index_select=sb)
        assert "sel_ids_kernels" in ctx.kernel_times()
        ctx.set_option("kernel_timing", 1)
        ctx.set_option("part_no_uni", 1)
        res["planner"] = ctx.index_and_search(irs, [qrs], index_select=sb)
        assert "sel_ids_kernels" not in ctx.kernel_times()
        ctx.set_option("part_no_uni", 0)
        ctx.set_option("index_mode", 1)
        res["atomic"] = ctx.index_and_search(irs, [qrs], index_select=sb)
    ids = np.flatnonzero(sel)
    sib = np.ascontiguousarray(ib.reshape(n, L)[ids]).reshape(-1)
    sio = np.arange(len(ids) + 1, dtype=np.uint64) * np.uint64(L)
    chunks = oracle_pool.chunks_from_counts(kc[ids], max_kmer or ob.max_kmer(k))
    found = np.zeros(len(q_reads) // 8 + 1, dtype=np.uint8)
    for (a, e) in chunks:
        f = ob.Bloom(k)
        f.index(sib[a * L: e * L], sio[a: e + 1] - sio[a])
        fnd, _ = f.search(2, qb, qo, ~found)
        found |= fnd
        f.close()
    for name, (tags, stats, info) in res.items():
        assert info["n_chunks"] == len(chunks), name
        assert np.array_equal(tags[0], found), name
        assert stats[0]["indexed"] == sum(e - a for a, e in chunks), name
    assert res["list"][1][0]["shared"] > 50
