"""Parity of the HIP kernels with the CPU checker, through the C ABI.
Bit-exact: integer / bit work only."""
import numpy as np
import pytest

import oracle_binding as ob
import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def commet():
    import commet_amd
    return commet_amd


def _mixed_reads(seed, n=700, lo=0, hi=140):
    rng = np.random.default_rng(seed)
    base = util.random_reads(rng, n // 2, max(lo, 1), hi, n_rate=0.02)
    rel = util.related_reads(rng, base, n - n // 2, max(lo, 1), hi, share=0.6, n_rate=0.02)
    reads = base + rel
    if lo == 0:
        reads[3] = b""
        reads[len(reads) // 2] = b""
    return reads


@pytest.mark.parametrize("k", [1, 2, 5, 8, 13, 20, 31, 32, 33, 36])
def test_pack_and_kmer_counts(commet, k):
    reads = _mixed_reads(11 + k)
    bases, offs = util.to_batch(reads)
    with commet.Context(k=min(k, 20) if k > 33 else k, t=2) as ctx:
        kk = ctx.k
        rs = commet.ReadSet.from_files(ctx, [(bases, offs)])
        got = rs.kmer_counts()
        exp = ob.kmer_counts(bases, offs, kk)
        assert np.array_equal(got, exp)
        rs.close()


@pytest.mark.parametrize("k", [1, 3, 8, 12, 16, 21, 25, 28])
def test_index_filter_bits_match_reference_layout(commet, k):
    reads = _mixed_reads(100 + k, n=900)
    bases, offs = util.to_batch(reads)
    rng = np.random.default_rng(k)
    sel = rng.random(len(reads)) < 0.7
    with commet.Context(k=k, t=2) as ctx:
        rs = commet.ReadSet.from_files(ctx, [(bases[: int(offs[400])], offs[:401]),
                                             (bases[int(offs[400]):], offs[400:] - offs[400])])
        assert rs.num_reads == len(reads) and rs.num_files == 2
        # all reads
        ctx.filter_reset()
        fed = ctx.index_reads(rs)
        f = ob.Bloom(k)
        assert fed == f.index(bases, offs)
        assert np.array_equal(ctx.export_filter_reference(), f.bytes())
        # a range with select bits
        ctx.filter_reset()
        first, count = 123, 555
        fed = ctx.index_reads(rs, first, count, util.bits_from_bools(sel))
        f2 = ob.Bloom(k)
        sel2 = sel.copy()
        sel2[:first] = False
        sel2[first + count:] = False
        assert fed == f2.index(bases, offs, util.bits_from_bools(sel2))
        assert np.array_equal(ctx.export_filter_reference(), f2.bytes())
        rs.close()


@pytest.mark.parametrize("k,t", [(1, 1), (4, 3), (8, 1), (8, 2), (12, 2), (12, 4), (16, 2), (20, 1), (20, 2), (25, 3),
                                  (28, 2), (31, 2), (32, 2)])
def test_search_matches_oracle(commet, k, t):
    rng = np.random.default_rng(1000 * k + t)
    idx_reads = util.random_reads(rng, 500, 20, 130, n_rate=0.01)
    q_reads = util.related_reads(rng, idx_reads, 1500, 1, 130, share=0.5, n_rate=0.02)
    ib, io = util.to_batch(idx_reads)
    qb, qo = util.to_batch(q_reads)
    active = rng.random(len(q_reads)) < 0.8
    with commet.Context(k=k, t=t) as ctx:
        irs = commet.ReadSet.from_files(ctx, [(ib, io)])
        qrs = commet.ReadSet.from_files(ctx, [(qb, qo)])
        ctx.filter_reset()
        ctx.index_reads(irs)
        f = ob.Bloom(k)
        f.index(ib, io)
        # all reads
        found, scanned, nfound = ctx.search_reads(qrs)
        exp, nexp = f.search(t, qb, qo)
        assert scanned == len(q_reads)
        assert nfound == nexp
        assert np.array_equal(found, exp)
        # with an active mask: inactive reads stay 0
        found, scanned, nfound = ctx.search_reads(qrs, util.bits_from_bools(active))
        exp, nexp = f.search(t, qb, qo, util.bits_from_bools(active))
        assert scanned == int(active.sum())
        assert nfound == nexp
        assert np.array_equal(found, exp)
        irs.close()
        qrs.close()


@pytest.mark.parametrize("k", [33, 34])
def test_wide_keys(commet, k):
    """k > 32: 64-bit keys, 4 / 8 GiB filter in HBM (reference default is k = 33)."""
    rng = np.random.default_rng(k)
    idx_reads = util.random_reads(rng, 300, 40, 150, n_rate=0.01)
    q_reads = util.related_reads(rng, idx_reads, 900, 1, 150, share=0.6, n_rate=0.01)
    ib, io = util.to_batch(idx_reads)
    qb, qo = util.to_batch(q_reads)
    with commet.Context(k=k, t=2) as ctx:
        irs = commet.ReadSet.from_files(ctx, [(ib, io)])
        qrs = commet.ReadSet.from_files(ctx, [(qb, qo)])
        ctx.filter_reset()
        fed = ctx.index_reads(irs)
        f = ob.Bloom(k)
        assert fed == f.index(ib, io)
        found, scanned, nfound = ctx.search_reads(qrs)
        exp, nexp = f.search(2, qb, qo)
        assert nfound == nexp and nexp > 0
        assert np.array_equal(found, exp)
        if k == 33:
            assert np.array_equal(ctx.export_filter_reference(), f.bytes())
        irs.close()
        qrs.close()


def _key_level_search(idx_reads, queries, k, t):
    """search_reads.h:45-83 over the CPU checker's KEYS (ok_keys_of_read: hash_key.h's add / rv_add per complete window) with the four
    lanes kept as Python sets — for k whose 2^(k-1)-byte filter the checker cannot allocate on a test host (k = 38: 128 GiB)"""
    lanes = [set(), set(), set(), set()]
    for r in idx_reads:
        keys, _ = ob.keys_of_read(r, k)
        for j in range(4):
            lanes[j].update(keys[:, j].tolist())
    out = np.zeros(len(queries), dtype=bool)
    for i, q in enumerate(queries):
        for rev in (False, True):
            keys, pos = ob.keys_of_read(q, k, reverse=rev)
            seen, next_end = 0, 0
            for (a, b, c, d), p in zip(keys.tolist(), pos.tolist()):
                if p < next_end:                       # hash.clear() after a hit: the next complete window ends k bases later
                    continue
                if a in lanes[0] and b in lanes[1] and c in lanes[2] and d in lanes[3]:
                    seen += 1
                    if seen >= t:
                        out[i] = True
                        break
                    next_end = p + k
            if out[i]:
                break
    return out


@pytest.mark.parametrize("k", [20, 33, 35, 36, 37, 38])
def test_largest_k_against_the_key_level_checker(commet, k):
    """k up to the largest the boundary takes (38: four planes of 2^38 bits = 128 GiB of HBM; the reference would need as much host
    RAM).  The key-level checker is pinned to the CPU checker's filter at k = 20 and 33 in this same test."""
    rng = np.random.default_rng(1000 + k)
    t = 2
    idx_reads = util.random_reads(rng, 250, 40, 170, n_rate=0.01)
    q_reads = util.related_reads(rng, idx_reads, 700, 1, 170, share=0.6, n_rate=0.01)
    ib, io = util.to_batch(idx_reads)
    qb, qo = util.to_batch(q_reads)
    want = _key_level_search(idx_reads, q_reads, k, t)
    assert 50 < want.sum() < len(q_reads)
    if k <= 33:
        f = ob.Bloom(k)
        f.index(ib, io)
        exp, _ = f.search(t, qb, qo)
        assert np.array_equal(util.bools_from_bits(exp, len(q_reads)), want)
    with commet.Context(k=k, t=t) as ctx:
        irs = commet.ReadSet.from_files(ctx, [(ib, io)])
        qrs = commet.ReadSet.from_files(ctx, [(qb, qo)])
        assert np.array_equal(irs.kmer_counts(), ob.kmer_counts(ib, io, k))
        ctx.filter_reset()
        fed = ctx.index_reads(irs)
        assert fed == int(ob.kmer_counts(ib, io, k).sum())
        found, scanned, nfound = ctx.search_reads(qrs)
        assert np.array_equal(util.bools_from_bits(found, len(q_reads)), want) and nfound == int(want.sum())
        tags, stats, info = ctx.index_and_search(irs, [qrs])       # the job path (one chunk)
        assert np.array_equal(util.bools_from_bits(tags[0], len(q_reads)), want) and stats[0]["shared"] == int(want.sum())
        assert info["n_chunks"] == 1 and info["kmers_indexed"] == fed


def test_uniform_and_ragged_layouts_agree(commet):
    """fixed-length sets take the offset-free addressing; results must not depend on it"""
    rng = np.random.default_rng(5)
    L = 100
    reads = util.random_reads(rng, 2000, L, L, n_rate=0.01)
    ragged = reads + [b"ACGT" * 7]                 # one odd read switches the set to explicit offsets
    k, t = 16, 2
    ib, io = util.to_batch(reads[:800])
    with commet.Context(k=k, t=t) as ctx:
        irs = commet.ReadSet.from_files(ctx, [(ib, io)])
        ctx.filter_reset()
        ctx.index_reads(irs)
        a = commet.ReadSet.from_files(ctx, [util.to_batch(reads)])
        b = commet.ReadSet.from_files(ctx, [util.to_batch(ragged)])
        fa, _, na = ctx.search_reads(a)
        fb, _, nb = ctx.search_reads(b)
        n = len(reads)
        assert np.array_equal(util.bools_from_bits(fa, n), util.bools_from_bits(fb, n))
        f = ob.Bloom(k)
        f.index(ib, io)
        exp, nexp = f.search(t, *util.to_batch(reads))
        assert np.array_equal(fa, exp) and na == nexp
        for r in (irs, a, b):
            r.close()


def test_empty_and_tiny_inputs(commet):
    with commet.Context(k=12, t=2) as ctx:
        e = commet.ReadSet.from_files(ctx, [(np.zeros(0, np.uint8), np.zeros(1, np.uint64))])
        assert e.num_reads == 0
        ctx.filter_reset()
        assert ctx.index_reads(e) == 0
        found, scanned, nfound = ctx.search_reads(e)
        assert scanned == 0 and nfound == 0 and found.size == 1
        one = commet.ReadSet.from_files(ctx, [util.to_batch([b"ACGTACGTACGTACGTACGTACGTACGT"])])
        ctx.index_reads(one)
        found, scanned, nfound = ctx.search_reads(one)
        assert (scanned, nfound) == (1, 1) and found[0] == 1
        e.close()
        one.close()


def test_errors_are_loud(commet):
    with pytest.raises(commet.CommetError):
        commet.Context(k=0)
    with pytest.raises(commet.CommetError):
        commet.Context(k=39)
    with commet.Context(k=10) as ctx:
        rs = commet.ReadSet(ctx, 10, 100)
        with pytest.raises(commet.CommetError):
            ctx.index_reads(rs)          # not finalized
        rs.add_file(*util.to_batch([b"ACGTACGTACGTAA"]))
        rs.finalize()
        with pytest.raises(commet.CommetError):
            ctx.index_reads(rs, 0, 5)    # out of range
        rs.close()


@pytest.mark.parametrize("k", [20, 21, 24, 26, 28])
def test_bucketed_index_matches_oracle(commet, k):
    """index_mode=2: the LDS-tile construction (index_part.hpp) must give the very same filter bits"""
    reads = _mixed_reads(300 + k, n=3000, hi=160)
    bases, offs = util.to_batch(reads)
    rng = np.random.default_rng(k)
    sel = rng.random(len(reads)) < 0.8
    with commet.Context(k=k, t=2) as ctx:
        ctx.set_option("index_mode", 2)
        rs = commet.ReadSet.from_files(ctx, [(bases, offs)])
        ctx.filter_reset()
        fed = ctx.index_reads(rs)
        f = ob.Bloom(k)
        assert fed == f.index(bases, offs)
        assert np.array_equal(ctx.export_filter_reference(), f.bytes())
        # range + select, then a second additive call on the rest: union must equal the oracle's union
        ctx.filter_reset()
        sb = util.bits_from_bools(sel)
        ctx.index_reads(rs, 100, 1900, sb)
        ctx.index_reads(rs, 2000, 1000, sb)
        f2 = ob.Bloom(k)
        s2 = sel.copy()
        s2[:100] = False
        f2.index(bases, offs, util.bits_from_bools(s2))
        assert np.array_equal(ctx.export_filter_reference(), f2.bytes())
        rs.close()


@pytest.mark.parametrize("k,uniform", [(30, True), (32, True), (32, False), (33, True)])
def test_bucketed_index_filter_bytes_at_large_k(commet, k, uniform):
    """the bucketed construction against the CPU checker's filter BYTES at the sizes the benchmark runs (k = 32: 2 GiB,
    k = 33: 4 GiB, the packed scatter2 and both hist / scatter1 item paths) — smaller k are compared in
    test_bucketed_index_matches_oracle; here the reads include hot buckets (poly-A, a tandem repeat)"""
    rng = np.random.default_rng(7 * k + int(uniform))
    L = 100
    reads = util.random_reads(rng, 150000, L, L, n_rate=0.002) if uniform else util.random_reads(rng, 150000, 40, 160, n_rate=0.002)
    reads += ([b"A" * L] * 3000 + [(b"ACG" * L)[:L]] * 2000) if uniform else ([b"A" * 130] * 3000 + [(b"ACG" * 50)[:117]] * 2000)
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    bases, offs = util.to_batch(reads)
    with commet.Context(k=k, t=2) as ctx:
        ctx.set_option("index_mode", 2)
        rs = commet.ReadSet.from_files(ctx, [(bases, offs)])
        ctx.filter_reset()
        fed = ctx.index_reads(rs)
        f = ob.Bloom(k)
        assert fed == f.index(bases, offs)
        got, want = ctx.export_filter_reference(), f.bytes()
        f.close()
        assert got.shape == want.shape and np.array_equal(got, want)
        rs.close()


@pytest.mark.parametrize("k,L", [(20, 20), (21, 100), (24, 37), (26, 150), (28, 64), (32, 100), (33, 101), (34, 250)])
def test_bucketed_index_uniform_length_fast_path(commet, k, L):
    """reads of one length and no selection take the arithmetic item path of hist / scatter1 (index_part.hpp, UNI):
    same filter as the planned path, the atomic kernel and the oracle; ranges that start / end anywhere"""
    rng = np.random.default_rng(1000 * k + L)
    reads = util.random_reads(rng, 20000, L, L, n_rate=0.01)
    reads[7] = b"N" * L                                     # no k-mer at all
    reads[8] = b"A" * L
    reads += [b"A" * L] * 700 + [(b"ACGT" * L)[:L]] * 500   # hot buckets
    bases, offs = util.to_batch(reads)
    q = util.related_reads(rng, reads[:4000], 6000, L, L, share=0.5)
    qb, qo = util.to_batch(q)
    out = []
    for mode, no_uni in ((2, 0), (2, 1), (1, 0)):
        with commet.Context(k=k, t=2) as ctx:
            ctx.set_option("index_mode", mode)
            ctx.set_option("part_no_uni", no_uni)
            rs = commet.ReadSet.from_files(ctx, [(bases, offs)])
            qs = commet.ReadSet.from_files(ctx, [(qb, qo)])
            ctx.filter_reset()
            fed = ctx.index_reads(rs, 3, len(reads) - 1004)
            fed += ctx.index_reads(rs, len(reads) - 1000, 1000)        # additive second call; reads 0-2 and one more skipped
            found, _, nfound = ctx.search_reads(qs)
            out.append((fed, nfound, found, ctx.export_filter_reference() if k <= 28 else None))
    for o in out[1:]:
        assert o[0] == out[0][0] and o[1] == out[0][1]
        assert np.array_equal(o[2], out[0][2])
        if k <= 28:
            assert np.array_equal(o[3], out[0][3])
    if k <= 28:
        sel = np.ones(len(reads), dtype=bool)
        sel[:3] = False
        sel[len(reads) - 1001] = False
        f = ob.Bloom(k)
        assert f.index(bases, offs, util.bits_from_bools(sel)) == out[0][0]
        assert np.array_equal(out[0][3], f.bytes())


@pytest.mark.parametrize("k", [20, 21, 24, 26, 28, 31, 32, 33, 34])
def test_bucketed_index_ragged_item_list(commet, k):
    """reads of many lengths take the chunk's item list in hist / scatter1 (index_part.hpp, LIST; round 6): same filter as the
    round planner (part_list = 1), the atomic kernel and the CPU checker — whole set, ranges that start / end anywhere, a selection
    bitmap, an additive second call; the reads include what the list must get right without knowing a read's length: reads whose
    length is a multiple of 32 (the next read's first windows must not look into them), reads shorter than k, reads of N only,
    non-ACGT bases at a read's ends, reads of 1 base, hot buckets"""
    rng = np.random.default_rng(4000 + k)
    reads = util.random_reads(rng, 12000, 1, 300, n_rate=0.004)
    reads += util.random_reads(rng, 3000, 32, 32, n_rate=0.0) + util.random_reads(rng, 3000, 64, 64, n_rate=0.001) + util.random_reads(rng, 2000, 96, 96)
    reads += util.random_reads(rng, 500, 128, 128) + util.random_reads(rng, 300, 400, 400, n_rate=0.001)
    reads += [b"N" * 77] * 50 + [b"A" * 130] * 400 + [(b"ACG" * 50)[:117]] * 300 + [b"C"] * 20 + [b"ACGT" * 8] * 200
    reads += [b"N" + r[1:-1] + b"N" for r in util.random_reads(rng, 500, 40, 160)]
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    n = len(reads)
    bases, offs = util.to_batch(reads)
    sel = rng.random(n) < 0.6
    sb = util.bits_from_bools(sel)
    q = util.related_reads(rng, reads[:4000], 6000, 30, 200, share=0.5)
    qb, qo = util.to_batch(q)
    out = []
    for mode, no_list in ((2, 0), (2, 1), (1, 0)):
        with commet.Context(k=k, t=2) as ctx:
            ctx.set_option("index_mode", mode)
            ctx.set_option("part_list", no_list)
            rs = commet.ReadSet.from_files(ctx, [(bases, offs)])
            qs = commet.ReadSet.from_files(ctx, [(qb, qo)])
            ctx.filter_reset()
            fed_all = ctx.index_reads(rs)
            whole = ctx.export_filter_reference() if k <= 28 else None
            found_all, _, n_all = ctx.search_reads(qs)
            ctx.filter_reset()
            fed = ctx.index_reads(rs, 5, n - 2005, sb)
            fed += ctx.index_reads(rs, n - 1999, 1999)            # additive second call, no selection; reads 0-4 and one more skipped
            found, _, nfound = ctx.search_reads(qs)
            out.append((fed_all, n_all, found_all, whole, fed, nfound, found, ctx.export_filter_reference() if k <= 28 else None))
    for o in out[1:]:
        for a, b in zip(o, out[0]):
            assert np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b
    if k <= 28:
        f = ob.Bloom(k)
        assert f.index(bases, offs) == out[0][0]
        assert np.array_equal(out[0][3], f.bytes())
        s2 = sel.copy()
        s2[:5] = False
        s2[n - 2000] = False
        s2[n - 1999:] = True
        f2 = ob.Bloom(k)
        assert f2.index(bases, offs, util.bits_from_bools(s2)) == out[0][4]
        assert np.array_equal(out[0][7], f2.bytes())


@pytest.mark.parametrize("k", [30, 32, 33])
def test_bucketed_index_equals_atomic_index_large_k(commet, k):
    """two-level radix geometry (k >= 26) incl. 64-bit keys; skewed input makes split tiles"""
    rng = np.random.default_rng(k)
    reads = util.random_reads(rng, 40000, 60, 120, n_rate=0.005)
    reads += [b"A" * 150] * 3000 + [b"ACGT" * 30] * 2000 + [b"T" * 100 + b"G" * 40] * 1500     # hot buckets
    bases, offs = util.to_batch(reads)
    q = util.related_reads(rng, reads[:5000], 8000, 40, 120, share=0.5)
    qb, qo = util.to_batch(q)
    out = []
    for mode in (1, 2):
        with commet.Context(k=k, t=2) as ctx:
            ctx.set_option("index_mode", mode)
            rs = commet.ReadSet.from_files(ctx, [(bases, offs)])
            qs = commet.ReadSet.from_files(ctx, [(qb, qo)])
            ctx.filter_reset()
            fed = ctx.index_reads(rs)
            found, _, nfound = ctx.search_reads(qs)
            out.append((fed, nfound, found, ctx.export_filter_reference() if k <= 32 else None))
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    assert np.array_equal(out[0][2], out[1][2])
    if k <= 32:
        assert np.array_equal(out[0][3], out[1][3])
        f = ob.Bloom(k)
        f.index(bases, offs)
        assert np.array_equal(out[1][3], f.bytes())


def test_parallel_host_ingest_equals_batch_upload(commet, tmp_path, monkeypatch):
    """commet_readset_from_fasta (multi-threaded parse, pieces uploaded out of order at explicit positions)
    must give the same resident set as the in-order batch API: FASTA (multi-line, > 8 MB so that it is cut
    into pieces), FASTQ and gzip files in one set."""
    monkeypatch.setenv("COMMET_INGEST_THREADS", "3")
    rng = np.random.default_rng(11)
    big = util.random_reads(rng, 90000, 60, 140, n_rate=0.01)
    big[1000] = b"ACGT" * 3                                  # short read
    small = util.random_reads(rng, 3000, 20, 90, n_rate=0.02)
    fq = util.random_reads(rng, 5000, 30, 120, n_rate=0.01)
    paths = [str(tmp_path / "big.fa"), str(tmp_path / "small.fa.gz"), str(tmp_path / "x.fq"), str(tmp_path / "y.fq.gz")]
    util.write_reads(paths[0], big, "fa", rng=rng, multiline=True)
    util.write_reads(paths[1], small, "fa.gz", rng=rng, multiline=True)
    util.write_reads(paths[2], fq, "fq")
    util.write_reads(paths[3], fq[::-1], "fq.gz")
    assert max(len(open(paths[0], "rb").read()), 0) > (8 << 20)
    k, t = 25, 2
    with commet.Context(k=k, t=t) as ctx:
        a = commet.ReadSet.from_fasta(ctx, paths)
        parsed = [util.parse_reads(p) for p in paths]
        b = commet.ReadSet.from_files(ctx, [util.to_batch(r) for r in parsed])
        assert a.file_reads() == b.file_reads() == [len(r) for r in parsed]
        assert np.array_equal(a.kmer_counts(), b.kmer_counts())
        idx = commet.ReadSet.from_files(ctx, [util.to_batch(big[:20000] + fq[:2000])])
        ta, sa, _ = ctx.index_and_search(idx, [a])
        tb, sb, _ = ctx.index_and_search(idx, [b])
        assert np.array_equal(ta[0], tb[0]) and sa[0]["shared"] == sb[0]["shared"] > 20000
        # and as an index set (chunk plan uses the per-read k-mer counts and the file spans)
        q = commet.ReadSet.from_files(ctx, [util.to_batch(big[:5000])])
        t1, s1, i1 = ctx.index_and_search(a, [q])
        t2, s2, i2 = ctx.index_and_search(b, [q])
        assert np.array_equal(t1[0], t2[0]) and i1["n_chunks"] == i2["n_chunks"] and i1["kmers_indexed"] == i2["kmers_indexed"]


def test_staging_api_device_pack_equals_host_pack():
    """commet_readset_stage_acquire / _commit (ASCII in pinned buffers, pack_reads_kernel on the device) against
    commet_readset_append (2-bit packed by the host threads): same k-mer counts, same filter, same search result — also
    for a set that mixes the two ways file by file"""
    import commet_amd
    rng = np.random.default_rng(77)
    reads = util.random_reads(rng, 3000, 1, 400, n_rate=0.02, lower_rate=0.2, other_rate=0.01) + [b""] * 0
    half = len(reads) // 2
    fa, fb = util.to_batch(reads[:half]), util.to_batch(reads[half:])
    q = util.related_reads(rng, reads, 2500, 30, 300, share=0.6)
    qb, qo = util.to_batch(q)
    for k in (12, 25, 33):
        with commet_amd.Context(k=k, t=2) as ctx:
            host = commet_amd.ReadSet.from_files(ctx, [fa, fb])
            dev = commet_amd.ReadSet(ctx, len(reads), int(fa[1][-1] + fb[1][-1]))
            dev.add_file_staged(*fa)
            dev.add_file_staged(*fb)
            dev.finalize()
            mixed = commet_amd.ReadSet(ctx, len(reads), int(fa[1][-1] + fb[1][-1]))
            mixed.add_file(*fa)
            mixed.add_file_staged(*fb)
            mixed.finalize()
            kc = host.kmer_counts()
            assert np.array_equal(dev.kmer_counts(), kc) and np.array_equal(mixed.kmer_counts(), kc)
            assert dev.file_reads() == host.file_reads() == mixed.file_reads() == [half, len(reads) - half]
            qs = commet_amd.ReadSet.from_files(ctx, [(qb, qo)])
            ref = ctx.index_and_search(host, [qs])
            for other in (dev, mixed):
                got = ctx.index_and_search(other, [qs])
                assert np.array_equal(got[0][0], ref[0][0]) and got[1][0]["shared"] == ref[1][0]["shared"] > 200
            back = ctx.index_and_search(qs, [dev])            # the device-packed set as the search set
            assert np.array_equal(back[0][0], ctx.index_and_search(qs, [host])[0][0])
