"""Opt-in code that ships: query lists above 4 GiB (option query_list_max_mb / COMMET_QUERY_LIST_MAX_GB, capi/search_dispatch.hpp
tiled_ok) — a set that large gets its list for its SECOND eligible scan (the first keeps the gather kernels), and both scans must
give the reference's bits.  A 16 M-read search set (est. list 4.7 GB > the 4 GiB default cap) against a 4 M-read index set (one
chunk filter): the first job takes search_kernel, the second builds the list and takes the tiled pair of kernels; the tags of
both are equal, and a 20 000-read sample is replayed on the CPU checker.  The same list once more with the cap back at its default
but the list's memory set aside beforehand (commet_readset_reserve_cache: the N x N driver's way)."""
import numpy as np
import pytest

import oracle_binding as ob
import oracle_pool
import util

pytestmark = pytest.mark.gpu


def test_a_query_list_above_4_gib_is_built_for_the_second_scan_and_gives_the_same_bits(tmp_path):
    import commet_amd
    from commet_amd import synth
    k, t, L, n_i, n_q = 32, 2, 100, 4_000_000, 16_000_000
    b0, o0 = synth.synth_set(0, n_i, L)
    b1, o1 = synth.synth_set(1, n_q, L)                  # its first 25 % are copies of set 0's reads (1 M of them of the indexed ones)
    with commet_amd.Context(k=k, t=t) as ctx:
        irs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
        qrs = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
        kc = irs.kmer_counts()
        chunks = oracle_pool.chunks_from_counts(kc, ob.max_kmer(k))
        assert len(chunks) == 1
        # the CPU checker's replay of a sample runs beside the GPU jobs (a worker process: one 2 GiB filter from 2.8e8 k-mers)
        rng = np.random.default_rng(5)
        smp = np.unique(np.concatenate([rng.choice(n_i, 8000, replace=False), n_i + rng.choice(n_q - n_i, 12000, replace=False)]))
        sb = np.ascontiguousarray(b1.reshape(n_q, L)[smp]).reshape(-1)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(1) as pool:
            fut = pool.submit(oracle_pool.search_sample_over_chunks, str(tmp_path / "orc"), "ll", b0, L, chunks, k, t, sb, 1)
            runs = []
            def scan_once():
                ctx.set_option("kernel_timing", 1)
                tags, stats, info = ctx.index_and_search(irs, [qrs])
                runs.append(dict(tags=tags[0], shared=stats[0]["shared"], kernels=set(ctx.kernel_times()), cache=qrs.cache_bytes))
                ctx.set_option("kernel_timing", 0)

            for scan in range(3):
                if scan == 1:
                    ctx.set_option("query_list_max_mb", 16 << 10)          # lists of up to 16 GiB from here on
                scan_once()
            # the other way to a list above the cap: its memory set aside beforehand (commet_readset_reserve_cache — what the N x N driver's
            # loader thread does for the sets it searches again and again), the cap itself back at its default
            qrs.drop_cache()
            ctx.set_option("query_list_max_mb", 4096)
            scan_once()                                                     # runs[3]: the cap applies again
            est, filed = qrs.cache_estimate(), commet_amd.device_cache_bytes(0)
            qrs.reserve_cache()
            reserved = commet_amd.device_cache_bytes(0) - filed
            scan_once()                                                     # runs[4]: the list, although the cap says no
            want, fed = fut.result()
    assert est > n_q * 37 * 6 and reserved >= est                            # (+ the context's result buffer when it had to grow)
    assert "search_kernel" in runs[3]["kernels"] and runs[3]["cache"] == 0
    assert {"tq_probe_kernel", "tq_replay_kernel", "tq_fill_kernel"} <= runs[4]["kernels"] and runs[4]["cache"] > (3 << 30)
    # default cap: a list of an estimated 16 M x 37 x 8 B = 4.7 GB is not built: the gather kernel
    assert "search_kernel" in runs[0]["kernels"] and "tq_replay_kernel" not in runs[0]["kernels"] and runs[0]["cache"] == 0
    # cap raised: the set's FIRST eligible scan still gathers (a set scanned once must not pay the list) ...
    assert "search_kernel" in runs[1]["kernels"] and "tq_replay_kernel" not in runs[1]["kernels"] and runs[1]["cache"] == 0
    # ... its second one builds the list (16 M x 37 records of 6 bytes + the tile tables) and takes the tiled search
    assert {"tq_probe_kernel", "tq_replay_kernel", "tq_fill_kernel"} <= runs[2]["kernels"] and "search_kernel" not in runs[2]["kernels"]
    assert runs[2]["cache"] > n_q * 37 * 6 > (3 << 30)
    for r in runs[1:]:
        assert np.array_equal(r["tags"], runs[0]["tags"]) and r["shared"] == runs[0]["shared"]
    commet_amd.device_cache_trim()                                            # (the 4 GB of the list go back to the driver: the next tests' child processes share the card)
    assert fed == [int(kc[a:e].sum()) for a, e in chunks]
    got = util.bools_from_bits(runs[2]["tags"], n_q)[smp]
    assert np.array_equal(got, want)
    assert 6000 < int(want.sum()) < 8000 + 200                        # the copies among the sampled reads, most of them found
