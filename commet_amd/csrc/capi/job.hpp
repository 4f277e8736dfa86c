// capi/job.hpp — commet_index_reads, commet_search_reads and the chunk loop of the tool on resident sets, commet_index_and_search (index_and_search.cpp:241-277)
// (a part of the one translation unit capi.hip: included there, in order, after the kernels and state.hpp)
#pragma once

extern "C" {

int commet_index_reads(commet_ctx *c, const commet_readset *rs, uint64_t first, uint64_t count,
                       const uint8_t *select_bits, uint64_t *kmers_fed)
{
    if (!rs->finalized) return fail("read set not finalized");
    if (rs->ctx != c) return fail("read set belongs to another context");
    if (first > rs->n_reads || count > rs->n_reads - first) return fail("index range out of bounds");
    HIP_OK(hipSetDevice(c->device));
    const uint64_t *d_sel = nullptr;
    if (select_bits) {
        if (upload_bits(c, rs->d_sel, select_bits, rs->n_reads)) return 1;
        d_sel = rs->d_sel;
    }
    unsigned long long *d_fed = nullptr;
    if (kmers_fed) {
        HIP_OK(hipMemsetAsync(c->d_counters, 0, sizeof(unsigned long long), c->stream));
        d_fed = c->d_counters;
    }
    HIP_OK(hipEventRecord(c->ev_i0, c->stream));
    // exact k-mer count of the launch (host copy of the per-read counts): lets the bucketed path run
    if (host_counts(rs)) return 1;
    uint64_t kmers = 0;
    for (uint64_t r = first; r < first + count; ++r)
        if (!select_bits || bit_at(select_bits, r)) kmers += rs->h_kcnt[r];
    if (launch_index(c, rs, first, count, d_sel, d_fed, kmers, false)) return 1;
    HIP_OK(hipEventRecord(c->ev_i1, c->stream));
    c->have_index_ev = true;
    if (kmers_fed) {
        HIP_OK(hipMemcpyAsync(c->h_counters, c->d_counters, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
        *kmers_fed = c->h_counters[0];
    }
    return 0;
}

int commet_search_reads(commet_ctx *c, const commet_readset *rs, const uint8_t *active_bits, uint8_t *found_bits,
                        uint64_t *n_scanned, uint64_t *n_found)
{
    if (!rs->finalized) return fail("read set not finalized");
    if (rs->ctx != c) return fail("read set belongs to another context");
    HIP_OK(hipSetDevice(c->device));
    const uint64_t *d_sel = nullptr;
    if (active_bits) {
        if (upload_bits(c, rs->d_sel, active_bits, rs->n_reads)) return 1;
        d_sel = rs->d_sel;
    }
    HIP_OK(hipMemsetAsync(c->d_counters, 0, 2 * sizeof(unsigned long long), c->stream));
    HIP_OK(hipMemsetAsync(rs->d_found, 0, bitmap_words(rs->n_reads) * 8, c->stream));
    HIP_OK(hipEventRecord(c->ev_s0, c->stream));
    if (launch_search(c, rs, d_sel, nullptr, rs->d_found, c->d_counters)) return 1;
    HIP_OK(hipEventRecord(c->ev_s1, c->stream));
    c->have_search_ev = true;
    HIP_OK(hipMemcpyAsync(c->h_counters, c->d_counters, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    if (found_bits)
        HIP_OK(hipMemcpyAsync(found_bits, rs->d_found, bitmap_bytes_host(rs->n_reads), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    if (n_scanned) *n_scanned = c->h_counters[0];
    if (n_found) *n_found = c->h_counters[1];
    return 0;
}

/* ---- the chunk loop (index_and_search.cpp:241-277) ------------------------ */

int commet_index_and_search(commet_ctx *c, const commet_readset *index_rs, const uint8_t *index_select, int n_search,
                            const commet_readset *const *search_rs, const uint8_t *const *search_select,
                            uint8_t *const *tags_out, commet_pair_stats *stats, commet_job_info *info)
{
    const auto wall0 = std::chrono::steady_clock::now();
    // host-side phase times of the call (COMMET_JOB_VERBOSE: one line per call on stderr)
    const bool job_verbose = c->job_verbose;
    auto lap_t = wall0;
    double ph_plan = 0, ph_upload = 0, ph_launch = 0, ph_wait = 0;
    auto lap = [&](double &acc) {
        const auto now = std::chrono::steady_clock::now();
        acc += std::chrono::duration<double, std::milli>(now - lap_t).count();
        lap_t = now;
    };
    if (!index_rs->finalized) return fail("index read set not finalized");
    if (index_rs->ctx != c) return fail("index read set belongs to another context");
    for (int s = 0; s < n_search; ++s) {
        if (!search_rs[s]->finalized) return fail("search read set %d not finalized", s);
        if (search_rs[s]->ctx != c) return fail("search read set %d belongs to another context", s);
        if (search_rs[s] == index_rs) return fail("a set cannot be searched against itself in one call");
        for (int q = 0; q < s; ++q)
            if (search_rs[q] == search_rs[s]) return fail("search read set listed twice");
    }
    HIP_OK(hipSetDevice(c->device));
    // the sets of this call keep their cached query lists whatever memory pressure another thread meets meanwhile
    struct InJob {
        commet_ctx *c;
        const commet_readset *index_rs;
        const commet_readset *const *srs;
        int n;
        void mark(bool v) const
        {
            std::lock_guard<std::mutex> lk(c->ql_mu);
            index_rs->in_job = v;
            for (int i = 0; i < n; ++i) srs[i]->in_job = v;
        }
        InJob(commet_ctx *c_, const commet_readset *i_, const commet_readset *const *s_, int n_) : c(c_), index_rs(i_), srs(s_), n(n_) { mark(true); }
        ~InJob() { mark(false); }
    } in_job(c, index_rs, search_rs, n_search);

    // an input filter that selects every read is no filter (Commet.py passes all-ones bvs when nothing was filtered)
    if (index_select && all_ones(index_select, index_rs->n_reads)) index_select = nullptr;
    // host plan: chunks of the index set, visited reads of each search set
    const uint64_t max_kmer = commet_max_kmer(c);
    // The plan is made from per-block k-mer sums computed on the device, where kcnt lives; the host walks only the
    // blocks in which a chunk starts or ends and fetches just those blocks' counts: no per-read loop over the set and
    // no host copy of its counts (a selection bitmap, when there is one, is uploaded first for the kernel to use).
    std::vector<uint64_t> blk_sums;
    if (index_rs->n_reads && plan_blocks_ok(index_rs->files, index_select, index_rs->empty_reads, max_kmer)) {
        const uint64_t nblk = (index_rs->n_reads + PLAN_BLOCK_READS - 1) / PLAN_BLOCK_READS;
        if (c->plansum_cap < nblk) {
            HIP_OK(hipStreamSynchronize(c->stream));
            (void) dm_free(c->d_plansum);
            c->d_plansum = nullptr;
            c->plansum_cap = 0;
            HIP_OK(dev_alloc(c, (void **) &c->d_plansum, nblk * sizeof(unsigned long long), true));
            c->plansum_cap = nblk;
        }
        if (index_select && upload_bits(c, index_rs->d_sel, index_select, index_rs->n_reads)) return 1;
        {
            KScope ks(c, "block_kmer_sums_kernel", c->stream);
            COMMET_LAUNCH(block_kmer_sums_kernel, dim3((unsigned) nblk), dim3(256), 0, c->stream, index_rs->d_kcnt,
                               index_select ? index_rs->d_sel : nullptr, index_rs->n_reads, c->d_plansum);
        }
        HIP_OK(hipGetLastError());
        blk_sums.resize(nblk);
        HIP_OK(hipMemcpyAsync(blk_sums.data(), c->d_plansum, nblk * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    }
    // counts of one block of reads, fetched on demand (or taken from the host copy when somebody made one)
    std::vector<uint32_t> kblock(PLAN_BLOCK_READS);
    uint64_t kblock_no = ~0ull;
    bool kfetch_failed = false;
    auto kcnt_of = [&](uint64_t q) -> uint32_t {
        if (index_rs->have_host_counts) return index_rs->h_kcnt[q];
        const uint64_t blk = q / PLAN_BLOCK_READS;
        if (blk != kblock_no) {
            const uint64_t lo = blk * PLAN_BLOCK_READS, cnt = std::min<uint64_t>(PLAN_BLOCK_READS, index_rs->n_reads - lo);
            if (hipMemcpy(kblock.data(), index_rs->d_kcnt + lo, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) kfetch_failed = true;
            kblock_no = blk;
        }
        return kblock[q % PLAN_BLOCK_READS];
    };
    if (blk_sums.empty() && host_counts(index_rs)) return 1;   // the other planners read the counts on the host
    const IndexPlan plan = !blk_sums.empty() ? plan_index_blocks(index_select, kcnt_of, index_rs->n_reads, max_kmer,
                                                                 blk_sums.data(), PLAN_BLOCK_READS)
                           : plan_fast_ok(index_rs->files, index_select, index_rs->empty_reads, max_kmer)
                               ? plan_index_fast(index_rs->h_kprefix, index_rs->n_reads, max_kmer)
                           : (index_select && index_rs->empty_reads.empty())
                               ? plan_index_select(index_rs->files, index_select, index_rs->h_kcnt.data(), index_rs->n_reads, max_kmer)
                               : plan_index(index_rs->files, index_select, index_rs->empty_reads, index_rs->h_kcnt.data(),
                                            index_rs->n_reads, max_kmer);
    if (kfetch_failed) return fail("k-mer count fetch failed: %s", hipGetErrorString(hipGetLastError()));
    std::vector<uint64_t> visited(n_search, 0);
    std::vector<std::vector<uint8_t>> vis(n_search);
    std::vector<char> all_visited(n_search, 0);   // every read of the set is visited: the kernels take a null bitmap
    lap(ph_plan);
    // a dense plan indexes whole read ranges: no bitmap needed on the device
    if (!plan.dense && upload_bits(c, index_rs->d_sel, plan.indexed_bits.data(), index_rs->n_reads)) return 1;
    // a selection on a fixed-length set (Commet.py's J2 / J3 jobs): the selected reads' numbers as a list, so that the
    // bucketed build walks them arithmetically (index_part.hpp, sel_ids_kernel); chunk j's reads are the next n_reads of the list
    const uint32_t *d_ids = nullptr;
    std::vector<uint64_t> chunk_pos;
    uint64_t ids_expected = ~0ull;
    if (!plan.dense && index_rs->uniform_len != 0 && !c->part_no_uni && plan.indexed_reads && c->index_mode != 1) {
        const uint64_t n_words = bitmap_words(index_rs->n_reads), nb = (n_words + IDS_BLOCK_WORDS - 1) / IDS_BLOCK_WORDS;
        bool ok = true;
        if (c->ids_cap < plan.indexed_reads || c->idblk_cap < nb + 1) {
            HIP_OK(hipStreamSynchronize(c->stream));
            (void) dm_free(c->d_ids), (void) dm_free(c->d_idblk);
            c->d_ids = c->d_idblk = nullptr, c->ids_cap = c->idblk_cap = 0;
            const uint64_t cap = std::max<uint64_t>(plan.indexed_reads, index_rs->n_reads / 2);   // (grown rarely)
            ok = dev_alloc(c, (void **) &c->d_ids, cap * sizeof(uint32_t), true) == hipSuccess &&
                 dev_alloc(c, (void **) &c->d_idblk, (nb + 1) * sizeof(uint32_t), true) == hipSuccess;
            if (ok) c->ids_cap = cap, c->idblk_cap = nb + 1;
            else (void) hipGetLastError();              // no room: the round planner walks the bitmap, as before
        }
        if (ok) {
            KScope ks(c, "sel_ids_kernels", c->stream);
            COMMET_LAUNCH(sel_count_kernel, dim3((unsigned) nb), dim3(64), 0, c->stream, index_rs->d_sel, n_words, c->d_idblk);
            COMMET_LAUNCH(sel_scan_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_idblk, (uint32_t) nb);
            COMMET_LAUNCH(sel_ids_kernel, dim3((unsigned) nb), dim3(64), 0, c->stream, index_rs->d_sel, n_words, c->d_idblk, c->d_ids);
            HIP_OK(hipGetLastError());
            // (the list must hold exactly the plan's indexed reads: checked when the job's stream is next synchronised)
            c->h_counters[N_COUNTERS - 1] = ~0ull;
            HIP_OK(hipMemcpyAsync(&c->h_counters[N_COUNTERS - 1], c->d_idblk + nb, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            ids_expected = plan.indexed_reads;
            d_ids = c->d_ids;
            uint64_t at = 0;
            for (const Chunk &ch : plan.chunks) chunk_pos.push_back(at), at += ch.n_reads;
        }
    }
    lap(ph_upload);
    for (int s = 0; s < n_search; ++s) {
        const commet_readset *rs = search_rs[s];
        const uint8_t *ssel = search_select ? search_select[s] : nullptr;
        if (ssel && all_ones(ssel, rs->n_reads)) ssel = nullptr;
        all_visited[s] = plan_fast_ok(rs->files, ssel, rs->empty_reads, 1);
        if (all_visited[s]) visited[s] = rs->n_reads;   // == plan_search_fast, whose bitmap nobody would read
        else
            vis[s] = (ssel && rs->empty_reads.empty()) ? plan_search_select(rs->files, ssel, rs->n_reads, &visited[s])
                                                       : plan_search(rs->files, ssel, rs->empty_reads, rs->n_reads, &visited[s]);
        lap(ph_plan);
        if (!all_visited[s] && upload_bits(c, rs->d_sel, vis[s].data(), rs->n_reads)) return 1;
        HIP_OK(hipMemsetAsync(rs->d_tags, 0, bitmap_words(rs->n_reads) * 8, c->stream));
        lap(ph_upload);
    }
    HIP_OK(hipStreamSynchronize(c->stream));   // the host bit arrays above are pageable
    lap(ph_upload);

    // per (chunk, set) counters {scanned, found}
    const uint64_t n_chunks = plan.chunks.size();
    const uint64_t n_cnt = 2 * n_chunks * (uint64_t) n_search + 1;   // last slot: probe counter
    std::vector<unsigned long long> h_cnt(n_cnt, 0);
    if (c->jobcnt_cap < n_cnt) {   // kept between calls: hipMalloc / hipFree per job cost more than the counters' kernels
        HIP_OK(hipStreamSynchronize(c->stream));
        (void) dm_free(c->d_jobcnt);
        c->d_jobcnt = nullptr;
        c->jobcnt_cap = 0;
        HIP_OK(dev_alloc(c, (void **) &c->d_jobcnt, std::max<uint64_t>(n_cnt, 64) * sizeof(unsigned long long), true));
        c->jobcnt_cap = std::max<uint64_t>(n_cnt, 64);
    }
    unsigned long long *const d_cnt = c->d_jobcnt;
    HIP_OK(hipMemsetAsync(d_cnt, 0, n_cnt * sizeof(unsigned long long), c->stream));

    // device timing: one event pair around all index work and one around all
    // search work would overlap; instead accumulate per phase with event pairs
    // on the (in-order) stream.
    std::vector<hipEvent_t> evs;
    auto new_event = [&](hipEvent_t *e) -> int {
        HIP_OK(hipEventCreate(e));
        evs.push_back(*e);
        return 0;
    };
    // the many-small-chunks regime: the chunk filters of a group live bit-sliced in one set of tables (slice_search.hpp)
    int slice_gw = slice_words(c, n_chunks);
    // no room for the staging planes / tables of that regime: the job takes the slot loop below (slower, same bits)
    if (slice_gw && ensure_slice_buffers(c, c->slice_wide == 1 || (c->slice_wide == 0 && n_chunks <= 256) ? slice_gw : 8, n_chunks)) {
        (void) hipGetLastError();
        slice_gw = 0;
    }
    const bool timed = (info != nullptr || stats != nullptr) &&
                       (slice_gw ? (n_chunks / (32 * slice_gw) + 1) * (uint64_t) (n_search + 2) : n_chunks * (uint64_t) (n_search + 4)) <= 16384;
    std::vector<hipEvent_t> e_idx0, e_idx1, e_zero0, e_zero1;
    std::vector<std::vector<hipEvent_t>> e_set(n_search);   // end of set s's search, per chunk
    uint64_t n_index_launches = 0, n_search_launches = 0;
    unsigned long long *d_probes = c->count_probes ? d_cnt + (n_cnt - 1) : nullptr;

    int rc = 0;
    // chunks are taken in groups of up to `chunk_group`: their filters are built into separate slots and every
    // search set is scanned ONCE per group (search_group_kernel) instead of once per chunk
    int group_cap = (c->k >= 2) ? std::max(1, std::min(8, c->chunk_group)) : 1;
    if (n_chunks < 2) group_cap = 1;
    if (group_cap > 4) {   // more than four filters per pass: every search set must qualify for the register-mask kernel
        bool ok8 = n_chunks > 4;
        for (int s = 0; s < n_search && ok8; ++s) ok8 = group8_ok(c, search_rs[s]);
        if (!ok8) group_cap = 4;
    }
    if (slice_gw) {
        std::vector<SliceChunk> hc(n_chunks);
        for (uint64_t i = 0; i < n_chunks; ++i) {
            const Chunk &ch = plan.chunks[i];
            hc[i].first = ch.first;
            hc[i].count = ch.n_reads ? ch.last - ch.first + 1 : 0;
        }
        WidePlan wide = wide_plan(c, n_chunks, slice_gw);
        if (wide.nw && ensure_wide_tables(c, wide)) wide = WidePlan();   // no room for the wide tables: groups of 256 chunks as before
        if (ensure_slice_buffers(c, wide.nw ? 8 : slice_gw, n_chunks)) rc = 1;   // (sized above already; a wide plan that fell back may need less)
        if (!rc && hipMemcpy(c->d_slice_chunks, hc.data(), n_chunks * sizeof(SliceChunk), hipMemcpyHostToDevice) != hipSuccess)
            rc = fail("chunk descriptor upload failed");
        // Wide rows or narrow tables?  The wide pass looks at EVERY chunk filter for every read; the narrow tables take 256
        // chunks per pass and skip, in later passes, the reads that earlier ones have found — 2.5x the cost per chunk and
        // read (configs[4]: 8.3 s against 2.6 s), but when most reads are found early there is little left to pay it on
        // (10 M x 100 bp reads, t = 2: k = 20 narrow 628 ms / wide 850 ms, k = 18 664 / 1391, k = 16 649 / 1709 — random
        // reads share that many short k-mers — but k = 22 491 / 384, k = 24 307 / 256).  In auto mode the first 64 chunk
        // filters are therefore searched with the narrow tables against a sample of every search set (one 64-read word in 128 of a large set); with
        // p = the share of them that a group of 256 chunks would find at that rate, the reads still unfound after g groups
        // are taken as (1 - p)^g of the set, a narrow pass is priced at 3.7x a wide one per chunk and read (the largest
        // ratio measured: reads that are found leave the narrow kernel early, too), and the cheaper plan runs.  The probe's
        // reads are searched for real (tags and counters): whichever plan follows skips the found ones and finds nothing
        // new in those chunks for the others.
        if (!rc && wide.nw && c->slice_wide == 0) {
            const int g0 = (int) std::min<uint64_t>(64, n_chunks);
            if (launch_slice_build(c, index_rs, plan.dense ? nullptr : index_rs->d_sel, 0, g0, 2)) rc = 1;
            n_index_launches += 2;
            uint64_t sampled = 0;
            std::vector<uint64_t> smp;
            for (int s = 0; s < n_search && !rc; ++s) {
                const commet_readset *rs = search_rs[s];
                if (!rs->n_reads) continue;
                const uint64_t nw64 = bitmap_words(rs->n_reads);
                smp.assign(nw64, 0);
                const uint64_t *vw = all_visited[s] ? nullptr : (const uint64_t *) vis[s].data();   // (n/8+1 bytes: the last word may be partial)
                const uint64_t stride = rs->n_reads >= (4ull << 20) ? 128 : rs->n_reads >= (1ull << 20) ? 32 : 8;   // >= ~16 k sampled reads
                // (a block of the kernel is 4 words: only the blocks that hold a sampled word are launched)
                for (uint64_t w = 0; w < nw64; w += stride) {
                    uint64_t bits = ~0ull;
                    if (vw) {
                        bits = 0;
                        const uint64_t nbytes = bitmap_bytes_host(rs->n_reads), o = w * 8;
                        memcpy(&bits, vis[s].data() + o, (size_t) std::min<uint64_t>(8, nbytes > o ? nbytes - o : 0));
                    }
                    if (w * 64 >= rs->n_reads) bits = 0;                                              // (bitmaps have a spare word)
                    else if (rs->n_reads - w * 64 < 64) bits &= (1ull << (rs->n_reads - w * 64)) - 1ull;   // reads past the end
                    smp[w] = bits;
                    sampled += (uint64_t) __builtin_popcountll(bits);
                }
                if (hipMemcpyAsync(rs->d_found, smp.data(), nw64 * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                    hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail("probe bitmap upload failed"); break; }   // (smp is reused)
                if (launch_search_sliced(c, rs, g0, 2, rs->d_found, rs->d_tags, d_cnt + 2 * (uint64_t) s, (uint32_t) (2 * n_search), (uint32_t) (stride / 4))) { rc = 1; break; }
                ++n_search_launches;
            }
            std::vector<unsigned long long> pc((size_t) 2 * g0 * n_search);
            if (!rc && (hipMemcpyAsync(pc.data(), d_cnt, pc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                        hipStreamSynchronize(c->stream) != hipSuccess)) rc = fail("probe counter copy failed");
            if (!rc) {
                uint64_t found = 0;
                for (size_t i = 1; i < pc.size(); i += 2) found += pc[i];
                const double p0 = sampled ? std::min(1.0, (double) found / (double) sampled) : 0.0;   // found in g0 chunks
                const double pf = 1.0 - std::pow(1.0 - p0, 256.0 / (double) g0);                       // ... in a group of 256, at that rate
                const uint64_t groups = (n_chunks + 255) / 256;
                double left = 1.0, narrow_cost = 0.0;
                for (uint64_t gi = 0; gi < groups; ++gi) narrow_cost += 3.7 * 256.0 * left, left *= 1.0 - pf;
                if (narrow_cost < (double) n_chunks) wide = WidePlan();   // most reads are found early: the narrow tables, group by group
            }
        }
        // wide rows: the filters of a pass's chunks (all of them when the tables fit) are built 256 at a time into their
        // columns of the rows, then every search set is scanned ONCE per pass
        for (uint64_t c0 = 0; wide.nw && c0 < n_chunks && !rc; c0 += wide.chunks_per_pass) {
            const uint64_t c1 = std::min<uint64_t>(n_chunks, c0 + wide.chunks_per_pass);
            hipEvent_t a = nullptr, b = nullptr;
            if (timed) {
                if (new_event(&a) || new_event(&b)) { rc = 1; break; }
                (void) hipEventRecord(a, c->stream);
            }
            for (uint64_t ci = c0; ci < c1 && !rc; ci += 256) {
                const int g = (int) std::min<uint64_t>(256, c1 - ci);
                if (launch_slice_build(c, index_rs, plan.dense ? nullptr : index_rs->d_sel, ci, g, 8, c->wide_tables, wide.rw,
                                       (uint32_t) ((ci - c0) / 256 * WIDE_GROUP_WORDS))) rc = 1;
                n_index_launches += 2;
            }
            if (rc) break;
            if (timed) {
                (void) hipEventRecord(b, c->stream);
                e_idx0.push_back(a);
                e_idx1.push_back(b);
            }
            for (int s = 0; s < n_search && !rc; ++s) {
                const commet_readset *rs = search_rs[s];
                if (launch_search_wide(c, rs, wide, (int) (c1 - c0), all_visited[s] ? nullptr : rs->d_sel, rs->d_tags,
                                       d_cnt + 2 * (c0 * n_search + s), (uint32_t) (2 * n_search))) { rc = 1; break; }
                if (rs->n_reads) ++n_search_launches;
                if (timed) {
                    hipEvent_t d = nullptr;
                    if (new_event(&d)) { rc = 1; break; }
                    (void) hipEventRecord(d, c->stream);
                    e_set[s].push_back(d);
                }
            }
        }
        const uint64_t G = 32ull * slice_gw;
        for (uint64_t ci = 0; ci < n_chunks && !rc && !wide.nw; ci += G) {
            const int g = (int) std::min<uint64_t>(G, n_chunks - ci);
            hipEvent_t a = nullptr, b = nullptr;
            if (timed) {
                if (new_event(&a) || new_event(&b)) { rc = 1; break; }
                (void) hipEventRecord(a, c->stream);
            }
            if (launch_slice_build(c, index_rs, plan.dense ? nullptr : index_rs->d_sel, ci, g, slice_gw)) { rc = 1; break; }
            n_index_launches += 2;
            if (timed) {
                (void) hipEventRecord(b, c->stream);
                e_idx0.push_back(a);
                e_idx1.push_back(b);
            }
            for (int s = 0; s < n_search && !rc; ++s) {
                const commet_readset *rs = search_rs[s];
                if (launch_search_sliced(c, rs, g, slice_gw, all_visited[s] ? nullptr : rs->d_sel, rs->d_tags,
                                         d_cnt + 2 * (ci * n_search + s), (uint32_t) (2 * n_search))) { rc = 1; break; }
                if (rs->n_reads) ++n_search_launches;
                if (timed) {
                    hipEvent_t d = nullptr;
                    if (new_event(&d)) { rc = 1; break; }
                    (void) hipEventRecord(d, c->stream);
                    e_set[s].push_back(d);
                }
            }
        }
    }
    for (uint64_t ci = 0; ci < n_chunks && !rc && !slice_gw;) {
        int g = (int) std::min<uint64_t>((uint64_t) group_cap, n_chunks - ci);
        const int gs = g <= 2 ? 2 : g <= 4 ? 4 : 8;
        if (g > 1 && ensure_slots(c, g, gs)) {   // not enough memory for the group
            (void) hipGetLastError();
            if (g > 4) {                          // eight slots do not fit: groups of four
                group_cap = 4;
                continue;
            }
            g = 1;                                // one chunk at a time
            group_cap = 1;
        }
        hipEvent_t a = nullptr, b = nullptr;
        if (timed) {
            if (new_event(&a) || new_event(&b)) { rc = 1; break; }
            (void) hipEventRecord(a, c->stream);
        }
        // two lanes: when every chunk of the group takes the bucketed construction (which writes all of its filter
        // slot itself), odd chunks are built on the second stream with the second workspace, beside the even ones
        // (round 6: groups of exactly two chunks only — configs[1]: 12.57 against 12.79 ms per step; a 50 M-read set's seven chunks build
        // in 58.7-60.9 ms on one lane and in 58.9-60.7 ms on two, and the second lane's workspace is 14 GiB more to ask the driver for)
        bool lanes = g == 2 && c->index_lanes > 1 && !c->kclock.on;   // per-kernel times are additive on one stream only
        for (int i = 0; i < g && lanes; ++i) {
            const Chunk &ch = plan.chunks[ci + i];
            lanes = ch.n_reads && would_partition(c, index_rs, ch.kmers);
        }
        if (lanes) {   // the second stream starts behind everything issued so far (the previous group's searches read the slots)
            if (hipEventRecord(c->ev_fork, c->stream) != hipSuccess || hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0) != hipSuccess) {
                rc = fail("stream fork failed");
                break;
            }
        }
        for (int i = 0; i < g && !rc; ++i) {
            const Chunk &ch = plan.chunks[ci + i];
            c->cur_slot = i;
            hipEvent_t z0 = nullptr, z1 = nullptr;
            if (timed) {
                if (new_event(&z0) || new_event(&z1)) { rc = 1; break; }
                (void) hipEventRecord(z0, c->stream);
            }
            // new BloomFilter per chunk: zero it, unless the bucketed build is going to write every tile anyway
            const bool self_zeroing = ch.n_reads && would_partition(c, index_rs, ch.kmers);
            if (!self_zeroing && commet_filter_reset(c)) { rc = 1; break; }
            if (timed) {
                (void) hipEventRecord(z1, c->stream);
                e_zero0.push_back(z0);
                e_zero1.push_back(z1);
            }
            if (ch.n_reads) {
                if (launch_index(c, index_rs, ch.first, ch.last - ch.first + 1, plan.dense ? nullptr : index_rs->d_sel, nullptr, ch.kmers, true, !self_zeroing,
                                 lanes ? (i & 1) : 0, d_ids, d_ids ? chunk_pos[ci + i] : 0, ch.n_reads)) { rc = 1; break; }
                ++n_index_launches;
            }
        }
        if (lanes && !rc) {
            if (hipEventRecord(c->ev_join, c->aux_stream) != hipSuccess || hipStreamWaitEvent(c->stream, c->ev_join, 0) != hipSuccess)
                rc = fail("stream join failed");
        }
        if (rc) break;
        if (g > 1 && launch_interleave(c, g, gs)) { rc = 1; break; }
        if (timed) {
            (void) hipEventRecord(b, c->stream);
            e_idx0.push_back(a);
            e_idx1.push_back(b);
        }
        for (int s = 0; s < n_search && !rc; ++s) {
            const commet_readset *rs = search_rs[s];
            unsigned long long *cnt = d_cnt + 2 * (ci * n_search + s);
            // the tiled search (tile_search.hpp) of one pass: 0 = launched, 1 = not for this set / group, 2 = error.  The set's
            // query list is made or found, and its kernels queued, under ql_mu: no other thread gives the list back in between
            auto try_tiled = [&](int tg, int slot0, unsigned long long *tcnt) -> int {
                std::lock_guard<std::mutex> qlk(c->ql_mu);
                if (!tiled_ok(c, rs, tg) || build_query_list(c, rs) != 0 || ensure_query_results(c, rs) != 0) return 1;
                return launch_search_tiled(c, rs, tg, slot0, all_visited[s] ? nullptr : rs->d_sel, rs->d_tags, tcnt, (uint32_t) (2 * n_search)) ? 2 : 0;
            };
            // a pass over few of the set's reads (the host plan visits less than half of them): their list, not the set (kernels.hpp,
            // ActiveList); the tiled search probes EVERY record of the set's query list, so such a pass takes the gather kernels
            const uint64_t *sel_s = all_visited[s] ? nullptr : rs->d_sel;
            const bool sparse = rs->n_reads && sparse_pass(c, rs, sel_s, visited[s]);
            ActiveList al{nullptr, nullptr};
            auto list_for_pass = [&]() -> bool {            // (re-made per pass: the tags of the pass before have shrunk it)
                al = ActiveList{nullptr, nullptr};
                return sparse && build_active_list(c, rs, sel_s, rs->d_tags, visited[s], &al) == 0;
            };
            const int tiled2 = (g == 2 && !sparse) ? try_tiled(2, 0, cnt) : 1;   // large set, two chunk filters: lane-a gathers served from L2, slice by slice
            if (tiled2 == 2) { rc = 1; break; }
            if (tiled2 == 0) {
                if (rs->n_reads) ++n_search_launches;
            } else if (g > 1 && (gs == 8 || group_searchable(c, rs, g))) {
                (void) list_for_pass();
                // a ragged set visited whole, the job's first pass over it (no tags yet): its reads in order of their window counts
                const uint64_t n_listed = ordered_pass(c, rs, sel_s, ci == 0, &al) ? rs->n_reads : visited[s];
                if (launch_search_group(c, rs, g, gs, sel_s, rs->d_tags, cnt, (uint32_t) (2 * n_search), d_probes, al, n_listed)) { rc = 1; break; }
                if (rs->n_reads) ++n_search_launches;
            } else {
                for (int i = 0; i < g && !rc; ++i) {
                    c->cur_slot = i;
                    const int tiled1 = sparse ? 1 : try_tiled(1, i, cnt + 2 * (uint64_t) i * n_search);   // the same, one filter at a time
                    if (tiled1 == 2) rc = 1;
                    else if (tiled1 == 1) {
                        (void) list_for_pass();
                        const uint64_t n_listed = ordered_pass(c, rs, sel_s, ci == 0 && i == 0, &al) ? rs->n_reads : visited[s];
                        if (launch_search(c, rs, sel_s, rs->d_tags, nullptr, cnt + 2 * (uint64_t) i * n_search, d_probes, al, n_listed)) rc = 1;
                    }
                    if (rs->n_reads) ++n_search_launches;
                }
            }
            if (timed && !rc) {
                hipEvent_t d = nullptr;
                if (new_event(&d)) { rc = 1; break; }
                (void) hipEventRecord(d, c->stream);
                e_set[s].push_back(d);
            }
        }
        c->cur_slot = 0;
        ci += (uint64_t) g;
    }
    c->cur_slot = 0;
    lap(ph_launch);
    if (!rc)
        if (hipMemcpyAsync(h_cnt.data(), d_cnt, n_cnt * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
            rc = fail("counter copy failed");
    for (int s = 0; s < n_search && !rc; ++s) {
        const commet_readset *rs = search_rs[s];
        if (tags_out && tags_out[s])
            if (hipMemcpyAsync(tags_out[s], rs->d_tags, bitmap_bytes_host(rs->n_reads), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
                rc = fail("tag copy failed");
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = fail("stream synchronize failed: %s", hipGetErrorString(hipGetLastError()));
    c->kclock.collect();
    lap(ph_wait);
    if (!rc && d_ids && (c->h_counters[N_COUNTERS - 1] & 0xFFFFFFFFull) != (ids_expected & 0xFFFFFFFFull))
        rc = fail("internal error: the selection list holds %llu reads, the plan indexes %llu", (unsigned long long) (c->h_counters[N_COUNTERS - 1] & 0xFFFFFFFFull),
                  (unsigned long long) ids_expected);

    if (!rc) {
        uint64_t scans = 0;
        for (int s = 0; s < n_search; ++s) {
            uint64_t shared = 0, last_scanned = 0;
            for (uint64_t ci = 0; ci < n_chunks; ++ci) {
                const unsigned long long *p = &h_cnt[2 * (ci * n_search + s)];
                // an empty search set launches nothing: scanned = visited - found so far
                last_scanned = visited[s] - shared;
                scans += last_scanned;
                if (search_rs[s]->n_reads && !slice_gw && p[0] != last_scanned)   // (the sliced kernel counts found reads only)
                    rc = fail("internal error: device scanned %llu reads, host plan says %llu (chunk %llu, set %d)",
                              p[0], (unsigned long long) last_scanned, (unsigned long long) ci, s);
                shared += p[1];
            }
            if (stats) {
                stats[s].indexed = plan.indexed_reads;
                stats[s].searched = n_chunks ? last_scanned : 0;
                stats[s].shared = shared;
                stats[s].search_ms = 0;
            }
        }
        double idx_ms = 0, srch_ms = 0, zero_ms = 0;
        if (timed && !rc) {
            for (size_t i = 0; i < e_zero0.size(); ++i) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, e_zero0[i], e_zero1[i]) == hipSuccess) zero_ms += ms;
            }
            for (size_t i = 0; i < e_idx0.size(); ++i) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, e_idx0[i], e_idx1[i]) == hipSuccess) idx_ms += ms;
                for (int s = 0; s < n_search; ++s) {
                    if (i >= e_set[s].size()) continue;
                    hipEvent_t prev = s == 0 ? e_idx1[i] : e_set[s - 1][i];
                    if (hipEventElapsedTime(&ms, prev, e_set[s][i]) == hipSuccess) {
                        srch_ms += ms;
                        if (stats) stats[s].search_ms += ms;
                    }
                }
            }
        }
        if (info) {
            info->n_chunks = n_chunks;
            info->kmers_indexed = plan.kmers;
            info->reads_scanned = scans;
            info->reads_indexed = plan.indexed_reads;
            info->index_launches = n_index_launches;
            info->search_launches = n_search_launches;
            info->probes = h_cnt[n_cnt - 1];
            info->zero_ms = zero_ms;
            info->index_ms = idx_ms;
            info->index_kernel_ms = idx_ms - zero_ms;
            info->search_ms = srch_ms;
        }
    }
    for (hipEvent_t e : evs) (void) hipEventDestroy(e);
    if (job_verbose) {
        double ph_tail = 0;
        lap(ph_tail);
        fprintf(stderr, "[job] plan %.2f ms, bitmap upload %.2f ms, launches %.2f ms, wait + download %.2f ms, stats + cleanup %.2f ms\n",
                ph_plan, ph_upload, ph_launch, ph_wait, ph_tail);
    }
    if (info && !rc)
        info->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    return rc;
}

}  // extern "C"
