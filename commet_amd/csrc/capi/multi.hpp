// capi/multi.hpp — commet_index_many_and_search: several index_and_search jobs that search the SAME read set, their chunk filters side by side in one pass
// (a part of the one translation unit capi.hip: included there, in order, after job.hpp)
//
// Commet.py runs, for a reference set S_ref and every other set S_i, J2 = "S_ref in (S_i restricted to J1's result)" and
// J3 = "S_i in (S_ref restricted to J2's result)" (Commet.py:220, 233): index sets of a few chunk filters each, and the search set
// of all J2 jobs of a reference set is that reference set (of all J3 jobs of a target, that target).  A job on its own costs ~55
// L2-missing requests per searched read of which 37 are the lane-a gathers of the read's first-hit windows — addresses that depend on
// the read alone.  With the filters of up to eight chunks of several such jobs in the slots of one pass (their A planes interleaved:
// one 32-byte gather serves them all) the set is scanned once per pass instead of once per job; behind the gather
// search_group8_kernel runs job by job (kernels.hpp, job_mask).  Every job's result is what commet_index_and_search gives for it
// alone — tested against exactly that, and against the CPU checker through the N x N driver.
//
// The fast path takes what the N x N driver's jobs are: index sets whose chunks (at most eight per job) take the bucketed
// construction, a search set that is visited whole and qualifies for the register-mask kernel.  Anything else — and n_jobs = 1 — is
// run job by job through commet_index_and_search itself.
#pragma once

extern "C" {

int commet_index_many_and_search(commet_ctx *c, int n_jobs, const commet_readset *const *index_rs, const uint8_t *const *index_select,
                                 const commet_readset *search_rs, const uint8_t *search_select, uint8_t *const *tags_out,
                                 commet_pair_stats *stats, commet_job_info *info)
{
    const auto wall0 = std::chrono::steady_clock::now();
    if (n_jobs < 0) return fail("n_jobs must be >= 0");
    commet_job_info sum = commet_job_info();
    auto lap_t = wall0;                            // host-side phase times of the call (COMMET_JOB_VERBOSE: one line per call on stderr)
    double ph_plan = 0, ph_launch = 0, ph_wait = 0;
    auto lap = [&](double &acc) {
        const auto now = std::chrono::steady_clock::now();
        acc += std::chrono::duration<double, std::milli>(now - lap_t).count();
        lap_t = now;
    };
    auto one_by_one = [&]() -> int {
        for (int j = 0; j < n_jobs; ++j) {
            commet_job_info ji = commet_job_info();
            const uint8_t *ss = search_select;
            uint8_t *to = tags_out ? tags_out[j] : nullptr;
            if (commet_index_and_search(c, index_rs[j], index_select ? index_select[j] : nullptr, 1, &search_rs, search_select ? &ss : nullptr,
                                        tags_out ? &to : nullptr, stats ? &stats[j] : nullptr, &ji))
                return 1;
            sum.n_chunks += ji.n_chunks, sum.kmers_indexed += ji.kmers_indexed, sum.reads_scanned += ji.reads_scanned;
            sum.reads_indexed += ji.reads_indexed, sum.index_launches += ji.index_launches, sum.search_launches += ji.search_launches;
            sum.probes += ji.probes, sum.zero_ms += ji.zero_ms, sum.index_ms += ji.index_ms, sum.index_kernel_ms += ji.index_kernel_ms;
            sum.search_ms += ji.search_ms;
        }
        sum.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        if (info) *info = sum;
        return 0;
    };
    if (!search_rs->finalized) return fail("search read set not finalized");
    if (search_rs->ctx != c) return fail("search read set belongs to another context");
    for (int j = 0; j < n_jobs; ++j) {
        if (!index_rs[j]->finalized) return fail("index read set %d not finalized", j);
        if (index_rs[j]->ctx != c) return fail("index read set %d belongs to another context", j);
        if (index_rs[j] == search_rs) return fail("a set cannot be searched against itself in one call");
    }
    HIP_OK(hipSetDevice(c->device));
    // the sets of this call keep their cached query lists, and are not exported, while it runs (as in commet_index_and_search; the jobs
    // that run one by one mark their own sets again: the flag is a flag, the outer scope clears it last)
    struct InJobs {
        commet_ctx *c;
        const commet_readset *const *irs;
        const commet_readset *srs;
        int n;
        void mark(bool v) const
        {
            std::lock_guard<std::mutex> lk(c->ql_mu);
            srs->in_job = v;
            for (int i = 0; i < n; ++i) irs[i]->in_job = v;
        }
        InJobs(commet_ctx *c_, const commet_readset *const *i_, const commet_readset *s_, int n_) : c(c_), irs(i_), srs(s_), n(n_) { mark(true); }
        ~InJobs() { mark(false); }
    } in_jobs(c, index_rs, search_rs, n_jobs);
    // ---- does the fast path take the call? ---------------------------------------------------------------------------------------
    const uint8_t *ssel = search_select;
    if (ssel && all_ones(ssel, search_rs->n_reads)) ssel = nullptr;
    const uint64_t max_kmer = commet_max_kmer(c);
    bool fast = n_jobs >= 2 && c->k >= 2 && !c->count_probes && c->chunk_group >= 8 && c->multi_job != 1 && search_rs->n_reads > 0 &&
                slice_words(c, 8) == 0 && group8_ok(c, search_rs) && plan_fast_ok(search_rs->files, ssel, search_rs->empty_reads, 1) &&
                (search_rs->n_reads + 255) / 256 < (1ull << 24);
    struct Job {
        IndexPlan plan;
        const uint8_t *sel = nullptr;
        std::vector<uint64_t> chunk_pos;        // first position of every chunk in the job's list of selected reads
    };
    std::vector<Job> jobs(fast ? (size_t) n_jobs : 0);
    for (int j = 0; j < n_jobs && fast; ++j) {
        const commet_readset *rs = index_rs[j];
        Job &job = jobs[(size_t) j];
        job.sel = index_select ? index_select[j] : nullptr;
        if (job.sel && all_ones(job.sel, rs->n_reads)) job.sel = nullptr;
        if (!rs->n_reads || c->part_no_uni || !plan_blocks_ok(rs->files, job.sel, rs->empty_reads, max_kmer)) {
            fast = false;
            break;
        }
        // the plan from per-block k-mer sums made on the device (as commet_index_and_search does)
        const uint64_t nblk = (rs->n_reads + PLAN_BLOCK_READS - 1) / PLAN_BLOCK_READS;
        if (c->plansum_cap < nblk) {
            HIP_OK(hipStreamSynchronize(c->stream));
            (void) dm_free(c->d_plansum);
            c->d_plansum = nullptr, c->plansum_cap = 0;
            HIP_OK(dev_alloc(c, (void **) &c->d_plansum, nblk * sizeof(unsigned long long), true));
            c->plansum_cap = nblk;
        }
        if (job.sel && upload_bits(c, rs->d_sel, job.sel, rs->n_reads)) return 1;
        {
            KScope ks(c, "block_kmer_sums_kernel", c->stream);
            COMMET_LAUNCH(block_kmer_sums_kernel, dim3((unsigned) nblk), dim3(256), 0, c->stream, rs->d_kcnt, job.sel ? rs->d_sel : nullptr,
                          rs->n_reads, c->d_plansum);
        }
        HIP_OK(hipGetLastError());
        std::vector<uint64_t> blk_sums(nblk);
        HIP_OK(hipMemcpyAsync(blk_sums.data(), c->d_plansum, nblk * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
        std::vector<uint32_t> kblock(PLAN_BLOCK_READS);
        uint64_t kblock_no = ~0ull;
        bool kfetch_failed = false;
        auto kcnt_of = [&](uint64_t q) -> uint32_t {
            if (rs->have_host_counts) return rs->h_kcnt[q];
            const uint64_t blk = q / PLAN_BLOCK_READS;
            if (blk != kblock_no) {
                const uint64_t lo = blk * PLAN_BLOCK_READS, cnt = std::min<uint64_t>(PLAN_BLOCK_READS, rs->n_reads - lo);
                if (hipMemcpy(kblock.data(), rs->d_kcnt + lo, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) kfetch_failed = true;
                kblock_no = blk;
            }
            return kblock[q % PLAN_BLOCK_READS];
        };
        job.plan = plan_index_blocks(job.sel, kcnt_of, rs->n_reads, max_kmer, blk_sums.data(), PLAN_BLOCK_READS);
        if (kfetch_failed) return fail("k-mer count fetch failed: %s", hipGetErrorString(hipGetLastError()));
        if (job.plan.chunks.empty() || job.plan.chunks.size() > 8) fast = false;
        uint64_t at = 0;
        for (const Chunk &ch : job.plan.chunks) {
            if (!ch.n_reads || !would_partition(c, rs, ch.kmers)) fast = false;    // (the bucketed build writes every tile of its slot itself)
            job.chunk_pos.push_back(at), at += ch.n_reads;
        }
    }
    bool pairs = false;                            // jobs of ONE chunk each on a search set that takes the tiled search: two jobs per scan (below)
    if (fast) {
        // A search set that takes the tiled search (a query list within the cap: sets of up to ~15 M reads) loses little on its own — its
        // lane-a gathers come out of L2 — and jobs of one or two chunks each need no eight filter slots there (20 GiB more at k = 32, which a
        // fresh box hands out at 15-30 ms per GiB): such jobs stay out of the eight-slot passes (configs[2]'s leg: 1.5 s either way on a
        // used box, 2.0 s against 1.5 s on a fresh one).
        size_t most = 0;
        for (const Job &job : jobs) most = std::max(most, job.plan.chunks.size());
        std::lock_guard<std::mutex> qlk(c->ql_mu);
        if (most <= 2 && tiled_ok(c, search_rs, 2)) {
            fast = false;
            pairs = most == 1 && c->multi_job != 2;
        }
    }
    if (!fast && !pairs) return one_by_one();
    lap(ph_plan);
    if (pairs) {
        // ---- two single-chunk jobs per tiled scan (round 6) ---------------------------------------------------------------------------
        // The J2 / J3 jobs of a matrix of 10 M-read sets index a fifth of a set (one chunk filter) and search a whole one through the
        // tiled search: probe 2.2 ms + replay 3 ms per job.  The probe's gather of a query record's plane-A word serves two interleaved
        // filters as cheaply as one, so consecutive jobs go through the scan in twos: their filters in slots 0 and 1, one probe, one
        // replay that keeps the two jobs apart (tq_replay_kernel, job_tag_words).  An odd job out, and everything when the list cannot
        // be had, runs through commet_index_and_search.
        const uint64_t tag_words = bitmap_words(search_rs->n_reads);
        if (c->mtags_cap < 2 * tag_words) {
            HIP_OK(hipStreamSynchronize(c->stream));
            (void) dm_free(c->d_mtags);
            c->d_mtags = nullptr, c->mtags_cap = 0;
            if (dev_alloc(c, (void **) &c->d_mtags, 8 * tag_words * sizeof(uint64_t), true) != hipSuccess) {
                (void) hipGetLastError();
                return one_by_one();
            }
            c->mtags_cap = 8 * tag_words;
        }
        if (c->jobcnt_cap < 16) {
            HIP_OK(hipStreamSynchronize(c->stream));
            (void) dm_free(c->d_jobcnt);
            c->d_jobcnt = nullptr, c->jobcnt_cap = 0;
            HIP_OK(dev_alloc(c, (void **) &c->d_jobcnt, 64 * sizeof(unsigned long long), true));
            c->jobcnt_cap = 64;
        }
        auto alone = [&](int j) -> int {           // one job through commet_index_and_search, summed into `sum`
            commet_job_info ji = commet_job_info();
            const uint8_t *ss = search_select;
            uint8_t *to = tags_out ? tags_out[j] : nullptr;
            if (commet_index_and_search(c, index_rs[j], index_select ? index_select[j] : nullptr, 1, &search_rs, search_select ? &ss : nullptr,
                                        tags_out ? &to : nullptr, stats ? &stats[j] : nullptr, &ji))
                return 1;
            sum.n_chunks += ji.n_chunks, sum.kmers_indexed += ji.kmers_indexed, sum.reads_scanned += ji.reads_scanned;
            sum.reads_indexed += ji.reads_indexed, sum.index_launches += ji.index_launches, sum.search_launches += ji.search_launches;
            sum.zero_ms += ji.zero_ms, sum.index_ms += ji.index_ms, sum.index_kernel_ms += ji.index_kernel_ms, sum.search_ms += ji.search_ms;
            return 0;
        };
        struct TidyP {
            commet_ctx *c;
            std::vector<hipEvent_t> evs;
            ~TidyP()
            {
                c->cur_slot = 0;
                for (hipEvent_t e : evs) (void) hipEventDestroy(e);
            }
        } tidy{c, {}};
        int j0 = 0;
        for (; j0 + 1 < n_jobs; j0 += 2) {
            if (ensure_slots(c, 2, 2)) {
                (void) hipGetLastError();
                break;                             // no room for two slots: the rest one by one
            }
            hipEvent_t ea = nullptr, eb = nullptr, ec = nullptr;
            for (hipEvent_t *e : {&ea, &eb, &ec}) {
                HIP_OK(hipEventCreate(e));
                tidy.evs.push_back(*e);
            }
            (void) hipEventRecord(ea, c->stream);
            // the two chunks are built side by side on the context's two index lanes (as the two chunks of one job are): first, on the main
            // stream, what each build reads — the selection bitmap and, for sets of one read length, the list of the selected reads, the
            // second job's in a buffer of its own — then the fork
            const uint32_t *ids_of[2] = {nullptr, nullptr};
            const bool same_set = index_rs[j0] == index_rs[j0 + 1];    // (one set in both jobs: ONE selection bitmap on the device — the second job's goes up behind the first build)
            auto prepare = [&](int j) -> int {       // job j's selection on the device: its bitmap and, for sets of one read length, the list of the selected reads
                const commet_readset *rs = index_rs[j];
                Job &job = jobs[(size_t) j];
                if (job.plan.dense) return 0;
                if (upload_bits(c, rs->d_sel, job.plan.indexed_bits.data(), rs->n_reads)) return 1;
                if (rs->uniform_len == 0) return 0;
                uint32_t *&ids = j == j0 ? c->d_ids : c->d_ids2, *&blk = j == j0 ? c->d_idblk : c->d_idblk2;
                uint64_t &ids_cap = j == j0 ? c->ids_cap : c->ids2_cap, &blk_cap = j == j0 ? c->idblk_cap : c->idblk2_cap;
                const uint64_t n_words = bitmap_words(rs->n_reads), nb = (n_words + IDS_BLOCK_WORDS - 1) / IDS_BLOCK_WORDS;
                if (ids_cap < job.plan.indexed_reads || blk_cap < nb + 1) {
                    HIP_OK(hipStreamSynchronize(c->stream));
                    HIP_OK(hipStreamSynchronize(c->aux_stream));
                    (void) dm_free(ids), (void) dm_free(blk);
                    ids = blk = nullptr, ids_cap = blk_cap = 0;
                    const uint64_t cap = std::max<uint64_t>(job.plan.indexed_reads, rs->n_reads / 2);
                    HIP_OK(dev_alloc(c, (void **) &ids, cap * sizeof(uint32_t), true));
                    HIP_OK(dev_alloc(c, (void **) &blk, (nb + 1) * sizeof(uint32_t), true));
                    ids_cap = cap, blk_cap = nb + 1;
                }
                KScope ks(c, "sel_ids_kernels", c->stream);
                COMMET_LAUNCH(sel_count_kernel, dim3((unsigned) nb), dim3(64), 0, c->stream, rs->d_sel, n_words, blk, (const uint64_t *) nullptr);
                COMMET_LAUNCH(sel_scan_kernel, dim3(1), dim3(1024), 0, c->stream, blk, (uint32_t) nb);
                COMMET_LAUNCH(sel_ids_kernel, dim3((unsigned) nb), dim3(64), 0, c->stream, rs->d_sel, n_words, blk, ids, (const uint64_t *) nullptr);
                HIP_OK(hipGetLastError());
                ids_of[j - j0] = ids;
                return 0;
            };
            if (prepare(j0) || (!same_set && prepare(j0 + 1))) return 1;
            const bool lanes = c->index_lanes > 1 && !c->kclock.on && !same_set;
            if (lanes) {
                HIP_OK(hipEventRecord(c->ev_fork, c->stream));
                HIP_OK(hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
            }
            for (int j = j0; j < j0 + 2; ++j) {
                const commet_readset *rs = index_rs[j];
                Job &job = jobs[(size_t) j];
                const Chunk &ch = job.plan.chunks[0];
                if (same_set && j == j0 + 1 && prepare(j)) return 1;
                c->cur_slot = j - j0;
                if (launch_index(c, rs, ch.first, ch.last - ch.first + 1, job.plan.dense ? nullptr : rs->d_sel, nullptr, ch.kmers, true, false, lanes ? j - j0 : 0,
                                 ids_of[j - j0], 0, ch.n_reads))
                    return 1;
                ++sum.index_launches;
            }
            if (lanes) {
                HIP_OK(hipEventRecord(c->ev_join, c->aux_stream));
                HIP_OK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
            }
            c->cur_slot = 0;
            if (launch_interleave(c, 2, 2)) return 1;
            (void) hipEventRecord(eb, c->stream);
            HIP_OK(hipMemsetAsync(c->d_mtags, 0, 2 * tag_words * sizeof(uint64_t), c->stream));
            HIP_OK(hipMemsetAsync(c->d_jobcnt, 0, 16 * sizeof(unsigned long long), c->stream));
            bool scanned = false;
            {
                std::lock_guard<std::mutex> qlk(c->ql_mu);
                if (tiled_ok(c, search_rs, 2) && build_query_list(c, search_rs) == 0 && ensure_query_results(c, search_rs) == 0) {
                    if (launch_search_tiled(c, search_rs, 2, 0, nullptr, c->d_mtags, c->d_jobcnt, 2, tag_words)) return 1;
                    scanned = true;
                }
            }
            if (!scanned) {                        // the list could not be had after all: these two and the rest through the jobs' own path
                HIP_OK(hipStreamSynchronize(c->stream));
                break;
            }
            ++sum.search_launches;
            (void) hipEventRecord(ec, c->stream);
            lap(ph_launch);
            unsigned long long h_cnt[4];
            HIP_OK(hipMemcpyAsync(h_cnt, c->d_jobcnt, sizeof h_cnt, hipMemcpyDeviceToHost, c->stream));
            for (int j = j0; j < j0 + 2; ++j)
                if (tags_out && tags_out[j])
                    HIP_OK(hipMemcpyAsync(tags_out[j], c->d_mtags + (uint64_t) (j - j0) * tag_words, bitmap_bytes_host(search_rs->n_reads), hipMemcpyDeviceToHost, c->stream));
            HIP_OK(hipStreamSynchronize(c->stream));
            c->kclock.collect();
            lap(ph_wait);
            float ms_i = 0, ms_s = 0;
            (void) hipEventElapsedTime(&ms_i, ea, eb);
            (void) hipEventElapsedTime(&ms_s, eb, ec);
            sum.index_ms += ms_i, sum.index_kernel_ms += ms_i, sum.search_ms += ms_s;
            for (int j = j0; j < j0 + 2; ++j) {
                const Job &job = jobs[(size_t) j];
                const unsigned long long sc = h_cnt[2 * (j - j0)], fd = h_cnt[2 * (j - j0) + 1];
                if (sc != search_rs->n_reads)
                    return fail("internal error: device scanned %llu reads, host plan says %llu (job %d)", sc, (unsigned long long) search_rs->n_reads, j);
                if (stats) {
                    stats[j].indexed = job.plan.indexed_reads;
                    stats[j].searched = search_rs->n_reads;
                    stats[j].shared = fd;
                    stats[j].search_ms = ms_s / 2.0;
                }
                sum.n_chunks += 1, sum.kmers_indexed += job.plan.kmers, sum.reads_indexed += job.plan.indexed_reads, sum.reads_scanned += search_rs->n_reads;
            }
        }
        for (int j = j0; j < n_jobs; ++j)
            if (alone(j)) return 1;
        sum.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        if (c->job_verbose)
            fprintf(stderr, "[jobs x%d, two per tiled scan] plans %.2f ms, launches %.2f ms, wait + download %.2f ms; device: index %.2f ms, search %.2f ms\n", n_jobs,
                    ph_plan, ph_launch, ph_wait, sum.index_ms, sum.search_ms);
        if (info) *info = sum;
        return 0;
    }
    // no room for what a shared pass needs (eight filter slots + their interleaved A planes are 20 GiB at k = 32): the jobs one after the
    // other, as the header promises — commet_index_and_search itself degrades to groups of four, then one.  Nothing a caller depends on
    // has been written by then that the jobs do not write again.
    auto no_room = [&]() -> int {
        (void) hipGetLastError();
        c->cur_slot = 0;
        sum = commet_job_info();
        return one_by_one();
    };

    // ---- passes: consecutive jobs while their chunks fit the eight slots ---------------------------------------------------------------
    const uint64_t tag_words = bitmap_words(search_rs->n_reads);
    if (c->mtags_cap < 8 * tag_words) {
        HIP_OK(hipStreamSynchronize(c->stream));
        (void) dm_free(c->d_mtags);
        c->d_mtags = nullptr, c->mtags_cap = 0;
        if (dev_alloc(c, (void **) &c->d_mtags, 8 * tag_words * sizeof(uint64_t), true) != hipSuccess) return no_room();
        c->mtags_cap = 8 * tag_words;
    }
    if (c->jobcnt_cap < 16) {
        HIP_OK(hipStreamSynchronize(c->stream));
        (void) dm_free(c->d_jobcnt);
        c->d_jobcnt = nullptr, c->jobcnt_cap = 0;
        HIP_OK(dev_alloc(c, (void **) &c->d_jobcnt, 64 * sizeof(unsigned long long), true));
        c->jobcnt_cap = 64;
    }
    struct Tidy {                                  // on every way out: the events destroyed, the context's slot cursor back at 0
        commet_ctx *c;
        std::vector<hipEvent_t> evs;
        ~Tidy()
        {
            c->cur_slot = 0;
            for (hipEvent_t e : evs) (void) hipEventDestroy(e);
        }
    } tidy{c, {}};
    auto new_event = [&](hipEvent_t *e) -> int {
        HIP_OK(hipEventCreate(e));
        tidy.evs.push_back(*e);
        return 0;
    };
    int rc = 0;
    for (int j0 = 0; j0 < n_jobs && !rc;) {
        int j1 = j0, g = 0;
        while (j1 < n_jobs && g + (int) jobs[(size_t) j1].plan.chunks.size() <= 8) g += (int) jobs[(size_t) j1].plan.chunks.size(), ++j1;
        if (ensure_slots(c, g, 8)) return no_room();
        hipEvent_t ea = nullptr, eb = nullptr, ec = nullptr;
        if (new_event(&ea) || new_event(&eb) || new_event(&ec)) { rc = 1; break; }
        (void) hipEventRecord(ea, c->stream);
        uint32_t job_mask = 0;
        int slot = 0;
        for (int j = j0; j < j1 && !rc; ++j) {
            const commet_readset *rs = index_rs[j];
            Job &job = jobs[(size_t) j];
            job_mask |= 1u << slot;
            const uint32_t *d_ids = nullptr;
            if (!job.plan.dense) {
                // the job's selected reads as a list (index_part.hpp, sel_ids_kernel); the chunks of every job are built on the one
                // stream, one after the other, so the list buffer of the context serves job after job
                if (upload_bits(c, rs->d_sel, job.plan.indexed_bits.data(), rs->n_reads)) { rc = 1; break; }
            }
            if (!job.plan.dense && rs->uniform_len != 0) {      // (ragged sets: the bucketed build lists the chunk's items itself, from the bitmap)
                const uint64_t n_words = bitmap_words(rs->n_reads), nb = (n_words + IDS_BLOCK_WORDS - 1) / IDS_BLOCK_WORDS;
                if (c->ids_cap < job.plan.indexed_reads || c->idblk_cap < nb + 1) {
                    HIP_OK(hipStreamSynchronize(c->stream));
                    (void) dm_free(c->d_ids), (void) dm_free(c->d_idblk);
                    c->d_ids = c->d_idblk = nullptr, c->ids_cap = c->idblk_cap = 0;
                    const uint64_t cap = std::max<uint64_t>(job.plan.indexed_reads, rs->n_reads / 2);
                    if (dev_alloc(c, (void **) &c->d_ids, cap * sizeof(uint32_t), true) != hipSuccess ||
                        dev_alloc(c, (void **) &c->d_idblk, (nb + 1) * sizeof(uint32_t), true) != hipSuccess) {
                        (void) dm_free(c->d_ids), (void) dm_free(c->d_idblk);
                        c->d_ids = c->d_idblk = nullptr;
                        (void) hipStreamSynchronize(c->stream);      // (chunks of earlier jobs of the pass may be under way)
                        return no_room();
                    }
                    c->ids_cap = cap, c->idblk_cap = nb + 1;
                }
                KScope ks(c, "sel_ids_kernels", c->stream);
                COMMET_LAUNCH(sel_count_kernel, dim3((unsigned) nb), dim3(64), 0, c->stream, rs->d_sel, n_words, c->d_idblk, (const uint64_t *) nullptr);
                COMMET_LAUNCH(sel_scan_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_idblk, (uint32_t) nb);
                COMMET_LAUNCH(sel_ids_kernel, dim3((unsigned) nb), dim3(64), 0, c->stream, rs->d_sel, n_words, c->d_idblk, c->d_ids, (const uint64_t *) nullptr);
                if (hipGetLastError() != hipSuccess) { rc = fail("selection list launch failed"); break; }
                d_ids = c->d_ids;
            }
            for (size_t ci = 0; ci < job.plan.chunks.size() && !rc; ++ci, ++slot) {
                const Chunk &ch = job.plan.chunks[ci];
                c->cur_slot = slot;
                if (launch_index(c, rs, ch.first, ch.last - ch.first + 1, job.plan.dense ? nullptr : rs->d_sel, nullptr, ch.kmers, true, false, 0, d_ids,
                                 d_ids ? job.chunk_pos[ci] : 0, ch.n_reads))
                    rc = 1;
                ++sum.index_launches;
            }
        }
        c->cur_slot = 0;
        if (rc) break;
        if (launch_interleave(c, g, 8)) { rc = 1; break; }
        (void) hipEventRecord(eb, c->stream);
        if (hipMemsetAsync(c->d_mtags, 0, (size_t) (j1 - j0) * tag_words * sizeof(uint64_t), c->stream) != hipSuccess ||
            hipMemsetAsync(c->d_jobcnt, 0, 16 * sizeof(unsigned long long), c->stream) != hipSuccess) { rc = fail("memset failed"); break; }
        ActiveList al{nullptr, nullptr};               // (a ragged search set: its reads in order of their window counts — every job's tags start empty)
        const uint64_t n_listed = ordered_pass(c, search_rs, nullptr, true, &al) ? search_rs->n_reads : 0;
        if (launch_search_group(c, search_rs, g, 8, nullptr, c->d_mtags, c->d_jobcnt, 2, nullptr, al, n_listed, job_mask, tag_words)) { rc = 1; break; }
        ++sum.search_launches;
        (void) hipEventRecord(ec, c->stream);
        lap(ph_launch);
        unsigned long long h_cnt[16];
        if (hipMemcpyAsync(h_cnt, c->d_jobcnt, sizeof h_cnt, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { rc = fail("counter copy failed"); break; }
        for (int j = j0; j < j1 && !rc; ++j)
            if (tags_out && tags_out[j] &&
                hipMemcpyAsync(tags_out[j], c->d_mtags + (uint64_t) (j - j0) * tag_words, bitmap_bytes_host(search_rs->n_reads), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
                rc = fail("tag copy failed");
        if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = fail("stream synchronize failed: %s", hipGetErrorString(hipGetLastError()));
        c->kclock.collect();
        lap(ph_wait);
        if (rc) break;
        float ms_i = 0, ms_s = 0;
        (void) hipEventElapsedTime(&ms_i, ea, eb);
        (void) hipEventElapsedTime(&ms_s, eb, ec);
        sum.index_ms += ms_i, sum.index_kernel_ms += ms_i, sum.search_ms += ms_s;
        slot = 0;
        for (int j = j0; j < j1; ++j) {
            const Job &job = jobs[(size_t) j];
            uint64_t shared = 0, last_scanned = 0;
            for (size_t ci = 0; ci < job.plan.chunks.size(); ++ci, ++slot) {
                last_scanned = search_rs->n_reads - shared;          // (the set is visited whole)
                if (h_cnt[2 * slot] != last_scanned)
                    rc = fail("internal error: device scanned %llu reads, host plan says %llu (job %d, chunk %zu)", h_cnt[2 * slot],
                              (unsigned long long) last_scanned, j, ci);
                shared += h_cnt[2 * slot + 1];
                sum.reads_scanned += last_scanned;
            }
            if (stats) {
                stats[j].indexed = job.plan.indexed_reads;
                stats[j].searched = last_scanned;
                stats[j].shared = shared;
                stats[j].search_ms = ms_s / (double) (j1 - j0);      // (the pass is shared: an equal part each)
            }
            sum.n_chunks += job.plan.chunks.size(), sum.kmers_indexed += job.plan.kmers, sum.reads_indexed += job.plan.indexed_reads;
        }
        j0 = j1;
    }
    if (rc) return rc;
    sum.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    if (c->job_verbose)
        fprintf(stderr, "[jobs x%d] plans %.2f ms, launches %.2f ms, wait + download %.2f ms; device: index %.2f ms, search %.2f ms\n", n_jobs, ph_plan,
                ph_launch, ph_wait, sum.index_ms, sum.search_ms);
    if (info) *info = sum;
    return 0;
}

}  // extern "C"
