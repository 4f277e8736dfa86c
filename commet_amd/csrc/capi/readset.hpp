// capi/readset.hpp — read sets resident in HBM: creation, the pinned staging API, the multi-threaded host ingest (2-bit packing on the parser threads, planes uploaded to their final place), finalize, counts
// (a part of the one translation unit capi.hip: included there, in order, after the kernels and state.hpp)
#pragma once

extern "C" {

/* ---- read sets ------------------------------------------------------------ */

commet_readset *commet_readset_create(commet_ctx *c, uint64_t max_reads, uint64_t max_bases)
{
    if (!c) {
        fail("null context");
        return nullptr;
    }
    HIP_OK_NULL(hipSetDevice(c->device));
    commet_readset *rs = new commet_readset;
    static std::atomic<uint64_t> next_uid{1};
    rs->uid = next_uid.fetch_add(1);
    rs->ctx = c;
    rs->max_reads = max_reads;
    rs->max_bases = max_bases;
    rs->stage_bases = max_bases < STAGE_BASES ? (max_bases ? max_bases : 1) : STAGE_BASES;
    rs->stage_reads = max_reads < STAGE_READS ? (max_reads ? max_reads : 1) : STAGE_READS;
    const uint64_t triples = (max_bases >> 5) + max_reads + 1;
    const uint64_t bw = bitmap_words(max_reads);
    // (a set may be made by a second host thread while a job runs: that thread never takes a list of the running job)
    hipError_t e = dev_alloc(c, (void **) &rs->d_planes, triples * 3 * sizeof(uint32_t), false);
    if (e == hipSuccess) e = dev_alloc(c, (void **) &rs->d_goff, (max_reads + 1) * sizeof(uint64_t), false);
    if (e == hipSuccess) e = dev_alloc(c, (void **) &rs->d_kcnt, (max_reads + 1) * sizeof(uint32_t), false);
    if (e == hipSuccess) e = dev_alloc(c, (void **) &rs->d_lenmm, 8 * sizeof(uint32_t), false)   /* [0..2] shortest / longest read, largest k-mer count; [4..5] one 64-bit sum (finalize) */;
    if (e == hipSuccess) e = dev_alloc(c, (void **) &rs->d_sel, bw * 8, false);
    if (e == hipSuccess) e = dev_alloc(c, (void **) &rs->d_tags, bw * 8, false);
    if (e == hipSuccess) e = dev_alloc(c, (void **) &rs->d_found, bw * 8, false);
    // the gap triples between reads and the closing triple are never written by the host packer when the neighbours
    // come from different staging buffers; no kernel reads them, but a packed image (commet_readset_save) carries them
    if (e == hipSuccess) e = hipMemsetAsync(rs->d_planes, 0, triples * 3 * sizeof(uint32_t), c->load_stream);
    if (e == hipSuccess) {
        const uint32_t mm[3] = {0xFFFFFFFFu, 0u, 0u};
        e = hipMemcpyAsync(rs->d_lenmm, mm, sizeof mm, hipMemcpyHostToDevice, c->load_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->load_stream);
    }
    if (e != hipSuccess) {
        fail("read set allocation failed (%llu reads, %llu bases): %s", (unsigned long long) max_reads,
             (unsigned long long) max_bases, hipGetErrorString(e));
        commet_readset_destroy(rs);
        return nullptr;
    }
    {
        std::lock_guard<std::mutex> lk(c->ql_mu);
        c->sets.push_back(rs);
    }
    return rs;
}

void commet_readset_destroy(commet_readset *rs)
{
    if (!rs) return;
    (void) hipSetDevice(rs->ctx->device);
    (void) hipStreamSynchronize(rs->ctx->load_stream);
    (void) hipStreamSynchronize(rs->ctx->stream);   // (a job that still reads the set)
    {
        commet_ctx *c = rs->ctx;
        std::lock_guard<std::mutex> lk(c->ql_mu);
        c->ql_bytes -= std::min(c->ql_bytes, rs->ql.bytes);
        c->sets.erase(std::remove(c->sets.begin(), c->sets.end(), rs), c->sets.end());
    }
    (void) dm_free(rs->d_planes);
    (void) dm_free(rs->d_goff);
    (void) dm_free(rs->d_kcnt);
    (void) dm_free(rs->d_lenmm);
    (void) dm_free(rs->d_sel);
    (void) dm_free(rs->d_tags);
    (void) dm_free(rs->d_found);
    rs->ql.release();
    (void) dm_free(rs->d_len_order);
    for (int i = 0; i < 2; ++i) {
        if (rs->st[i].h_bases) (void) hipHostFree(rs->st[i].h_bases);
        if (rs->st[i].h_offs) (void) hipHostFree(rs->st[i].h_offs);
        (void) dm_free(rs->st[i].d_bases);
        (void) dm_free(rs->st[i].d_offs);
        if (rs->st[i].done) (void) hipEventDestroy(rs->st[i].done);
    }
    delete rs;
}

int commet_readset_begin_file(commet_readset *rs)
{
    if (rs->finalized) return fail("read set already finalized");
    if (rs->acquired) return fail("begin_file with an uncommitted staging buffer");
    rs->files.push_back(FileSpan{rs->n_reads, 0});
    return 0;
}

int commet_readset_stage_acquire(commet_readset *rs, uint8_t **bases, uint64_t *bases_cap, uint64_t **offsets,
                                 uint64_t *reads_cap)
{
    if (rs->finalized) return fail("read set already finalized");
    if (rs->files.empty()) return fail("commet_readset_begin_file must be called first");
    if (rs->acquired) return fail("staging buffer already acquired");
    HIP_OK(hipSetDevice(rs->ctx->device));
    commet_readset::Stage &s = rs->st[rs->cur];
    if (!s.h_bases) {   // staging buffers are created on first use (commet_readset_from_fasta has its own)
        HIP_OK(hipHostMalloc((void **) &s.h_bases, rs->stage_bases));
        HIP_OK(hipHostMalloc((void **) &s.h_offs, (rs->stage_reads + 1) * sizeof(uint64_t)));
        HIP_OK(dm_malloc((void **) &s.d_bases, rs->stage_bases));
        HIP_OK(dm_malloc((void **) &s.d_offs, (rs->stage_reads + 1) * sizeof(uint64_t)));
        HIP_OK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    if (s.inflight) {
        HIP_OK(hipEventSynchronize(s.done));
        s.inflight = false;
    }
    *bases = s.h_bases;
    *bases_cap = rs->stage_bases;
    *offsets = s.h_offs;
    *reads_cap = rs->stage_reads;
    rs->acquired = true;
    return 0;
}

int commet_readset_stage_commit(commet_readset *rs, uint64_t n)
{
    if (!rs->acquired) return fail("commit without acquire");
    rs->acquired = false;
    if (n == 0) return 0;
    commet_readset::Stage &s = rs->st[rs->cur];
    if (n > rs->stage_reads) return fail("too many reads in one staging batch");
    if (s.h_offs[0] != 0) return fail("offsets[0] must be 0");
    const uint64_t nbases = s.h_offs[n];
    if (nbases > rs->stage_bases) return fail("staging batch overflows its base buffer");
    if (rs->n_reads + n > rs->max_reads || rs->n_bases + nbases > rs->max_bases)
        return fail("read set capacity exceeded (%llu reads / %llu bases reserved)", (unsigned long long) rs->max_reads,
                    (unsigned long long) rs->max_bases);
    for (uint64_t i = 0; i < n; ++i) {
        if (s.h_offs[i + 1] < s.h_offs[i]) return fail("offsets must be non-decreasing");
        if (s.h_offs[i + 1] - s.h_offs[i] > 0x7FFFFFFFull) return fail("read longer than 2^31-1 bases");
        if (s.h_offs[i + 1] == s.h_offs[i]) rs->empty_reads.push_back(rs->n_reads + i);
    }
    commet_ctx *c = rs->ctx;
    HIP_OK(hipSetDevice(c->device));
    if (nbases) HIP_OK(hipMemcpyAsync(s.d_bases, s.h_bases, nbases, hipMemcpyHostToDevice, c->load_stream));
    HIP_OK(hipMemcpyAsync(s.d_offs, s.h_offs, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->load_stream));
    const unsigned grid = (unsigned) ((n + 1 + 255) / 256);
    COMMET_LAUNCH(pack_reads_kernel, dim3(grid), dim3(256), 0, c->load_stream, s.d_bases, s.d_offs, n, rs->n_reads,
                       rs->n_bases, rs->d_planes, rs->d_goff, rs->d_kcnt, rs->d_lenmm, c->k);
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventRecord(s.done, c->load_stream));
    s.inflight = true;
    rs->n_reads += n;
    rs->n_bases += nbases;
    rs->files.back().count += n;
    rs->cur ^= 1;
    return 0;
}

}  // extern "C"

namespace {

// ---- host ingest: records are 2-bit packed by the ingest threads (host/ingest_pack.hpp) and uploaded as planes ----
#ifndef INGEST_STAGE_KIB
#define INGEST_STAGE_KIB 3072
#endif
#ifndef INGEST_STAGE_READS_LOG2
#define INGEST_STAGE_READS_LOG2 17
#endif
constexpr uint64_t INGEST_STAGE_BYTES = (uint64_t) INGEST_STAGE_KIB << 10;       // one pinned staging buffer of planes (12 bytes per triple): 8 M bases;
                                                          // small, because pinning memory costs ~0.2 ms per MiB on first use
constexpr uint64_t INGEST_STAGE_READS = 1ull << INGEST_STAGE_READS_LOG2;       // base offsets per staging buffer

// the upload side of host/ingest_pack.hpp: two pinned staging buffers per worker out of the context's pool; a flush
// queues hipMemcpyAsync of the planes (and the reads' base offsets) straight to their final place in the read set
struct HipPackSink {
    commet_readset *rs = nullptr;
    std::vector<int> cur;                    // which of its two buffers a worker fills next
    std::vector<char> inflight;              // per pool buffer

    bool prepare(commet_readset *set, int workers)
    {
        rs = set;
        commet_ctx *c = rs->ctx;
        if (hipSetDevice(c->device) != hipSuccess) return false;
        // the pool's entries exist up front (workers never resize it); their pinned memory is made by the worker that
        // first needs it, in acquire(): pinning costs ~0.2 ms per MiB, and paid here, on one thread before any packing,
        // it was 57 ms of the first set's 107 ms
        if (c->ingest_pool.size() < (size_t) workers * 2) c->ingest_pool.resize((size_t) workers * 2);
        cur.assign(workers, 0);
        inflight.assign((size_t) workers * 2, 0);
        return true;
    }
    bool acquire(int worker, commet_host::PackStage &st)
    {
        const size_t bi = (size_t) worker * 2 + cur[worker];
        commet_ctx::IngestBuf &b = rs->ctx->ingest_pool[bi];
        if (!b.done) {   // hipHostMalloc is slow: buffers stay with the context
            if (hipSetDevice(rs->ctx->device) != hipSuccess) return false;
            const bool ok = hipHostMalloc((void **) &b.h_planes, INGEST_STAGE_BYTES) == hipSuccess &&
                            hipHostMalloc((void **) &b.h_goff, INGEST_STAGE_READS * sizeof(uint64_t)) == hipSuccess &&
                            hipEventCreateWithFlags(&b.done, hipEventDisableTiming) == hipSuccess;
            if (!ok) {   // a half-made entry must not stay
                if (b.h_planes) (void) hipHostFree(b.h_planes);
                if (b.h_goff) (void) hipHostFree(b.h_goff);
                if (b.done) (void) hipEventDestroy(b.done);
                b = commet_ctx::IngestBuf();
                (void) hipGetLastError();
                return false;
            }
        }
        if (inflight[bi]) {
            if (hipEventSynchronize(b.done) != hipSuccess) return false;
            inflight[bi] = 0;
        }
        st.planes = b.h_planes;
        st.goff = b.h_goff;
        st.cap_triples = INGEST_STAGE_BYTES / 12;
        st.cap_reads = INGEST_STAGE_READS;
        return true;
    }
    bool flush(int worker, const commet_host::PackStage &st, uint64_t triple0, uint64_t n_triples, uint64_t read0, uint64_t n_reads)
    {
        commet_ctx *c = rs->ctx;
        const size_t bi = (size_t) worker * 2 + cur[worker];
        if (triple0 + n_triples > (rs->max_bases >> 5) + rs->max_reads + 1 || read0 + n_reads > rs->max_reads) return false;
        if (hipSetDevice(c->device) != hipSuccess) return false;
        if (n_triples && hipMemcpyAsync(rs->d_planes + 3 * triple0, st.planes, n_triples * 12, hipMemcpyHostToDevice, c->load_stream) != hipSuccess) return false;
        if (n_reads && hipMemcpyAsync(rs->d_goff + read0, st.goff, n_reads * sizeof(uint64_t), hipMemcpyHostToDevice, c->load_stream) != hipSuccess) return false;
        if (hipEventRecord(c->ingest_pool[bi].done, c->load_stream) != hipSuccess) return false;
        inflight[bi] = 1;
        cur[worker] ^= 1;
        return true;
    }
};

void absorb_summary(commet_readset *rs, const commet_host::PackSummary &sm)
{
    rs->host_packed = true;
    rs->host_min_len = std::min(rs->host_min_len, sm.min_len);
    rs->host_max_len = std::max(rs->host_max_len, sm.max_len);
    rs->empty_reads.insert(rs->empty_reads.end(), sm.empty_reads.begin(), sm.empty_reads.end());
}

}  // namespace

extern "C" {

int commet_readset_append(commet_readset *rs, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads)
{
    if (rs->finalized) return fail("read set already finalized");
    if (rs->files.empty()) return fail("commet_readset_begin_file must be called first");
    if (rs->acquired) return fail("append with an uncommitted staging buffer");
    if (n_reads == 0) return 0;
    if (offsets[0] != 0) return fail("offsets[0] must be 0");
    const uint64_t nbases = offsets[n_reads];
    if (rs->n_reads + n_reads > rs->max_reads || rs->n_bases + nbases > rs->max_bases)
        return fail("read set capacity exceeded (%llu reads / %llu bases reserved)", (unsigned long long) rs->max_reads,
                    (unsigned long long) rs->max_bases);
    HipPackSink sink;
    const int T = commet_host::ingest_threads();
    const bool verbose = rs->ctx->ingest_verbose;
    const auto tv0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tv0).count(); };
    if (!sink.prepare(rs, T)) return fail("cannot allocate the ingest staging buffers");
    if (verbose) fprintf(stderr, "[ingest] staging ready      %8.1f ms\n", since());
    commet_host::PackSummary sm;
    std::string err;
    const bool ok = commet_host::ingest_arrays(bases, offsets, n_reads, rs->n_reads, rs->n_bases, T, sink, sm, err);
    if (verbose) fprintf(stderr, "[ingest] packed + queued    %8.1f ms\n", since());
    // the staging buffers go back to the pool only once their copies are done
    const bool synced = hipStreamSynchronize(rs->ctx->load_stream) == hipSuccess;
    if (verbose) fprintf(stderr, "[ingest] uploaded           %8.1f ms\n", since());
    if (!synced && ok) return fail("upload failed: %s", hipGetErrorString(hipGetLastError()));
    if (!ok) return fail("%s", err.empty() ? "read set ingest failed" : err.c_str());
    absorb_summary(rs, sm);
    rs->n_reads += n_reads;
    rs->n_bases += nbases;
    rs->files.back().count += n_reads;
    return 0;
}

commet_readset *commet_readset_from_fasta(commet_ctx *c, const char *const *paths, int n_paths)
{
    std::vector<std::unique_ptr<commet_host::ReadFileData>> maps;
    std::vector<const char *> data;
    std::vector<uint64_t> sizes;
    // one thread per file maps it or, when gzipped, inflates it (a zlib stream is sequential; files are independent)
    std::vector<std::future<std::unique_ptr<commet_host::ReadFileData>>> opening;
    for (int i = 0; i < n_paths; ++i) {
        const std::string path = paths[i];
        opening.push_back(std::async(std::launch::async, [path]() {
            std::unique_ptr<commet_host::ReadFileData> f(new commet_host::ReadFileData);
            if (!f->open_file(path)) f.reset();
            return f;
        }));
    }
    for (int i = 0; i < n_paths; ++i) {
        std::unique_ptr<commet_host::ReadFileData> mf = opening[i].get();
        if (!mf) {
            fail("Cannot open file %s", paths[i]);
            return nullptr;
        }
        if (mf->format() == commet_host::ReadFormat::Unknown) {
            fail("Unknown format: %s", paths[i]);
            return nullptr;
        }
        data.push_back(mf->data());
        sizes.push_back(mf->size());
        maps.push_back(std::move(mf));
    }
    return commet_readset_from_buffers(c, data.data(), sizes.data(), n_paths);
}

commet_readset *commet_readset_from_buffers(commet_ctx *c, const char *const *data, const uint64_t *sizes, int n_paths)
{
    std::vector<const char *> d(data, data + n_paths);
    std::vector<size_t> n(sizes, sizes + n_paths);
    std::vector<commet_host::ReadFormat> fmts;
    for (int i = 0; i < n_paths; ++i) {
        fmts.push_back(commet_host::sniff_format(data[i], (size_t) sizes[i]));
        if (fmts.back() == commet_host::ReadFormat::Unknown) {
            fail("Unknown format: file %d of the set is neither FASTA nor FASTQ text", i);
            return nullptr;
        }
    }
    const bool verbose = c->ingest_verbose;
    const auto tv0 = std::chrono::steady_clock::now();
    commet_readset *rs = nullptr;
    HipPackSink sink;
    std::vector<uint64_t> file_reads;
    uint64_t total_reads = 0, total_bases = 0;
    commet_host::PackSummary sm;
    std::string err;
    const bool ok = commet_host::ingest_files<HipPackSink>(
        d, n, fmts, commet_host::ingest_threads(),
        [&](uint64_t reads, uint64_t bases, int workers) -> HipPackSink * {
            if (verbose)
                fprintf(stderr, "[ingest] counted            %8.1f ms\n",
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tv0).count());
            rs = commet_readset_create(c, reads, bases);
            if (!rs || !sink.prepare(rs, workers)) return nullptr;
            return &sink;
        },
        file_reads, total_reads, total_bases, sm, err);
    if (rs) (void) hipStreamSynchronize(c->load_stream);
    if (verbose)
        fprintf(stderr, "[ingest] packed + uploaded  %8.1f ms\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tv0).count());
    if (!ok) {
        if (rs) fail("%s", err.empty() ? "read set ingest failed" : err.c_str());   // (else the failing call has set the message)
        if (rs) commet_readset_destroy(rs);
        return nullptr;
    }
    uint64_t pos = 0;
    for (int f = 0; f < n_paths; ++f) {
        rs->files.push_back(FileSpan{pos, file_reads[f]});
        pos += file_reads[f];
    }
    rs->n_reads = total_reads;
    rs->n_bases = total_bases;
    absorb_summary(rs, sm);
    return rs;
}


uint64_t commet_readset_file_reads(const commet_readset *rs, uint64_t file_index)
{
    return file_index < rs->files.size() ? rs->files[file_index].count : 0;
}

// host copy of the per-read k-mer counts + prefix sums, on first need (a set that is only searched never needs them)
static int host_counts(const commet_readset *rs)
{
    if (rs->have_host_counts) return 0;
    HIP_OK(hipSetDevice(rs->ctx->device));
    rs->h_kcnt.resize(rs->n_reads);
    if (rs->n_reads) HIP_OK(hipMemcpy(rs->h_kcnt.data(), rs->d_kcnt, rs->n_reads * sizeof(uint32_t), hipMemcpyDeviceToHost));
    build_kmer_prefix(rs->h_kcnt.data(), rs->n_reads, rs->h_kprefix);
    rs->have_host_counts = true;
    return 0;
}

int commet_readset_finalize(commet_readset *rs)
{
    if (rs->finalized) return 0;
    if (rs->acquired) return fail("finalize with an uncommitted staging buffer");
    commet_ctx *c = rs->ctx;
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->load_stream));
    rs->st[0].inflight = rs->st[1].inflight = false;
    // shortest / longest read: from the packing kernel (reads that came through the staging API) and from the host
    // packer (append / from_fasta); the host copy of the per-read counts and their prefix sums (chunk planning) are made
    // when the set is first used as an index set (host_counts)
    uint32_t mm[3] = {0xFFFFFFFFu, 0, 0};
    if (rs->n_reads) HIP_OK(hipMemcpy(mm, rs->d_lenmm, sizeof mm, hipMemcpyDeviceToHost));
    mm[0] = std::min(mm[0], rs->host_min_len);
    mm[1] = std::max(mm[1], rs->host_max_len);
    rs->uniform_len = (rs->n_reads && mm[0] == mm[1] && mm[0] != 0) ? mm[0] : 0;
    rs->max_len = rs->n_reads ? mm[1] : 0;
    rs->min_len = rs->n_reads ? mm[0] : 0;
    if (rs->host_packed && rs->n_reads) {
        // host-packed reads have no counts yet: complete k-mers of every read from its validity plane, on the device
        const uint64_t nb = rs->n_bases;
        HIP_OK(hipMemcpyAsync(rs->d_goff + rs->n_reads, &nb, sizeof nb, hipMemcpyHostToDevice, c->load_stream));   // closes the offsets
        COMMET_LAUNCH(kmer_counts_kernel, dim3((unsigned) ((rs->n_reads + 255) / 256)), dim3(256), 0, c->load_stream, rs->view(), c->k,
                           rs->d_kcnt, rs->d_lenmm);
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(c->load_stream));
        HIP_OK(hipMemcpy(mm, rs->d_lenmm, sizeof mm, hipMemcpyDeviceToHost));
    }
    rs->max_kcnt = rs->n_reads ? mm[2] : 0;
    // first-hit windows of the set for this context's (k, t) — the records its query list can hold at most (tile_search.hpp): n x
    // (len - t k + 1) for reads of one length; summed over the reads on the device otherwise (sizing the list of a ragged set by its
    // LONGEST read kept 50-150 bp sets out of the tiled search: 7 GB estimated for a 2.3 GB list)
    {
        const int64_t tk = (int64_t) std::min<uint64_t>((uint64_t) c->t, (uint64_t) rs->max_len / (uint64_t) c->k + 1) * c->k;   // (t_eff, search_dispatch.hpp)
        if (!rs->n_reads) rs->fhw_total = 0;
        else if (rs->uniform_len) rs->fhw_total = rs->n_reads * (uint64_t) std::max<int64_t>(0, (int64_t) rs->uniform_len - tk + 1);
        else {
            unsigned long long *d_sum = (unsigned long long *) (rs->d_lenmm + 4);
            HIP_OK(hipMemsetAsync(d_sum, 0, sizeof(unsigned long long), c->load_stream));
            COMMET_LAUNCH(first_hit_windows_kernel, dim3((unsigned) std::min<uint64_t>((rs->n_reads + 255) / 256, 1u << 16)), dim3(256), 0, c->load_stream,
                          rs->view(), (uint32_t) std::min<int64_t>(tk, 0x7FFFFFFF), d_sum);
            HIP_OK(hipGetLastError());
            unsigned long long h_sum = 0;
            HIP_OK(hipMemcpyAsync(&h_sum, d_sum, sizeof h_sum, hipMemcpyDeviceToHost, c->load_stream));
            HIP_OK(hipStreamSynchronize(c->load_stream));
            rs->fhw_total = h_sum;
        }
    }
    std::sort(rs->empty_reads.begin(), rs->empty_reads.end());
    // the staging buffers are no longer needed: give the memory back
    for (int i = 0; i < 2; ++i) {
        if (rs->st[i].h_bases) (void) hipHostFree(rs->st[i].h_bases);
        if (rs->st[i].h_offs) (void) hipHostFree(rs->st[i].h_offs);
        (void) dm_free(rs->st[i].d_bases);
        (void) dm_free(rs->st[i].d_offs);
        rs->st[i].h_bases = nullptr;
        rs->st[i].h_offs = nullptr;
        rs->st[i].d_bases = nullptr;
        rs->st[i].d_offs = nullptr;
    }
    rs->finalized = true;
    return 0;
}

uint64_t commet_readset_num_reads(const commet_readset *rs) { return rs->n_reads; }
uint64_t commet_readset_num_files(const commet_readset *rs) { return rs->files.size(); }

int commet_readset_kmer_counts(const commet_readset *rs, uint32_t *out)
{
    if (!rs->finalized) return fail("read set not finalized");
    if (host_counts(rs)) return 1;
    if (rs->n_reads) memcpy(out, rs->h_kcnt.data(), rs->n_reads * sizeof(uint32_t));
    return 0;
}

}  // extern "C"
