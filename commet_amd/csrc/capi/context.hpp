// capi/context.hpp — commet_create / _destroy and the small context queries of the ABI
// (a part of the one translation unit capi.hip: included there, in order, after the kernels and state.hpp)
#pragma once

extern "C" {

const char *commet_version(void) { return "commet-amd 0.1 (gfx950)"; }
const char *commet_last_error(void) { return g_err.c_str(); }

int commet_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

commet_ctx *commet_create(int device, int kmer_size, int min_hits)
{
    if (kmer_size < 1 || kmer_size > 38) {
        fail("k-mer size %d out of range [1,38]", kmer_size);
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fail("no HIP device available: the index_and_search path has no CPU fallback");
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        fail("device %d out of range (have %d)", device, ndev);
        return nullptr;
    }
    HIP_OK_NULL(hipSetDevice(device));
    commet_ctx *c = new commet_ctx;
    c->device = device;
    c->k = kmer_size;
    c->t = min_hits < 1 ? 1 : min_hits;
    if (const char *e = getenv("COMMET_INDEX_LANES")) c->index_lanes = atoi(e) == 1 ? 1 : 2;   // 1: one kernel at a time (per-kernel profiles)
    if (const char *e = getenv("COMMET_TILED")) c->tiled_mode = std::max(0, std::min(2, atoi(e)));
    if (const char *e = getenv("COMMET_MULTI_JOB")) c->multi_job = atoi(e) == 1 ? 1 : 0;   // A/B runs: 1 = commet_index_many_and_search job by job
    if (const char *e = getenv("COMMET_SPARSE_SEARCH")) c->sparse_search = std::max(0, std::min(2, atoi(e)));   // A/B runs
    if (const char *e = getenv("COMMET_TQ_SBITS")) c->tq_sbits = atoi(e);
    if (const char *e = getenv("COMMET_TQ_WPX")) c->tq_wpx = (unsigned) std::max(1, atoi(e));
    if (const char *e = getenv("COMMET_TQ_PARTS")) c->tq_parts = std::max(1, std::min(16, atoi(e)));
    if (const char *e = getenv("COMMET_LANE_STAGGER")) c->lane_stagger = atoi(e) != 0;
    c->stage_reads = getenv("COMMET_NO_STAGE_READS") == nullptr;
    c->job_verbose = getenv("COMMET_JOB_VERBOSE") != nullptr;
    c->ingest_verbose = getenv("COMMET_INGEST_VERBOSE") != nullptr;
    {   // query lists: half the device at most (64 GiB on small devices).  One list: 4 GiB (sets of up to ~15 M reads).  Larger lists
        // (COMMET_QUERY_LIST_MAX_GB / option query_list_max_mb; a 50 M-read set's is 11 GB) pay in a long-lived context — a J2 / J3
        // job of configs[3] 54.7 against 62.3 ms, its matrix 11.1-11.4 against 11.8 s when the device memory comes cheap — but not
        // where the memory is allocated for the one run: hipMalloc + hipFree of large buffers cost 15-30 ms per GiB on this driver
        // (64 GiB: 1.9 s, 128 GiB: 3.8-4.8 s, commet_membench's table), ten lists = 110 GB: the same matrix 14.7-15.0 s
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) c->ql_budget = std::max<uint64_t>(64ull << 30, (uint64_t) tot / 2);
    }
    if (const char *e = getenv("COMMET_QUERY_LIST_MAX_GB")) c->ql_max_list = (uint64_t) (std::max(0.0, atof(e)) * (double) (1ull << 30));
    if (const char *e = getenv("COMMET_QUERY_LIST_GB")) c->ql_budget = (uint64_t) (std::max(0.0, atof(e)) * (double) (1ull << 30));
    if (const char *e = getenv("COMMET_SLICE_MODE")) c->slice_mode = std::max(0, std::min(2, atoi(e)));     // A/B runs of bench.py
    if (const char *e = getenv("COMMET_SLICE_WIDE")) c->slice_wide = std::max(0, std::min(2, atoi(e)));     // A/B runs of bench.py
    if (const char *e = getenv("COMMET_SLICE_WORDS")) {
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8) c->slice_gw = v;
    }
    // 2^k bits per plane, at least one word; 4 planes = 2^(k-1) bytes (bloom_filter.h:73)
    const uint64_t plane_bits = 1ull << kmer_size;
    c->plane_words = plane_bits < 32 ? 1 : plane_bits / 32;
    c->filter_bytes = 4 * c->plane_words * sizeof(uint32_t);
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = dm_malloc((void **) &c->filter, c->filter_bytes);
    if (e != hipSuccess) {
        if (e == hipErrorOutOfMemory)
            fail("Index memory allocation impossible, try with a lower k value or with more RAM memory");
        else fail("context creation failed: %s", hipGetErrorString(e));
        commet_destroy(c);
        return nullptr;
    }
    e = dm_malloc((void **) &c->d_counters, N_COUNTERS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc((void **) &c->h_counters, N_COUNTERS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipEventCreate(&c->ev_i0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_i1);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_s0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_s1);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->load_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_stagger, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->list_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_list, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMemsetAsync(c->filter, 0, c->filter_bytes, c->stream);
    if (e != hipSuccess) {
        fail("context creation failed: %s", hipGetErrorString(e));
        commet_destroy(c);
        return nullptr;
    }
    return c;
}

void commet_destroy(commet_ctx *c)
{
    if (!c) return;
    (void) hipSetDevice(c->device);
    if (c->stream) (void) hipStreamSynchronize(c->stream);
    if (c->filter) (void) dm_free(c->filter);
    for (commet_ctx::IngestBuf &b : c->ingest_pool) {
        if (b.h_planes) (void) hipHostFree(b.h_planes);
        if (b.h_goff) (void) hipHostFree(b.h_goff);
        if (b.done) (void) hipEventDestroy(b.done);
    }
    c->kclock.release();
    (void) dm_free(c->d_qres);
    (void) dm_free(c->slice_stage);
    (void) dm_free(c->slice_tables);
    (void) dm_free(c->wide_tables);
    (void) dm_free(c->d_slice_chunks);
    (void) dm_free(c->il_a);
    (void) dm_free(c->d_jobcnt);
    (void) dm_free(c->d_plansum);
    (void) dm_free(c->d_ids);
    (void) dm_free(c->d_idblk);
    (void) dm_free(c->d_lo_cnt);
    (void) dm_free(c->d_ids2);
    (void) dm_free(c->d_idblk2);
    (void) dm_free(c->d_mtags);
    (void) dm_free(c->d_act);
    (void) dm_free(c->d_actblk);
    c->part[0].release();
    c->part[1].release();
    if (c->aux_stream) (void) hipStreamSynchronize(c->aux_stream), (void) hipStreamDestroy(c->aux_stream);
    if (c->load_stream) (void) hipStreamSynchronize(c->load_stream), (void) hipStreamDestroy(c->load_stream);
    if (c->ev_stagger) (void) hipEventDestroy(c->ev_stagger);
    if (c->list_stream) (void) hipStreamSynchronize(c->list_stream), (void) hipStreamDestroy(c->list_stream);
    if (c->ev_list) (void) hipEventDestroy(c->ev_list);
    (void) dm_free(c->d_ql_totals);
    if (c->ev_fork) (void) hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void) hipEventDestroy(c->ev_join);
    if (c->d_counters) (void) dm_free(c->d_counters);
    if (c->h_counters) (void) hipHostFree(c->h_counters);
    if (c->ev_i0) (void) hipEventDestroy(c->ev_i0);
    if (c->ev_i1) (void) hipEventDestroy(c->ev_i1);
    if (c->ev_s0) (void) hipEventDestroy(c->ev_s0);
    if (c->ev_s1) (void) hipEventDestroy(c->ev_s1);
    if (c->stream) (void) hipStreamDestroy(c->stream);
    delete c;
}

int commet_kmer_size(const commet_ctx *c) { return c->k; }
int commet_min_hits(const commet_ctx *c) { return c->t; }

uint64_t commet_max_kmer(const commet_ctx *c)
{
    if (c->max_kmer_test) return c->max_kmer_test;          // test hook, see commet_set_option
    return (uint64_t) (1000000000.0 / pow(2, 33 - c->k));   // index_and_search.cpp:73,146
}

int commet_device_memory(const commet_ctx *c, uint64_t *free_bytes, uint64_t *total_bytes)
{
    HIP_OK(hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIP_OK(hipMemGetInfo(&f, &t));
    f += dm_filed_bytes(c->device);                          // (blocks the library keeps for reuse are free to its callers)
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}

int commet_synchronize(commet_ctx *c)
{
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
