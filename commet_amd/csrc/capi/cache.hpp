// capi/cache.hpp — device memory under pressure: the query lists cached with resident sets (accounted, least recently used first), allocations that retry after giving lists back, scatter workspaces picked from several candidates; the cache entry points of the ABI
// (a part of the one translation unit capi.hip: included there, in order, after the kernels and state.hpp)
#pragma once

namespace {
// drops one set's query list (caller holds ql_mu); hipFree waits for the kernels that read it
void drop_query_list(commet_ctx *c, const commet_readset *rs)
{
    if (!rs->ql.built && !rs->ql.bytes) return;
    c->ql_bytes -= std::min(c->ql_bytes, rs->ql.bytes);
    rs->ql.release();
    ++c->ql_evictions;
}

// gives cached query lists back until at most `target` bytes of them are left: least recently used first, never a list
// of the running job unless `even_in_job` (the job thread itself is out of memory and holds no list between build and
// launch).  Returns the bytes released.  Caller holds ql_mu.
uint64_t shrink_query_lists(commet_ctx *c, uint64_t target, bool even_in_job)
{
    uint64_t freed = 0;
    while (c->ql_bytes > target) {
        const commet_readset *victim = nullptr;
        for (const commet_readset *rs : c->sets)
            if (rs->ql.built && (even_in_job || !rs->in_job) && (!victim || rs->ql.last_use < victim->ql.last_use)) victim = rs;
        if (!victim) break;
        freed += victim->ql.bytes;
        drop_query_list(c, victim);
    }
    return freed;
}

void trim_ws_pool(commet_ctx *c);

// hipMalloc that gives the cached query lists (and workspace candidates still waiting in the pool) back and tries once more when
// the device is out of memory
hipError_t dev_alloc(commet_ctx *c, void **p, size_t bytes, bool job_thread)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipErrorOutOfMemory) return e;
    (void) hipGetLastError();
    uint64_t freed;
    {
        std::lock_guard<std::mutex> lk(c->ws_pool_mu);
        freed = c->ws_pool.size();
    }
    trim_ws_pool(c);
    {
        std::lock_guard<std::mutex> lk(c->ql_mu);
        freed += shrink_query_lists(c, 0, job_thread);
    }
    if (!freed) return e;
    e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) (void) hipGetLastError();
    return e;
}

// A buffer of several GB that kernels sweep as a whole — the scatter workspaces: 2^17 output streams all over it — is as fast as
// hipMalloc happened to back it: one allocation in three streams ~20 % faster than the others (fills of 5.8 GB at 5.4 against
// 4.5 TB/s, scatter1 4.2 against 5.0 ms per configs[1] step), the draw is made once per allocation, a fill shows it as well as
// the kernels do, and most of the box-to-box spread of the whole step was this.  So the first workspace of a
// context is picked from `candidates` buffers: each touched, a second fill of each timed, the fastest kept.  The others are NOT
// freed at once: they wait in the context's pool, and the workspaces asked for next — the same job's other buffer, the second
// index lane's two — take the fastest one left that is large enough.  Four candidates thus serve the four buffers of a two-lane
// context with no allocation beyond the four it needs anyway; what is left in the pool is freed when the job ends.  (Freeing tens
// of GB and allocating again is not free on this driver: a hipMalloc of 8.5 GB now and then blocks for 1-2 s when the process has
// just given that much back — measured with 4 candidates PER buffer, 12 of 16 freed at once — so churn is what to avoid.)
void trim_ws_pool(commet_ctx *c)
{
    std::lock_guard<std::mutex> lk(c->ws_pool_mu);
    for (auto &b : c->ws_pool) (void) hipFree(b.ptr);
    c->ws_pool.clear();
}

hipError_t alloc_fastest(commet_ctx *c, void **p, size_t bytes, hipStream_t stream, int candidates, const char *what)
{
    *p = nullptr;
    {   // a timed buffer left over from an earlier pick: the fastest that fits.  A much larger one (the second lane's chunk is often a
        // third of the first's) is taken, too, while the device has memory to spare: keeping it costs nothing then, freeing it is churn
        size_t fr = 0, tot = 0;
        const bool roomy = hipMemGetInfo(&fr, &tot) == hipSuccess && fr > ((size_t) 64 << 30);
        std::lock_guard<std::mutex> lk(c->ws_pool_mu);
        int best = -1;
        for (int i = 0; i < (int) c->ws_pool.size(); ++i) {
            const auto &b = c->ws_pool[i];
            if (b.bytes < bytes || (!roomy && b.bytes > 2 * bytes + (64u << 20))) continue;
            // (fill times of buffers of different sizes compare per byte)
            if (best < 0 || b.ms / (double) b.bytes < c->ws_pool[best].ms / (double) c->ws_pool[best].bytes) best = i;
        }
        if (best >= 0) {
            *p = c->ws_pool[best].ptr;
            if (c->ws_verbose)
                fprintf(stderr, "commet: %s, %.2f GB: from the pool (%.2f GB, fill %.3f ms, %zu left)\n", what, bytes / 1e9, c->ws_pool[best].bytes / 1e9,
                        c->ws_pool[best].ms, c->ws_pool.size() - 1);
            c->ws_pool.erase(c->ws_pool.begin() + best);
            return hipSuccess;
        }
    }
    int n = std::max(1, std::min(candidates, 8));
    {   // never more candidates than the device holds with room to spare (several processes may share it)
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
            const size_t spare = (size_t) 16 << 30;
            n = (int) std::max<size_t>(1, std::min<size_t>((size_t) n, fr > spare ? (fr - spare) / std::max<size_t>(bytes, 1) : 1));
        }
    }
    void *cand[8] = {nullptr};
    float ms[8] = {0};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (n > 1 && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) return hipErrorUnknown;
    int got = 0;
    hipError_t err = hipSuccess;
    double alloc_ms[8] = {0};
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int i = 0; i < n; ++i) {
        const double a0 = now_ms();
        // (only the first candidate may take memory back from the cached query lists: the others are optional)
        err = i == 0 ? dev_alloc(c, &cand[i], bytes, true) : hipMalloc(&cand[i], bytes);
        alloc_ms[i] = now_ms() - a0;
        if (err != hipSuccess) break;                        // (the candidates before this one are still candidates)
        ++got;
        // first touch here, not inside the first scatter launch (measured: 15.6 ms instead of 2.5 ms for that one launch)
        err = hipMemsetAsync(cand[i], 0, bytes, stream);
        if (err == hipSuccess && n > 1) {
            (void) hipEventRecord(e0, stream);
            err = hipMemsetAsync(cand[i], 0, bytes, stream);
            (void) hipEventRecord(e1, stream);
            if (err == hipSuccess) err = hipStreamSynchronize(stream);
            if (err == hipSuccess) err = hipEventElapsedTime(&ms[i], e0, e1);
        }
        if (err != hipSuccess) break;
    }
    if (e0) (void) hipEventDestroy(e0);
    if (e1) (void) hipEventDestroy(e1);
    if (got == 0) return err == hipSuccess ? hipErrorOutOfMemory : err;
    int best = 0;
    for (int i = 1; i < got; ++i)
        if (ms[i] > 0 && (ms[best] <= 0 || ms[i] < ms[best])) best = i;
    {
        std::lock_guard<std::mutex> lk(c->ws_pool_mu);
        for (int i = 0; i < got; ++i)
            if (i != best) c->ws_pool.push_back({cand[i], bytes, ms[i]});
    }
    if (c->ws_verbose) {
        fprintf(stderr, "commet: %s, %.2f GB, %d candidate(s), fill ms:", what, bytes / 1e9, got);
        for (int i = 0; i < got; ++i) fprintf(stderr, " %.3f%s", ms[i], i == best ? "*" : "");
        fprintf(stderr, "; hipMalloc ms:");
        for (int i = 0; i < got; ++i) fprintf(stderr, " %.1f", alloc_ms[i]);
        fprintf(stderr, "; the others wait in the pool\n");
    }
    *p = cand[best];
    (void) hipGetLastError();                               // (a failed extra candidate is not an error of the caller)
    return hipSuccess;
}
}  // namespace

extern "C" {

uint64_t commet_readset_cache_bytes(const commet_readset *rs)
{
    std::lock_guard<std::mutex> lk(rs->ctx->ql_mu);
    return rs->ql.built ? rs->ql.bytes : 0;
}

void commet_readset_drop_cache(commet_readset *rs)
{
    commet_ctx *c = rs->ctx;
    (void) hipSetDevice(c->device);
    std::lock_guard<std::mutex> lk(c->ql_mu);
    if (rs->in_job) return;                              // (never under a running job)
    drop_query_list(c, rs);
    rs->ql.failed = false;                               // a list that did not fit once may fit now
}

int commet_cache_stats(commet_ctx *c, uint64_t *bytes, uint64_t *budget_bytes, uint64_t *evictions)
{
    std::lock_guard<std::mutex> lk(c->ql_mu);
    if (bytes) *bytes = c->ql_bytes;
    if (budget_bytes) *budget_bytes = c->ql_budget;
    if (evictions) *evictions = c->ql_evictions;
    return 0;
}

}  // extern "C"
