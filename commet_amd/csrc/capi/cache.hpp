// capi/cache.hpp — device memory under pressure: the query lists cached with resident sets (accounted, least recently used first), allocations that retry after giving lists back, scatter workspaces; the cache entry points of the ABI
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

// hipMalloc that gives the cached query lists back and tries once more when the device is out of memory
hipError_t dev_alloc(commet_ctx *c, void **p, size_t bytes, bool job_thread)
{
    hipError_t e = dm_malloc(p, bytes);
    if (e != hipErrorOutOfMemory) return e;
    (void) hipGetLastError();
    uint64_t freed;
    {
        std::lock_guard<std::mutex> lk(c->ql_mu);
        freed = shrink_query_lists(c, 0, job_thread);
    }
    if (!freed) return e;
    e = dm_malloc(p, bytes);
    if (e == hipErrorOutOfMemory) (void) hipGetLastError();
    return e;
}

// A scatter workspace (several GB that kernels sweep as a whole): allocated and touched here, not inside the first scatter launch
// (measured: 15.6 ms instead of 2.5 ms for that one launch).  Rounds 3 and 4 picked these buffers from several timed candidates —
// how hipMalloc backs a multi-GB buffer decides how fast kernels sweep it, drawn once per allocation: one in three ~20 % faster on
// some boxes — but on the boxes of round 4 the candidates lay within 5 % of each other, a fill's time did not predict the scatter
// kernels' inside that range, the spread of the step over fresh contexts was 6.1 % with the pool against 8.8 % without
// (profiles/r04_ws_candidates), and the selection cost ~10 ms of every cold first job and three extra 8.5 GB allocations: removed
// in round 5.
hipError_t alloc_workspace(commet_ctx *c, void **p, size_t bytes, hipStream_t stream)
{
    *p = nullptr;
    hipError_t e = dev_alloc(c, p, bytes, true);
    if (e != hipSuccess) return e;
    return hipMemsetAsync(*p, 0, bytes, stream);
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

uint64_t commet_device_cache_trim(int device)
{
    return (uint64_t) dm_trim(device);
}

uint64_t commet_device_cache_bytes(int device)
{
    return (uint64_t) dm_filed_bytes(device);
}

uint64_t commet_device_pooled_bytes(int device)
{
    return (uint64_t) dm_pooled_bytes(device);
}

int commet_device_alloc_stats(int device, double *wait_ms, uint64_t *fresh_bytes, uint64_t *calls)
{
    uint64_t ns = 0, by = 0, n = 0;
    for (int d = 0; d < 16; ++d) {
        if (device >= 0 && d != device) continue;
        ns += g_devmem.drv_ns[d].load(), by += g_devmem.drv_bytes[d].load(), n += g_devmem.drv_calls[d].load();
    }
    if (wait_ms) *wait_ms = (double) ns * 1e-6;
    if (fresh_bytes) *fresh_bytes = by;
    if (calls) *calls = n;
    return 0;
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
