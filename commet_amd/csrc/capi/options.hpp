// capi/options.hpp — commet_set_option and the measurement hooks (reference-layout filter export, launched kernels, kernel times)
// (a part of the one translation unit capi.hip: included there, in order, after the kernels and state.hpp)
#pragma once

extern "C" {

int commet_set_option(commet_ctx *c, const char *name, int64_t value)
{
    if (!strcmp(name, "query_list_max_mb")) {      // auto mode: largest list (estimated) a set may get; larger sets keep the gather kernels
        if (value < 0) return fail("query_list_max_mb must be >= 0");
        c->ql_max_list = (uint64_t) value << 20;
        return 0;
    }
    if (!strcmp(name, "query_list_budget_mb")) {   // HBM the cached query lists of this context's read sets may hold (default 64 GiB)
        if (value < 0) return fail("query_list_budget_mb must be >= 0");
        std::lock_guard<std::mutex> lk(c->ql_mu);
        c->ql_budget = (uint64_t) value << 20;
        (void) shrink_query_lists(c, c->ql_budget, false);
        return 0;
    }
    if (!strcmp(name, "count_probes")) {
        c->count_probes = value != 0;
        return 0;
    }
    if (!strcmp(name, "index_mode")) {        // 0 auto, 1 atomic kernel, 2 bucketed construction
        if (value < 0 || value > 2) return fail("index_mode must be 0, 1 or 2");
        c->index_mode = (int) value;
        return 0;
    }
    if (!strcmp(name, "chunk_group")) {       // chunk filters searched per pass over a set (1 = reference order)
        if (value < 1 || value > 8) return fail("chunk_group must be 1..8");
        c->chunk_group = (int) value;
        return 0;
    }
    if (!strcmp(name, "kernel_timing")) {     // 1: time every kernel of the following commet_index_and_search calls (totals reset)
        HIP_OK(hipSetDevice(c->device));
        HIP_OK(hipStreamSynchronize(c->stream));
        c->kclock.collect();
        c->kclock.on = value != 0;
        if (value) c->kclock.reset();
        return 0;
    }
    if (!strcmp(name, "tiled_search")) {      // 0 auto, 1 never, 2 whenever the set and the group allow it (tests)
        if (value < 0 || value > 2) return fail("tiled_search must be 0, 1 or 2");
        c->tiled_mode = (int) value;
        return 0;
    }
    if (!strcmp(name, "multi_job")) {         // 0 = commet_index_many_and_search puts the chunk filters of several jobs into one pass where it can, 1 = job by job
        if (value < 0 || value > 1) return fail("multi_job must be 0 or 1");
        c->multi_job = (int) value;
        return 0;
    }
    if (!strcmp(name, "sparse_search")) {     // 0 auto, 1 never, 2 whenever a selection applies: a pass over few of a set's reads walks their list (kernels.hpp, ActiveList)
        if (value < 0 || value > 2) return fail("sparse_search must be 0, 1 or 2");
        c->sparse_search = (int) value;
        return 0;
    }
    if (!strcmp(name, "tq_parts")) {          // tiled search: parts of the set whose replay runs beside the next part's probe (1 = off)
        if (value < 1 || value > 16) return fail("tq_parts must be 1..16");
        c->tq_parts = (int) value;
        return 0;
    }
    if (!strcmp(name, "slice_mode")) {        // 0 auto (8 chunks or more, 12 <= k <= 24), 1 never, 2 whenever k allows it
        if (value < 0 || value > 2) return fail("slice_mode must be 0, 1 or 2");
        c->slice_mode = (int) value;
        return 0;
    }
    if (!strcmp(name, "slice_words")) {       // chunk filters per pass / 32 in the sliced regime: 0 auto, 1, 2, 4 or 8
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return fail("slice_words must be 0, 1, 2, 4 or 8");
        c->slice_gw = (int) value;
        return 0;
    }
    if (!strcmp(name, "slice_wide")) {        // wide rows in the many-small-chunks regime: 0 auto (more than 256 chunks), 1 never, 2 always
        if (value < 0 || value > 2) return fail("slice_wide must be 0, 1 or 2");
        c->slice_wide = (int) value;
        return 0;
    }
    if (!strcmp(name, "slice_wide_words")) {  // cap on the words per wide row (32 chunk filters each; tests: several passes); 0 = the budget decides
        if (value < 0 || value > 512 || value % 8) return fail("slice_wide_words must be a multiple of 8 in 0..512");
        c->wide_cap_words = (uint32_t) value;
        return 0;
    }
    if (!strcmp(name, "max_kmer")) {          // chunk size in k-mers (0 = the reference's constant); changes the chunking
        if (value < 0) return fail("max_kmer must be >= 0");
        c->max_kmer_test = (uint64_t) value;
        return 0;
    }
    if (!strcmp(name, "index_lanes")) {       // 1 = the chunks of a group are built one after the other
        if (value < 1 || value > 2) return fail("index_lanes must be 1 or 2");
        c->index_lanes = (int) value;
        return 0;
    }
    if (!strcmp(name, "lane_stagger")) {      // 1 = the second index lane's chunk starts behind the first lane's scatter1 (default), 0 = both at once
        c->lane_stagger = value != 0;
        return 0;
    }
    if (!strcmp(name, "drop_workspaces")) {   // frees the scatter workspaces; the next bucketed index build allocates them again
        (void) hipStreamSynchronize(c->stream);
        (void) hipStreamSynchronize(c->aux_stream);
        for (auto &w : c->part) {
            (void) dm_free(w.bufA); (void) dm_free(w.bufB);
            w.bufA = w.bufB = nullptr;
            w.cap_keys = 0;
        }
        return 0;
    }
    if (!strcmp(name, "part_packed")) {
        c->part_packed = value != 0;
        return 0;
    }
    if (!strcmp(name, "part_no_uni")) {
        c->part_no_uni = value != 0;
        return 0;
    }
    if (!strcmp(name, "ordered_scan")) {
        c->ordered_scan = (int) std::max<int64_t>(0, std::min<int64_t>(value, 2));
        return 0;
    }
    if (!strcmp(name, "mask_split")) {
        c->mask_split = value != 0;
        return 0;
    }
    if (!strcmp(name, "tq_hit_cap")) {
        c->tq_hit_cap = (int) std::max<int64_t>(0, std::min<int64_t>(value, TQ_HIT_CAP));
        return 0;
    }
    if (!strcmp(name, "part_list")) {
        c->part_list = value != 0;
        return 0;
    }
    if (!strcmp(name, "part_b1")) {
        c->part_b1 = (int) value;
        return 0;
    }
    if (!strcmp(name, "s2_swizzle")) {
        c->s2_swizzle = (int) std::max<int64_t>(0, std::min<int64_t>(value, 1 << 20));
        return 0;
    }
    if (!strcmp(name, "part_min_kmers")) {    // auto mode: chunks with fewer k-mers use the atomic kernel
        c->part_min_kmers = (uint64_t) value;
        return 0;
    }
    return fail("unknown option '%s'", name);
}

int commet_filter_export_reference(commet_ctx *c, uint8_t *out, uint64_t out_bytes)
{
    const uint64_t nbytes = (uint64_t) pow(2, c->k - 1);   // bloom_filter.h:73
    if (out_bytes < nbytes) return fail("export buffer too small");
    if (nbytes == 0) return 0;
    HIP_OK(hipSetDevice(c->device));
    uint8_t *d_out = nullptr;
    HIP_OK(dm_malloc((void **) &d_out, nbytes));
    const uint64_t blocks = std::min<uint64_t>((nbytes + 255) / 256, 1u << 20);   // grid-stride beyond
    COMMET_LAUNCH(export_reference_kernel, dim3((unsigned) blocks), dim3(256), 0, c->stream, c->view(), c->k, nbytes, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, nbytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void) dm_free(d_out);
    if (e != hipSuccess) return fail("filter export failed: %s", hipGetErrorString(e));
    return 0;
}

int commet_launched_kernels(const void **out, int cap, int *n_out)
{
    std::lock_guard<std::mutex> lk(g_launch_mu);
    int i = 0;
    for (const void *e : g_launched) {
        if (i < cap && out) out[i] = e;
        ++i;
    }
    if (n_out) *n_out = i;
    return 0;
}

int commet_last_kernel_ms(commet_ctx *c, double *index_ms, double *search_ms)
{
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    float ms = 0;
    if (index_ms) {
        *index_ms = 0;
        if (c->have_index_ev) {
            HIP_OK(hipEventElapsedTime(&ms, c->ev_i0, c->ev_i1));
            *index_ms = ms;
        }
    }
    if (search_ms) {
        *search_ms = 0;
        if (c->have_search_ev) {
            HIP_OK(hipEventElapsedTime(&ms, c->ev_s0, c->ev_s1));
            *search_ms = ms;
        }
    }
    return 0;
}

int commet_kernel_times(commet_ctx *c, commet_kernel_time *out, int cap, int *n_out)
{
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    c->kclock.collect();
    const int n = (int) c->kclock.names.size();
    if (n_out) *n_out = n;
    for (int i = 0; i < n && i < cap; ++i) {
        snprintf(out[i].name, sizeof out[i].name, "%s", c->kclock.names[i].c_str());
        out[i].launches = c->kclock.launches[i];
        out[i].total_ms = c->kclock.total_ms[i];
    }
    return 0;
}

}  // extern "C"
