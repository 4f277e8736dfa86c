// capi/microbench.hpp — the memory-system microbenchmarks behind bench.py's ceilings (random gathers / atomics over a table, LDS atomics)
// (a part of the one translation unit capi.hip: included there, in order, after the kernels and state.hpp)
#pragma once

extern "C" {

int commet_membench(commet_ctx *c, int atomic, uint64_t table_bytes, uint64_t n_access, double *ms_out)
{
    HIP_OK(hipSetDevice(c->device));
    if (atomic == 4 || atomic == 5) {   // streaming ceilings: 4 = device-to-device copy of table_bytes, 5 = fill
        uint8_t *a = nullptr, *b = nullptr;
        HIP_OK(dm_malloc((void **) &a, table_bytes));
        HIP_OK(dm_malloc((void **) &b, table_bytes));
        hipEvent_t e0, e1;
        HIP_OK(hipEventCreate(&e0));
        HIP_OK(hipEventCreate(&e1));
        HIP_OK(hipMemsetAsync(a, 1, table_bytes, c->stream));
        HIP_OK(hipMemsetAsync(b, 2, table_bytes, c->stream));
        for (int rep = 0; rep < 2; ++rep) {
            HIP_OK(hipEventRecord(e0, c->stream));
            if (atomic == 4) HIP_OK(hipMemcpyAsync(b, a, table_bytes, hipMemcpyDeviceToDevice, c->stream));
            else HIP_OK(hipMemsetAsync(b, 3, table_bytes, c->stream));
            HIP_OK(hipEventRecord(e1, c->stream));
        }
        HIP_OK(hipStreamSynchronize(c->stream));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        if (ms_out) *ms_out = ms;
        (void) hipEventDestroy(e0);
        (void) hipEventDestroy(e1);
        (void) dm_free(a);
        (void) dm_free(b);
        return 0;
    }
    if (atomic >= 100) {   // windowed gathers: atomic = 100 + log2(window bytes), +1000 = XCD-aware sweep; n_access gathers in all
        const int xcd = atomic >= 1000 ? 1 : 0;
        const uint32_t win_words = (1u << ((atomic % 1000) - 100)) / 4;
        const uint64_t n_windows = table_bytes / 4 / win_words;
        uint32_t *table = nullptr, *sink = nullptr;
        HIP_OK(dm_malloc((void **) &table, n_windows * win_words * 4));
        HIP_OK(dm_malloc((void **) &sink, 4));
        HIP_OK(hipMemsetAsync(table, 0, n_windows * win_words * 4, c->stream));
        // one resident set of workgroups (8 per CU); every thread does `iters` gathers in each window of its XCD's eighth
        const uint32_t grid = 256 * 8;
        const uint32_t iters = (uint32_t) std::max<uint64_t>(1, n_access / (grid * 256ull) / std::max<uint64_t>(n_windows / 8, 1));
        hipEvent_t e0, e1;
        HIP_OK(hipEventCreate(&e0));
        HIP_OK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            HIP_OK(hipEventRecord(e0, c->stream));
            COMMET_LAUNCH(membench_window_kernel, dim3(grid), dim3(256), 0, c->stream, table, n_windows, win_words, iters, xcd, sink);
            HIP_OK(hipEventRecord(e1, c->stream));
        }
        HIP_OK(hipStreamSynchronize(c->stream));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        if (ms_out) *ms_out = ms / ((double) iters * (double) (n_windows / 8) * grid * 256.0) * (double) n_access;
        (void) hipEventDestroy(e0);
        (void) hipEventDestroy(e1);
        (void) dm_free(table);
        (void) dm_free(sink);
        return 0;
    }
    uint64_t words = 1;
    while (words * 2 * 4 <= table_bytes) words *= 2;   // power of two words
    uint32_t *table = nullptr, *sink = nullptr;
    HIP_OK(dm_malloc((void **) &table, words * 4));
    HIP_OK(dm_malloc((void **) &sink, 4));
    HIP_OK(hipMemsetAsync(table, 0, words * 4, c->stream));
    const uint64_t threads = 256ull * 256 * 32;   // 32 blocks of 256 per CU
    const uint32_t iters = (uint32_t) std::max<uint64_t>(1, n_access / threads);
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {   // rep 0 warms up
        HIP_OK(hipEventRecord(e0, c->stream));
        const dim3 g((unsigned) (threads / 256)), b(256);
        switch (atomic) {
        case 1: COMMET_LAUNCH(membench_kernel<1>, g, b, 0, c->stream, table, words - 1, iters, sink); break;
        case 2: COMMET_LAUNCH(membench_kernel<2>, g, b, 0, c->stream, table, words - 1, iters, sink); break;
        case 3: COMMET_LAUNCH(membench_kernel<3>, g, b, 0, c->stream, table, words - 1, iters, sink); break;
        default: COMMET_LAUNCH(membench_kernel<0>, g, b, 0, c->stream, table, words - 1, iters, sink); break;
        }
        HIP_OK(hipEventRecord(e1, c->stream));
    }
    HIP_OK(hipStreamSynchronize(c->stream));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms / ((double) iters * threads) * (double) n_access;   // scaled to n_access
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    (void) dm_free(table);
    (void) dm_free(sink);
    return 0;
}

int commet_ldsbench(commet_ctx *c, int mode, uint32_t n_words, uint64_t n_access, double *ms_out)
{
    HIP_OK(hipSetDevice(c->device));
    if (n_words == 0 || (n_words & (n_words - 1)) || n_words > 32768) return fail("ldsbench: n_words must be a power of two <= 32768");
    if (mode < 0 || mode > 5) return fail("ldsbench: mode 0..5");
    uint32_t *sink = nullptr;
    HIP_OK(dm_malloc((void **) &sink, 4));
    const uint64_t threads = 512ull * 256 * 8;   // 8 workgroups of 512 per CU (LDS permitting)
    const uint32_t iters = (uint32_t) std::max<uint64_t>(1, n_access / threads);
    const size_t lds = (size_t) n_words * 4;
    const void *fns[6] = {(const void *) ldsbench_kernel<0>, (const void *) ldsbench_kernel<1>, (const void *) ldsbench_kernel<2>,
                          (const void *) ldsbench_kernel<3>, (const void *) ldsbench_kernel<4>, (const void *) ldsbench_kernel<5>};
    HIP_OK(hipFuncSetAttribute(fns[mode], hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {   // rep 0 warms up
        HIP_OK(hipEventRecord(e0, c->stream));
        uint32_t it = iters;
        void *args[] = {&n_words, &it, &sink};
        HIP_OK(hipLaunchKernel(fns[mode], dim3((unsigned) (threads / 512)), dim3(512), args, lds, c->stream));
        HIP_OK(hipEventRecord(e1, c->stream));
    }
    HIP_OK(hipStreamSynchronize(c->stream));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms / ((double) iters * threads) * (double) n_access;
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    (void) dm_free(sink);
    return 0;
}

}  // extern "C"
