// capi/index_dispatch.hpp — which index construction runs for a chunk (atomic kernel or bucketed build, index_part.hpp) and its launches; commet_filter_reset
// (a part of the one translation unit capi.hip: included there, in order, after the kernels and state.hpp)
#pragma once

extern "C" {

/* ---- kernels -------------------------------------------------------------- */

int commet_filter_reset(commet_ctx *c)
{
    HIP_OK(hipSetDevice(c->device));
    KScope ks(c, "filter_memset", c->stream);
    HIP_OK(hipMemsetAsync(c->slot_ptr(c->cur_slot), 0, c->filter_bytes, c->stream));
    return 0;
}

}  // extern "C"

namespace {

// uploads a host bit array (n/8+1 bytes) into a device bitmap of bitmap_words(n) words
int upload_bits(commet_ctx *c, uint64_t *d_bits, const uint8_t *h_bits, uint64_t n)
{
    HIP_OK(hipMemsetAsync(d_bits, 0, bitmap_words(n) * 8, c->stream));
    HIP_OK(hipMemcpyAsync(d_bits, h_bits, bitmap_bytes_host(n), hipMemcpyHostToDevice, c->stream));
    return 0;
}

int launch_index_atomic(commet_ctx *c, const commet_readset *rs, uint64_t first, uint64_t count, const uint64_t *d_sel,
                        unsigned long long *d_fed)
{
    if (count == 0) return 0;
    const uint64_t blocks = (count + 255) / 256;
    if (blocks >= (1ull << 24)) return fail("index launch too large (>= 2^32 reads in one chunk)");
    KScope ks(c, "index_kernel", c->stream);
    if (c->k <= 32)
        COMMET_LAUNCH(index_kernel<uint32_t>, dim3((unsigned) blocks), dim3(256), 0, c->stream, rs->view(), c->view(),
                           c->k, first, count, d_sel, d_fed);
    else
        COMMET_LAUNCH(index_kernel<uint64_t>, dim3((unsigned) blocks), dim3(256), 0, c->stream, rs->view(), c->view(),
                           c->k, first, count, d_sel, d_fed);
    HIP_OK(hipGetLastError());
    return 0;
}

bool partition_eligible(const commet_ctx *c, const commet_readset *rs)
{
    return c->k >= 20 && c->k <= 34 && (uint64_t) rs->max_kcnt * 4 <= S1_KEYS &&
           ((uint64_t) rs->max_len + 7) / 8 <= S1_ITEMS;
}

// Bucketed construction of the filter for one chunk (index_part.hpp).  The
// filter must have been zeroed on the stream before.  kmers = exact number of
// complete k-mers of the selected reads of [first, first+count).
// d_ids != nullptr (fixed-length sets only): the chunk's selected reads are ids[pos_first .. pos_first + pos_count) (sel_ids_kernel);
// hist and scatter1 then take the arithmetic item path over that list instead of planning rounds over the bitmap
int launch_index_partitioned(commet_ctx *c, const commet_readset *rs, uint64_t first, uint64_t count, const uint64_t *d_sel,
                             uint64_t kmers, bool additive, bool zero_fill, int lane = 0, const uint32_t *d_ids = nullptr,
                             uint64_t pos_first = 0, uint64_t pos_count = 0)
{
    if (count == 0 || kmers == 0) return 0;
    if (d_ids && rs->uniform_len != 0 && !c->part_no_uni && pos_count) first = pos_first, count = pos_count, d_sel = nullptr;
    else d_ids = nullptr;
    commet_ctx::PartWs &ws = c->part[lane];
    hipStream_t stream = lane ? c->aux_stream : c->stream;
    if (lane == 1 && c->stagger_armed) {   // (see behind scatter1 below)
        HIP_OK(hipStreamWaitEvent(stream, c->ev_stagger, 0));
        c->stagger_armed = false;
    }
    uint32_t *const slot = c->slot_ptr(c->cur_slot);
    PartGeom g = make_geom(c->k);
    g.xcd_swizzle = c->s2_swizzle;
    g.packed = c->part_packed;
    if (c->part_b1 > 0 && c->part_b1 < g.nb_bits && c->part_b1 <= 8 && g.nb_bits - c->part_b1 <= 9) {
        g.b1 = c->part_b1;
        g.b2 = g.nb_bits - g.b1;
        g.nb1 = 1u << g.b1;
    }
    if (g.b2 == 0) g.packed = 0;   // single level: scatter1 writes the final buckets itself, as plain keys
    const uint64_t total = 4 * kmers;
    if (ws.nb != g.nb) {
        (void) dm_free(ws.hist); (void) dm_free(ws.wl); (void) dm_free(ws.off); (void) dm_free(ws.goff);
        (void) dm_free(ws.cur2);
        ws.hist = ws.wl = nullptr; ws.off = ws.goff = nullptr; ws.cur2 = nullptr;
        HIP_OK(dev_alloc(c, (void **) &ws.hist, (g.nb + 1) * sizeof(uint32_t), true));
        HIP_OK(dev_alloc(c, (void **) &ws.wl, (g.nb + 1) * sizeof(uint32_t), true));
        HIP_OK(dev_alloc(c, (void **) &ws.off, (g.nb + 1) * sizeof(uint64_t), true));
        HIP_OK(dev_alloc(c, (void **) &ws.goff, (g.nb + 1) * sizeof(uint64_t), true));
        if (!ws.blockcnt) HIP_OK(dev_alloc(c, (void **) &ws.blockcnt, (size_t) S1_GRID_MAX * MAX_L1 * sizeof(uint32_t), true));
        if (!ws.blockoff) HIP_OK(dev_alloc(c, (void **) &ws.blockoff, (size_t) S1_GRID_MAX * MAX_L1 * sizeof(unsigned long long), true));
        HIP_OK(dev_alloc(c, (void **) &ws.cur2, g.nb * sizeof(unsigned long long), true));
        ws.nb = g.nb;
    }
    // bufA: one word per key.  bufB: the same when the final level holds plain keys; in the packed geometry 8-byte groups of three keys,
    // every (scatter2 slab, final bucket) run rounded up to a whole group (part_scan_kernel: a bucket of c keys gets c / 3 + the slabs
    // that overlap its coarse bucket + 1 groups): two words x (total / 3 + 2^b2 x (total / S2_KEYS + 3 x 2^b1)) — 0.73 words per key
    // at k = 32, where a full word per key was 2.3 GB per workspace asked from the driver for nothing
    const bool packed_b = g.packed && g.b2 && (1u << g.b2) <= S2P_MAX_SUB;
    auto words_b = [&](uint64_t keys) -> uint64_t {
        return packed_b ? 2 * (keys / 3 + ((uint64_t) 1 << g.b2) * (keys / S2_KEYS + 3ull * g.nb1)) + 4096 : keys;
    };
    if (ws.cap_keys < total || ws.cap_b < words_b(total)) {
        HIP_OK(hipStreamSynchronize(stream));
        (void) dm_free(ws.bufA); (void) dm_free(ws.bufB);
        ws.bufA = ws.bufB = nullptr;
        ws.cap_keys = 0, ws.cap_b = 0;
        const uint64_t cap = total + total / 16 + (1ull << 20);
        // (touched there, too: a first touch inside the first scatter launch cost 15.6 instead of 2.5 ms)
        HIP_OK(alloc_workspace(c, (void **) &ws.bufA, cap * sizeof(uint32_t), stream));
        HIP_OK(alloc_workspace(c, (void **) &ws.bufB, words_b(cap) * sizeof(uint32_t), stream));
        ws.cap_keys = cap, ws.cap_b = words_b(cap);
    }
    const bool wide = c->k > 32;
    // every read of one length and no selection bitmap: items by arithmetic, no round planning (index_part.hpp, UNI)
    const bool uni = rs->uniform_len != 0 && d_sel == nullptr && !c->part_no_uni;   // (d_ids: positions in the list of selected reads)
    // ragged reads (with or without a selection bitmap): the chunk's items written out once, then walked like a fixed-length set
    // (index_part.hpp, LIST) — where an item's triple, counted from the chunk's first read, fits its 28 bits
    const uint64_t span_bound = std::min<uint64_t>((rs->n_bases >> 5) + rs->n_reads + 1, count * (((uint64_t) rs->max_len >> 5) + 2));
    bool list = !uni && rs->uniform_len == 0 && !c->part_no_uni && c->part_list != 1 && span_bound < ITEM_MAX_TRIPLES;
    const uint32_t *d_items = nullptr, *d_nitems = nullptr;
    if (list) {
        const uint64_t nblk = (count + ITEMS_BLOCK - 1) / ITEMS_BLOCK;
        // items of the chunk at most: every read of it as long as the set's longest / the set's bases in octets plus one per read
        const uint64_t sel_reads = pos_count ? pos_count : count;
        const uint64_t need = std::min<uint64_t>(sel_reads * std::max<uint64_t>(1, ((uint64_t) rs->max_len + 7) / 8), rs->n_bases / 8 + rs->n_reads) + 1;
        if (nblk + 1 >= (1ull << 24) || need >= (1ull << 32)) list = false;
        if (list && (ws.items_cap < need || ws.itemblk_cap < nblk + 1)) {
            HIP_OK(hipStreamSynchronize(stream));
            (void) dm_free(ws.items), (void) dm_free(ws.itemblk);
            ws.items = ws.itemblk = nullptr, ws.items_cap = ws.itemblk_cap = 0;
            const uint64_t cap = need + need / 8;
            ws.items_set = 0;
            if (dev_alloc(c, (void **) &ws.items, cap * sizeof(uint32_t), true) != hipSuccess ||
                dev_alloc(c, (void **) &ws.itemblk, (nblk + 1 + 1024) * sizeof(uint32_t), true) != hipSuccess) {
                (void) hipGetLastError();               // no room: the round planner walks the reads, as before
                (void) dm_free(ws.items), (void) dm_free(ws.itemblk);
                ws.items = ws.itemblk = nullptr;
                list = false;
            } else {
                ws.items_cap = cap, ws.itemblk_cap = nblk + 1 + 1024;
            }
        }
        // the list of an UNSELECTED read range is a function of the set alone: a workspace that still holds it (the same chunk of the same
        // set indexed again: the reference set of a rank's J1 calls, a benchmark's steady state) does not write it again
        const bool held = list && !d_sel && ws.items_set == rs->uid && ws.items_first == first && ws.items_count == count && ws.items_nblk == (uint32_t) nblk;
        if (held) {
            d_items = ws.items, d_nitems = ws.itemblk + nblk;
        } else if (list) {
            ws.items_set = d_sel ? 0 : rs->uid, ws.items_first = first, ws.items_count = count, ws.items_nblk = (uint32_t) nblk;
            KScope ks(c, "part_items_kernels", stream);
            COMMET_LAUNCH(part_items_kernel<false>, dim3((unsigned) nblk), dim3(ITEMS_BLOCK), 0, stream, rs->view(), rs->d_kcnt, d_sel, first, count, c->k,
                          ws.itemblk, ws.items);
            COMMET_LAUNCH(sel_scan_kernel, dim3(1), dim3(1024), 0, stream, ws.itemblk, (uint32_t) nblk);
            COMMET_LAUNCH(part_items_kernel<true>, dim3((unsigned) nblk), dim3(ITEMS_BLOCK), 0, stream, rs->view(), rs->d_kcnt, d_sel, first, count, c->k,
                          ws.itemblk, ws.items);
            HIP_OK(hipGetLastError());
            d_items = ws.items, d_nitems = ws.itemblk + nblk;
        }
    }
    const int mode = uni ? 1 : list ? 2 : 0;
    // TIMING BOUND ONLY (COMMET_HIST_REUSE=1, never set by the product): a workspace that counted this very chunk last time keeps its
    // histogram — what the index costs without the counting pass, measured on a benchmark's steady state
    static const bool hist_reuse_on = getenv("COMMET_HIST_REUSE") && atoi(getenv("COMMET_HIST_REUSE")) != 0;
    const bool hist_held = hist_reuse_on && !d_sel && !d_ids && ws.hist_set == rs->uid && ws.hist_first == first && ws.hist_count == count;
    ws.hist_set = (d_sel || d_ids) ? 0 : rs->uid, ws.hist_first = first, ws.hist_count = count;
    if (!hist_held) HIP_OK(hipMemsetAsync(ws.hist, 0, (g.nb + 1) * sizeof(uint32_t), stream));
    // scatter1's grid fixes how the read range is cut; hist counts with the same cut, two ranges per workgroup
    const uint32_t grid1 = (uint32_t) std::min<uint64_t>(S1_GRID_MAX, (count + 63) / 64);
    {
        const unsigned grid = (grid1 + 1) / 2;
        const bool full = g.nb <= HIST_MAX_BUCKETS;
        // 32-bit keys (k <= 32): at most 2^15 buckets, the LDS histogram always covers them all (FULL); 64-bit keys: never.
        // Only those six instantiations exist (tests/test_gpu_zz_dispatch_coverage.py checks that each is reached).
        if (full == wide) return fail("internal error: histogram geometry (k = %d, %u buckets)", c->k, g.nb);
        const void *fn = wide ? (mode == 1 ? (const void *) part_hist_kernel<uint64_t, 1, false> : mode == 2 ? (const void *) part_hist_kernel<uint64_t, 2, false>
                                                                                                             : (const void *) part_hist_kernel<uint64_t, 0, false>)
                              : (mode == 1 ? (const void *) part_hist_kernel<uint32_t, 1, true> : mode == 2 ? (const void *) part_hist_kernel<uint32_t, 2, true>
                                                                                                            : (const void *) part_hist_kernel<uint32_t, 0, true>);
        for (uint32_t b_lo = 0; b_lo < g.nb && !hist_held; b_lo += HIST_MAX_BUCKETS) {
            const uint32_t n_b = std::min<uint32_t>(HIST_MAX_BUCKETS, g.nb - b_lo);
            const size_t lds = ((size_t) n_b + 2 * HIST_NT + 24) * 4 + (size_t) HIST_NT * 8;
            HIP_OK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
            ReadsView rv = rs->view();
            const uint32_t *kc = rs->d_kcnt;
            uint32_t *hist = ws.hist, *bcnt = ws.blockcnt;
            uint32_t nblk = grid1;
            void *args[] = {&rv, &kc, &d_sel, &first, &count, &g, &b_lo, (void *) &n_b, &hist, &nblk, &bcnt, &d_ids, &d_items, &d_nitems};
            KScope ks(c, "part_hist_kernel", stream);
            note_launch(fn);
            HIP_OK(hipLaunchKernel(fn, dim3(grid), dim3(HIST_NT), args, lds, stream));
        }
    }
    {
        const bool lds_hist = (size_t) g.nb * 4 <= (128u << 10);   // stage the histogram in LDS (coalesced loads) when it fits
        const size_t lds = lds_hist ? ((size_t) g.nb + g.nb / 32 + 1) * 4 : 0;   // (padded: see the kernel)
        if (lds) HIP_OK(hipFuncSetAttribute((const void *) part_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        KScope ks(c, "part_scan_kernel", stream);
        COMMET_LAUNCH(part_scan_kernel, dim3(1), dim3(1024), lds, stream, ws.hist, g, zero_fill ? 1 : 0, ws.off,
                           ws.cur2, ws.wl, ws.goff, lds_hist ? 1 : 0);
    }
    HIP_OK(hipGetLastError());
    {
        KScope ks(c, "part_blockoff_kernel", stream);
        COMMET_LAUNCH(part_blockoff_kernel, dim3(g.nb1), dim3(512), 0, stream, ws.blockcnt, ws.off, g, grid1,
                           ws.blockoff);
    }
    HIP_OK(hipGetLastError());
    // scatter 1 (straight into the final buckets when there is a single level)
    uint32_t *level1_out = g.b2 ? ws.bufA : ws.bufB;
    {
        const void *fn = wide ? (mode == 1 ? (const void *) part_scatter1_kernel<uint64_t, 1> : mode == 2 ? (const void *) part_scatter1_kernel<uint64_t, 2>
                                                                                                          : (const void *) part_scatter1_kernel<uint64_t, 0>)
                              : (mode == 1 ? (const void *) part_scatter1_kernel<uint32_t, 1> : mode == 2 ? (const void *) part_scatter1_kernel<uint32_t, 2>
                                                                                                          : (const void *) part_scatter1_kernel<uint32_t, 0>);
        ReadsView rv = rs->view();
        const uint32_t *kc = rs->d_kcnt;
        const unsigned long long *boff = ws.blockoff;
        void *args[] = {&rv, &kc, &d_sel, &first, &count, &g, &boff, &level1_out, &d_ids, &d_items, &d_nitems};
        KScope ks(c, "part_scatter1_kernel", stream);
        note_launch(fn);
        HIP_OK(hipLaunchKernel(fn, dim3(grid1), dim3(S1_NT), args, 0, stream));
    }
    // Two lanes, staggered: the chunk on the second lane starts when this one's scatter1 is through, so that its hist and
    // scatter1 (VALU, LDS atomics, a write pattern) run beside this chunk's scatter2 and build (HBM streams) instead of beside
    // the same phases of this chunk.  Four boxes, configs[1], per step: -0.36, -0.25, 0.0 and +0.04 ms (the slower the box's
    // draw, the more); a 2 x 50 M-read pair 169.4 -> 168.1 ms.  Staggering behind hist instead: +0.5 ms.
    if (c->lane_stagger && lane == 0 && c->ev_stagger) {
        HIP_OK(hipEventRecord(c->ev_stagger, stream));
        c->stagger_armed = true;
    }
    if (COMMET_ABLATE & 31) return 0;   // ablation builds only: scatter1 left garbage in bufA, nothing downstream may consume it
    if (g.b2) {
        const uint64_t grid = (total + S2_KEYS - 1) / S2_KEYS;
        if (grid >= (1ull << 24)) return fail("scatter launch too large");
        {
            KScope ks(c, (g.packed && (1u << g.b2) <= S2P_MAX_SUB) ? "part_scatter2_packed_kernel" : "part_scatter2_kernel", stream);
            if (g.packed && (1u << g.b2) <= S2P_MAX_SUB)
                COMMET_LAUNCH(part_scatter2_packed_kernel, dim3((unsigned) grid), dim3(S2_NT), 0, stream, ws.bufA, (uint2 *) ws.bufB,
                                   ws.off, g, ws.cur2, total);
            else
                COMMET_LAUNCH(part_scatter2_kernel, dim3((unsigned) grid), dim3(S2_NT), 0, stream, ws.bufA, ws.bufB,
                                   ws.off, g, ws.cur2, total);
        }
        HIP_OK(hipGetLastError());
    }
    if (COMMET_ABLATE) return 0;   // ablation builds only: bufB holds garbage
    {
        const uint64_t grid = (uint64_t) g.nb + total / BUILD_CAP + 1;
        if (grid >= (1ull << 24)) return fail("build launch too large");
        if (zero_fill) {   // no memset happened: clear the tiles that several workgroups OR into
            KScope ks(c, "part_zero_split_kernel", stream);
            COMMET_LAUNCH(part_zero_split_kernel, dim3(g.nb), dim3(256), 0, stream, ws.wl, g, slot);
            HIP_OK(hipGetLastError());
        }
        HIP_OK(hipFuncSetAttribute((const void *) part_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int) (TILE_WORDS * sizeof(uint32_t))));
        {
            KScope ks(c, "part_build_kernel", stream);
            COMMET_LAUNCH(part_build_kernel, dim3((unsigned) grid), dim3(BUILD_NT), TILE_WORDS * sizeof(uint32_t), stream,
                               ws.bufB, g.packed ? ws.goff : ws.off, ws.wl, g, slot, additive ? 1 : 0, ws.cur2);
        }
        HIP_OK(hipGetLastError());
    }
    return 0;
}

// whether a launch of `kmers` complete k-mers takes the bucketed construction
bool would_partition(const commet_ctx *c, const commet_readset *rs, uint64_t kmers)
{
    if (kmers == ~0ull || kmers == 0) return false;
    if (c->index_mode == 2) return partition_eligible(c, rs);
    if (c->index_mode == 0) return partition_eligible(c, rs) && kmers >= c->part_min_kmers;
    return false;
}

// kmers: exact complete-k-mer count of the launch when known (enables the bucketed path), else ~0.
// fresh_filter: the filter holds nothing yet; filter_zeroed: the caller has zeroed it (if not, a bucketed build
// zero-fills what it does not set; the atomic kernel always needs a zeroed filter).
int launch_index(commet_ctx *c, const commet_readset *rs, uint64_t first, uint64_t count, const uint64_t *d_sel,
                 unsigned long long *d_fed, uint64_t kmers = ~0ull, bool fresh_filter = false, bool filter_zeroed = true,
                 int lane = 0, const uint32_t *d_ids = nullptr, uint64_t pos_first = 0, uint64_t pos_count = 0)
{
    if (c->index_mode == 2) {
        if (!partition_eligible(c, rs)) return fail("bucketed index construction needs 20 <= k <= 34 and reads of at most %u k-mers", S1_KEYS / 4);
        if (kmers == ~0ull) return fail("bucketed index construction needs the k-mer count of the launch");
    }
    if (!would_partition(c, rs, kmers)) {
        if (!filter_zeroed) return fail("internal error: atomic index launch on a filter that was not zeroed");
        return launch_index_atomic(c, rs, first, count, d_sel, d_fed);
    }
    if (d_fed) {
        // the count is known exactly on the host
        const unsigned long long v = kmers;
        HIP_OK(hipMemcpyAsync(d_fed, &v, sizeof v, hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    }
    return launch_index_partitioned(c, rs, first, count, d_sel, kmers, !fresh_filter, fresh_filter && !filter_zeroed, lane, d_ids, pos_first,
                                    pos_count);
}

}  // namespace
