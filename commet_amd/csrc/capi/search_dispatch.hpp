// capi/search_dispatch.hpp — which search regime runs for a set and a group of chunk filters (plain, group, group8, tiled, bit-sliced, wide rows) and its launches
// (a part of the one translation unit capi.hip: included there, in order, after the kernels and state.hpp)
#pragma once

namespace {

// min_hits as the kernels get it: a read of max_len bases holds at most max_len / k non-overlapping k-mers, so every
// t above max_len / k + 1 behaves like that value (never found, same probes); clamping keeps (t - seen - 1) * k and
// last - (t - 1) * k inside 32-bit int whatever atoi handed to commet_create
inline int t_eff(const commet_ctx *c, const commet_readset *rs)
{
    return (int) std::min<uint64_t>((uint64_t) c->t, (uint64_t) rs->max_len / (uint64_t) c->k + 1);
}

// al.ids != nullptr: a pass over the n_launch (at most) listed reads (kernels.hpp, ActiveList) instead of the whole set
int launch_search(commet_ctx *c, const commet_readset *rs, const uint64_t *d_sel, uint64_t *d_tags, uint64_t *d_found,
                  unsigned long long *d_counters, unsigned long long *d_probes = nullptr, ActiveList al = ActiveList{nullptr, nullptr},
                  uint64_t n_launch = 0)
{
    if (rs->n_reads == 0) return 0;
    if (al.ids && n_launch == 0) return 0;
    const uint64_t blocks = ((al.ids ? n_launch : rs->n_reads) + 255) / 256;
    if (blocks >= (1ull << 24)) return fail("search launch too large (>= 2^32 reads in one set)");
    const dim3 g((unsigned) blocks), b(256);
    const bool cnt = d_probes != nullptr;
    KScope ks(c, "search_kernel", c->stream);
    if (c->k <= 32) {
        if (cnt)
            COMMET_LAUNCH((search_kernel<uint32_t, true>), g, b, 0, c->stream, rs->view(), c->view(), c->k, t_eff(c, rs), d_sel,
                               d_tags, d_found, d_counters, d_probes, al);
        else
            COMMET_LAUNCH((search_kernel<uint32_t, false>), g, b, 0, c->stream, rs->view(), c->view(), c->k, t_eff(c, rs), d_sel,
                               d_tags, d_found, d_counters, d_probes, al);
    } else {
        if (cnt)
            COMMET_LAUNCH((search_kernel<uint64_t, true>), g, b, 0, c->stream, rs->view(), c->view(), c->k, t_eff(c, rs), d_sel,
                               d_tags, d_found, d_counters, d_probes, al);
        else
            COMMET_LAUNCH((search_kernel<uint64_t, false>), g, b, 0, c->stream, rs->view(), c->view(), c->k, t_eff(c, rs), d_sel,
                               d_tags, d_found, d_counters, d_probes, al);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

// makes `g` filter slots (+ the interleaved A planes with stride gs) available; slot contents are undefined after a grow
int ensure_slots(commet_ctx *c, int g, int gs)
{
    if (c->n_slots < g) {
        HIP_OK(hipStreamSynchronize(c->stream));
        uint32_t *nf = nullptr;
        // a whole group's worth at once (2, 4 or 8 slots): a job of seven chunks followed by one of eight (the N x N driver's J1, then its
        // shared J2 passes) used to take 14 GiB and then 16 GiB from the driver at k = 32; if that does not fit, what is needed now
        int want = std::max(g, gs);
        hipError_t e = dev_alloc(c, (void **) &nf, (size_t) want * c->filter_bytes, true);
        if (e != hipSuccess && want > g) {
            (void) hipGetLastError();
            want = g;
            e = dev_alloc(c, (void **) &nf, (size_t) want * c->filter_bytes, true);
        }
        if (e != hipSuccess) return fail("cannot allocate %d filter slots: %s", g, hipGetErrorString(e));
        (void) dm_free(c->filter);
        c->filter = nf;
        c->n_slots = want;
    }
    if (c->il_stride < gs) {
        HIP_OK(hipStreamSynchronize(c->stream));
        (void) dm_free(c->il_a);
        c->il_a = nullptr;
        c->il_stride = 0;
        HIP_OK(dev_alloc(c, (void **) &c->il_a, (size_t) gs * c->plane_words * sizeof(uint32_t), true));
        c->il_stride = gs;
    }
    return 0;
}

int launch_interleave(commet_ctx *c, int g, int gs)
{
    const uint64_t blocks = std::min<uint64_t>((c->plane_words + 255) / 256, 1u << 16);
    KScope ks(c, "interleave_a_kernel", c->stream);
    if (gs == 2)
        COMMET_LAUNCH(interleave_a_kernel<2>, dim3((unsigned) blocks), dim3(256), 0, c->stream, c->filter, 4 * c->plane_words,
                           c->plane_words, g, c->il_a);
    else if (gs == 4)
        COMMET_LAUNCH(interleave_a_kernel<4>, dim3((unsigned) blocks), dim3(256), 0, c->stream, c->filter, 4 * c->plane_words,
                           c->plane_words, g, c->il_a);
    else
        COMMET_LAUNCH(interleave_a_kernel<8>, dim3((unsigned) blocks), dim3(256), 0, c->stream, c->filter, 4 * c->plane_words,
                           c->plane_words, g, c->il_a);
    HIP_OK(hipGetLastError());
    return 0;
}

template <typename W, int GS>
int launch_search_group_t(commet_ctx *c, const commet_readset *rs, const FilterGroupView &fg, uint32_t nw_max, const uint64_t *d_sel,
                          uint64_t *d_tags, unsigned long long *d_counters, uint32_t cstride, unsigned long long *d_probes, ActiveList al,
                          uint64_t n_launch)
{
    const dim3 g((unsigned) (((al.ids ? n_launch : rs->n_reads) + 255) / 256)), b(256);
    size_t lds = (size_t) fg.g * 2 * nw_max * 256 * sizeof(uint32_t);
    // the lanes' reads staged in LDS too (3 * nw_max words each) when that still fits 64 KiB
    uint32_t rw_nw = 0;
    if (!d_probes && nw_max <= 8 && lds + (size_t) 3 * nw_max * 256 * sizeof(uint32_t) <= (64u << 10) && c->stage_reads) {
        rw_nw = nw_max;
        lds += (size_t) 3 * nw_max * 256 * sizeof(uint32_t);
    }
    KScope ks(c, "search_group_kernel", c->stream);
    if (d_probes) {
        HIP_OK(hipFuncSetAttribute((const void *) search_group_kernel<W, GS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        COMMET_LAUNCH((search_group_kernel<W, GS, true>), g, b, lds, c->stream, rs->view(), fg, c->k, t_eff(c, rs), nw_max, d_sel, d_tags,
                           d_counters, cstride, d_probes, rw_nw, al);
    } else {
        HIP_OK(hipFuncSetAttribute((const void *) search_group_kernel<W, GS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        COMMET_LAUNCH((search_group_kernel<W, GS, false>), g, b, lds, c->stream, rs->view(), fg, c->k, t_eff(c, rs), nw_max, d_sel, d_tags,
                           d_counters, cstride, d_probes, rw_nw, al);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

// mask words (32 first-hit windows each) a read of the set needs per strand and filter in the register-mask kernels
// (search_group8_kernel, tq_replay_kernel): 2, 3, 4, 6 or 8 — reads of up to 318 bases at k = 32, t = 2 (round 6; 96 windows before)
constexpr int MASK_MAX_WIN = 255;      // (= TQ_MAX_WIN)
inline int mask_words(const commet_ctx *c, const commet_readset *rs)
{
    const int64_t fhw = (int64_t) rs->max_len - (int64_t) t_eff(c, rs) * c->k + 1;
    return fhw <= 64 ? 2 : fhw <= 96 ? 3 : fhw <= 128 ? 4 : fhw <= 192 ? 6 : 8;
}

// one pass of rs over the `g` chunk filters in slots 0..g-1 (A planes already interleaved with stride gs)
// job_mask != 0 (gs == 8 only): the g filters belong to several jobs, bit i = filter i opens one; job j's tags at d_tags + j * job_tag_words
int launch_search_group(commet_ctx *c, const commet_readset *rs, int g, int gs, const uint64_t *d_sel, uint64_t *d_tags,
                        unsigned long long *d_counters, uint32_t cstride, unsigned long long *d_probes, ActiveList al = ActiveList{nullptr, nullptr},
                        uint64_t n_launch = 0, uint32_t job_mask = 0, uint64_t job_tag_words = 0)
{
    if (rs->n_reads == 0) return 0;
    if (al.ids && n_launch == 0) return 0;
    if ((rs->n_reads + 255) / 256 >= (1ull << 24)) return fail("search launch too large (>= 2^32 reads in one set)");
    FilterGroupView fg;
    fg.il_a = c->il_a;
    fg.slot0 = c->filter;
    fg.slot_words = 4 * c->plane_words;
    fg.plane_words = c->plane_words;
    fg.g = g;
    if (gs == 8) {   // register masks, no LDS (group8_ok)
        const int mw_set = mask_words(c, rs);   // mask words per strand and filter: 2, 3, 4, 6 or 8
        auto launch = [&](int mw, ActiveList l, uint64_t n_l) {
            KScope ks(c, "search_group8_kernel", c->stream);
            const dim3 grid((unsigned) (((l.ids ? n_l : rs->n_reads) + 255) / 256)), block(256);
#define COMMET_G8(W, MW) COMMET_LAUNCH((search_group8_kernel<W, MW>), grid, block, 0, c->stream, rs->view(), fg, c->k, t_eff(c, rs), d_sel, d_tags, d_counters, cstride, l, job_mask, job_tag_words)
            if (c->k <= 32) {
                if (mw == 2) COMMET_G8(uint32_t, 2);
                else if (mw == 3) COMMET_G8(uint32_t, 3);
                else if (mw == 4) COMMET_G8(uint32_t, 4);
                else if (mw == 6) COMMET_G8(uint32_t, 6);
                else COMMET_G8(uint32_t, 8);
            } else {
                if (mw == 2) COMMET_G8(uint64_t, 2);
                else if (mw == 3) COMMET_G8(uint64_t, 3);
                else if (mw == 4) COMMET_G8(uint64_t, 4);
                else if (mw == 6) COMMET_G8(uint64_t, 6);
                else COMMET_G8(uint64_t, 8);
            }
#undef COMMET_G8
        };
        if (al.ids && al.ids == rs->d_len_order && rs->n_len_seg > 1 && !c->mask_split) {
            // the set's reads in order of their window counts: every segment of the list with the narrowest masks its reads fit
            // (three mask words cost 135 VGPRs and a fifth of the request rate of two; a 50-150-bp set has half its windows in
            // reads that need two)
            for (int sg = 0; sg < rs->n_len_seg; ++sg) {
                const commet_readset::LenSeg &seg = rs->len_seg[sg];
                launch(std::min(seg.mw, mw_set), ActiveList{rs->d_len_order + seg.start, rs->d_len_order + rs->n_reads + 1 + sg}, seg.count);
            }
        } else {
            launch(mw_set, al, n_launch);
        }
        HIP_OK(hipGetLastError());
        return 0;
    }
    const uint32_t nw_max = (rs->max_len + 31) / 32;
    if (c->k <= 32) {
        if (gs == 2) return launch_search_group_t<uint32_t, 2>(c, rs, fg, nw_max, d_sel, d_tags, d_counters, cstride, d_probes, al, n_launch);
        return launch_search_group_t<uint32_t, 4>(c, rs, fg, nw_max, d_sel, d_tags, d_counters, cstride, d_probes, al, n_launch);
    }
    if (gs == 2) return launch_search_group_t<uint64_t, 2>(c, rs, fg, nw_max, d_sel, d_tags, d_counters, cstride, d_probes, al, n_launch);
    return launch_search_group_t<uint64_t, 4>(c, rs, fg, nw_max, d_sel, d_tags, d_counters, cstride, d_probes, al, n_launch);
}

bool group_searchable(const commet_ctx *c, const commet_readset *rs, int g)
{
    // LDS masks: g chunks x 2 strands x ceil(max_len/32) words per lane, at most 64 KiB per workgroup
    const uint64_t nw = ((uint64_t) rs->max_len + 31) / 32;
    return c->k >= 2 && nw >= 1 && (uint64_t) g * 2 * nw * 256 * 4 <= (64u << 10);
}

// groups of 5..8 chunk filters: search_group8_kernel keeps the gathered bits of at most 255 first-hit windows per read in
// registers (kernels.hpp: two to eight mask words per strand and filter); the probe-counting builds exist for groups of <= 4 only
bool group8_ok(const commet_ctx *c, const commet_readset *rs)
{
    const int64_t first_hit_windows = (int64_t) rs->max_len - (int64_t) t_eff(c, rs) * c->k + 1;
    return c->k >= 2 && !c->count_probes && first_hit_windows <= MASK_MAX_WIN;
}

// ---- sparse passes: the reads of a pass as a list (kernels.hpp, ActiveList) --------------------------------------
// A pass that searches less than half of a set's reads — Commet.py's third job of a pair searches a set restricted to the first
// job's result (Commet.py:233), ~22 % of it — walks the list of those reads (sel & ~tags, in order) instead of the set: every lane
// of a wave then has a read, where the bitmap form kept 64 lanes waiting through a read's dependent round trips for the 14 that had
// one (a J3 job of configs[3]: search 46 -> 12 ms).  `selected` = the reads the host plan visits (tags can only shrink the list).
bool sparse_pass(const commet_ctx *c, const commet_readset *rs, const uint64_t *d_sel, uint64_t selected)
{
    if (!d_sel || c->sparse_search == 1 || c->count_probes || rs->n_reads >= (1ull << 32)) return false;
    if (c->sparse_search == 2) return true;
    return rs->n_reads >= 4096 && selected * 2 < rs->n_reads;
}

// builds the list on the job's stream (three small launches, no synchronisation); 0 = done, 1 = no room (the caller takes the bitmap form)
int build_active_list(commet_ctx *c, const commet_readset *rs, const uint64_t *d_sel, const uint64_t *d_tags, uint64_t selected, ActiveList *al)
{
    const uint64_t n_words = bitmap_words(rs->n_reads), nb = (n_words + IDS_BLOCK_WORDS - 1) / IDS_BLOCK_WORDS;
    if (c->act_cap < selected || c->actblk_cap < nb + 1) {
        if (hipStreamSynchronize(c->stream) != hipSuccess) return 1;
        (void) dm_free(c->d_act), (void) dm_free(c->d_actblk);
        c->d_act = c->d_actblk = nullptr, c->act_cap = c->actblk_cap = 0;
        const uint64_t cap = std::max<uint64_t>(selected, 1024);
        if (dev_alloc(c, (void **) &c->d_act, cap * sizeof(uint32_t), true) != hipSuccess ||
            dev_alloc(c, (void **) &c->d_actblk, (nb + 1) * sizeof(uint32_t), true) != hipSuccess) {
            (void) hipGetLastError();
            (void) dm_free(c->d_act), (void) dm_free(c->d_actblk);
            c->d_act = c->d_actblk = nullptr;
            return 1;
        }
        c->act_cap = cap, c->actblk_cap = nb + 1;
    }
    KScope ks(c, "active_list_kernels", c->stream);
    COMMET_LAUNCH(sel_count_kernel, dim3((unsigned) nb), dim3(64), 0, c->stream, d_sel, n_words, c->d_actblk, d_tags);
    COMMET_LAUNCH(sel_scan_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_actblk, (uint32_t) nb);
    COMMET_LAUNCH(sel_ids_kernel, dim3((unsigned) nb), dim3(64), 0, c->stream, d_sel, n_words, c->d_actblk, c->d_act, d_tags);
    if (hipGetLastError() != hipSuccess) return 1;
    al->ids = c->d_act;
    al->n = c->d_actblk + nb;
    return 0;
}

// ---- ragged sets in order of their window counts (tile_search.hpp, lo_*_kernel) -------------------------------------------
// al = the set's list when a pass qualifies: a ragged set visited whole (no selection, no tags yet: the first pass of a job), no other
// list in use.  The list is made on the job's stream on first use (three small launches + the scan) and kept with the set; no room =
// the natural order, as before.
bool ordered_pass(commet_ctx *c, const commet_readset *rs, const uint64_t *d_sel, bool tags_empty, ActiveList *al)
{
    if (c->ordered_scan == 1 || rs->uniform_len != 0 || d_sel || !tags_empty || al->ids || c->count_probes || rs->len_order_failed) return false;
    if (rs->n_reads == 0 || rs->n_reads >= (1ull << 32) - 1 || (c->ordered_scan == 0 && rs->n_reads < (1u << 16))) return false;
    if (!rs->d_len_order) {
        const uint64_t n_blocks = (rs->n_reads + LO_BLOCK - 1) / LO_BLOCK, entries = n_blocks * LO_CLASSES;
        const uint32_t nb = (uint32_t) ((entries + 4095) / 4096);
        uint32_t *ids = nullptr;
        hipError_t e = dm_malloc((void **) &ids, (rs->n_reads + 8) * sizeof(uint32_t));   // (+ 1: the list's length; + 5: its segments')
        if (e == hipSuccess && c->lo_cnt_cap < entries + 1 + nb + 2) {
            (void) dm_free(c->d_lo_cnt);
            c->d_lo_cnt = nullptr, c->lo_cnt_cap = 0;
            e = dm_malloc((void **) &c->d_lo_cnt, (entries + 1 + nb + 2) * sizeof(unsigned long long));
            if (e == hipSuccess) c->lo_cnt_cap = entries + 1 + nb + 2;
        }
        if (e != hipSuccess) {
            (void) hipGetLastError();
            (void) dm_free(ids);
            rs->len_order_failed = true;
            return false;
        }
        unsigned long long *cnt = c->d_lo_cnt, *totals = cnt + entries + 1;
        const uint32_t tk = (uint32_t) std::min<int64_t>((int64_t) t_eff(c, rs) * c->k, 0x7FFFFFFF);
        KScope ks(c, "len_order_kernels", c->stream);
        COMMET_LAUNCH(lo_count_kernel, dim3((unsigned) n_blocks), dim3(256), 0, c->stream, rs->view(), tk, n_blocks, cnt);
        COMMET_LAUNCH(tq_scan_blocks_kernel, dim3(nb), dim3(1024), 0, c->stream, cnt, entries, totals);
        COMMET_LAUNCH(tq_scan_totals_kernel, dim3(1), dim3(1024), 0, c->stream, totals, nb, totals + nb);
        COMMET_LAUNCH(tq_scan_add_kernel, dim3(nb), dim3(1024), 0, c->stream, cnt, entries, totals, totals + nb);
        COMMET_LAUNCH(lo_fill_kernel, dim3((unsigned) n_blocks), dim3(256), 0, c->stream, rs->view(), tk, n_blocks, cnt, ids);
        if (hipGetLastError() != hipSuccess) {
            (void) hipStreamSynchronize(c->stream);
            (void) dm_free(ids);
            rs->len_order_failed = true;
            return false;
        }
        // where the classes of 64, 96, 128 and 192 windows and more end (class = 31 - windows / 8: tile_search.hpp, lo_class; after the
        // scan cnt[class * n_blocks] = the class's first place in the list): the list's segments by mask width.  One read-back per
        // set; a segment of few reads joins its wider neighbour
        rs->n_len_seg = 0;
        unsigned long long first[4] = {0, 0, 0, 0};
        static const uint32_t cls[4] = {8, 16, 20, 24};
        static const int width[5] = {8, 6, 4, 3, 2};
        bool ok = true;
        for (int i = 0; i < 4 && ok; ++i)
            ok = hipMemcpyAsync(&first[i], cnt + (uint64_t) cls[i] * n_blocks, sizeof first[i], hipMemcpyDeviceToHost, c->stream) == hipSuccess;
        ok = ok && hipStreamSynchronize(c->stream) == hipSuccess;
        if (ok) {
            const uint64_t bound[6] = {0, first[0], first[1], first[2], first[3], rs->n_reads};
            const uint64_t few = c->ordered_scan == 2 ? 1 : 4096;
            for (int i = 0; i < 5; ++i) {
                const uint64_t a = bound[i], b = bound[i + 1];
                if (b <= a || b > rs->n_reads) continue;
                if (rs->n_len_seg > 0 && b - a < few)
                    rs->len_seg[rs->n_len_seg - 1].count += (uint32_t) (b - a);
                else
                    rs->len_seg[rs->n_len_seg++] = commet_readset::LenSeg{(uint32_t) a, (uint32_t) (b - a), width[i]};
            }
            uint32_t counts[5] = {0, 0, 0, 0, 0};
            for (int i = 0; i < rs->n_len_seg; ++i) counts[i] = rs->len_seg[i].count;
            ok = hipMemcpyAsync(ids + rs->n_reads + 1, counts, sizeof counts, hipMemcpyHostToDevice, c->stream) == hipSuccess &&
                 hipStreamSynchronize(c->stream) == hipSuccess;       // (counts[] is on this stack)
        }
        if (!ok) {
            (void) hipGetLastError();
            rs->n_len_seg = 0;                                         // (the list itself is good: one launch at the set's width)
        }
        rs->d_len_order = ids;
    }
    al->ids = rs->d_len_order;
    al->n = rs->d_len_order + rs->n_reads;
    return true;
}

// ---- tiled search (tile_search.hpp) ----------------------------------------------------------------------------
constexpr int TQ_SBITS = 24;          // slice = 2^24 bits of plane A's address space: 2 MiB per chunk filter, 4 MiB for a group of two
                                      // (measured on configs[1]: 22 / 23 / 24 -> probe 2.43 / 2.56 / 2.35 ms, gpurun_out/r02_tq_ab2.log)

constexpr int TQ_MAX_K = 34;          // 64-bit keys from k = 33 (the reference's default k, index_and_search.cpp:71): 2^(k - 24) <= 1024 slices

bool tiled_ok(const commet_ctx *c, const commet_readset *rs, int g)
{
    if (c->tiled_mode == 1 || c->count_probes || rs->ql.failed) return false;
    if (c->k <= TQ_SBITS || c->k > TQ_MAX_K || g < 1 || g > 2) return false;
    const int64_t first_hit_windows = (int64_t) rs->max_len - (int64_t) t_eff(c, rs) * c->k + 1;
    if (first_hit_windows < 1 || first_hit_windows > TQ_MAX_WIN) return false;
    if (rs->max_len >= TQ_MAX_LEN) return false;        // (the replay keeps a piece's read extents in 16 + 16 bits; such reads pass the line above only with t in the hundreds)
    if (rs->n_reads >= (1ull << 32)) return false;
    // (rs->fhw_total: the set's first-hit windows summed over its reads — n x first_hit_windows for reads of one length, less for ragged sets)
    if (rs->fhw_total >= (1ull << 32)) return false;   // record numbers are 32 bits (forced mode, too)
    if (c->tiled_mode == 2) return true;
    // auto: sets of a million reads or more whose query list (8 bytes per first-hit window) stays under 4 GiB.  Measured
    // against the fused kernels: configs[1] search 9.4 -> 8.6 ms, configs[2] jobs 1.92-1.96 -> 1.84 s.
    const uint64_t est = rs->fhw_total * 8;
    // (a list above the cap when its memory was set aside beforehand, commet_readset_reserve_cache: the allocation is then not on this job's path)
    const bool set_aside = rs->ql_reserved.load() && rs->ql_reserved_at.load() == g_devmem.trims.load();   // (no trim since: the blocks are still filed)
    if (rs->n_reads < (1ull << 20) || (est > c->ql_max_list && !set_aside)) return false;
    // A list of more than 4 GiB (sets of 15 M reads and up) is built for a set's SECOND such scan: a set that is scanned once —
    // every target of a rank that holds few pairs of a large matrix — would pay 12 ms of kernels and an 11 GB allocation for a
    // 7.6 ms gain (one J2 / J3 job of configs[3]: 54.7 against 62.3 ms), a set that is scanned again and again — every set of a
    // matrix on one GPU, the reference set of a rank's pairs — gets its list one scan late.
    if (est > (4ull << 30) && !rs->ql.built && rs->ql_wanted++ == 0) return false;
    return true;
}

// the set's query list for this context's (k, t): counted, scanned, filled; kept with the set
int build_query_list(commet_ctx *c, const commet_readset *rs)
{
    commet_readset::QueryList &ql = rs->ql;
    if (ql.built) {
        ql.last_use = ++c->ql_clock;
        return 0;
    }
    ql.sbits = TQ_SBITS;
    if (c->tq_sbits) ql.sbits = std::max(c->k - 10, std::min(c->k - 1, c->tq_sbits));   // A/B runs
    ql.n_slices = 1u << (c->k - ql.sbits);
    ql.n_pieces = (uint32_t) ((rs->n_reads + TQ_PIECE - 1) / TQ_PIECE);
    const uint64_t entries = (uint64_t) ql.n_slices * ql.n_pieces;
    const uint32_t nb = (uint32_t) ((entries + 4095) / 4096);
    // The list is built on a stream of its own: at this point the job's index kernels are queued on the main streams, and the
    // list's kernels (VALU and LDS work on the SEARCH set) run beside them instead of behind them; the main stream waits for the
    // list before the probe (ev_list).  Nothing here may synchronise the device: the scan's block totals live in a scratch
    // buffer kept with the context (a hipFree would wait for the index build).
    hipStream_t ls = c->list_stream;
    unsigned long long *d_totals = nullptr;
    // (the caller holds ql_mu: on an allocation failure here the other sets' lists are given back directly)
    auto alloc = [&](void **ptr, size_t bytes) -> hipError_t {
        hipError_t ae = dm_malloc(ptr, bytes);
        if (ae != hipErrorOutOfMemory) return ae;
        (void) hipGetLastError();
        if (!shrink_query_lists(c, 0, false)) return ae;
        ae = dm_malloc(ptr, bytes);
        if (ae == hipErrorOutOfMemory) (void) hipGetLastError();
        return ae;
    };
    hipError_t e = alloc((void **) &ql.d_tile_off, (entries + 1) * sizeof(unsigned long long));
    if (e == hipSuccess && c->ql_totals_cap < (uint64_t) nb + 1) {
        (void) dm_free(c->d_ql_totals);                     // (grows a few times in a context's life)
        c->d_ql_totals = nullptr, c->ql_totals_cap = 0;
        e = alloc((void **) &c->d_ql_totals, ((size_t) nb + 1) * sizeof(unsigned long long));
        if (e == hipSuccess) c->ql_totals_cap = (uint64_t) nb + 1;
    }
    d_totals = c->d_ql_totals;
    if (e == hipSuccess) {
        const int t = t_eff(c, rs);
        const size_t lds = (size_t) ql.n_slices * 4;
        {
            KScope ks(c, "tq_count_kernel", ls);
            if (c->k <= 32)
                COMMET_LAUNCH(tq_count_kernel<uint32_t>, dim3(ql.n_pieces), dim3(256), lds, ls, rs->view(), c->k, t, ql.sbits, ql.n_slices,
                              ql.n_pieces, ql.d_tile_off);
            else
                COMMET_LAUNCH(tq_count_kernel<uint64_t>, dim3(ql.n_pieces), dim3(256), lds, ls, rs->view(), c->k, t, ql.sbits, ql.n_slices,
                              ql.n_pieces, ql.d_tile_off);
        }
        {
            KScope ks(c, "tq_scan_kernels", ls);
            COMMET_LAUNCH(tq_scan_blocks_kernel, dim3(nb), dim3(1024), 0, ls, ql.d_tile_off, entries, d_totals);
            COMMET_LAUNCH(tq_scan_totals_kernel, dim3(1), dim3(1024), 0, ls, d_totals, nb, d_totals + nb);
            COMMET_LAUNCH(tq_scan_add_kernel, dim3(nb), dim3(1024), 0, ls, ql.d_tile_off, entries, d_totals, d_totals + nb);
        }
        e = hipGetLastError();
        unsigned long long total = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&total, d_totals + nb, sizeof total, hipMemcpyDeviceToHost, ls);
        if (e == hipSuccess) e = hipStreamSynchronize(ls);
        ql.n_records = total;
        if (e == hipSuccess && total >= (1ull << 32)) e = hipErrorOutOfMemory;   // tstart is 32 bits (tiled_ok keeps such sets out; before any large allocation)
        if (e == hipSuccess) e = alloc((void **) &ql.d_qaddr, std::max<uint64_t>(total, 1) * 4);
        if (e == hipSuccess) e = alloc((void **) &ql.d_qwho, std::max<uint64_t>(total, 1) * 2);
        if (e == hipSuccess) e = alloc((void **) &ql.d_tstart, std::max<uint64_t>(entries, 1) * 4);
        if (e == hipSuccess) e = alloc((void **) &ql.d_tlen, std::max<uint64_t>(entries, 1) * 2);
        if (e == hipSuccess) {
            KScope ks(c, "tq_bounds_kernel", ls);
            COMMET_LAUNCH(tq_bounds_kernel, dim3((unsigned) ((entries + 255) / 256)), dim3(256), 0, ls, ql.d_tile_off, ql.n_slices,
                               ql.n_pieces, ql.d_tstart, ql.d_tlen);
            e = hipGetLastError();
        }
        if (e == hipSuccess) {
            // reads sorted per round in LDS: as many as keep rpr * (first-hit windows per read) within TQ_FILL_CAP records
            const int64_t fhw = std::max<int64_t>(1, (int64_t) rs->max_len - (int64_t) t * c->k + 1);
            // (even rounds: e.g. 256 reads x 87 windows = 22 272 records are three rounds of 86 reads)
            const uint32_t rounds = (uint32_t) (((uint64_t) TQ_PIECE * (uint64_t) fhw + TQ_FILL_CAP - 1) / TQ_FILL_CAP);
            uint32_t rpr = (TQ_PIECE + rounds - 1) / rounds;
            while (rpr > 1 && (uint64_t) rpr * (uint64_t) fhw > TQ_FILL_CAP) --rpr;
            const size_t lds_fill = ((size_t) 4 * ql.n_slices + 2 * TQ_FILL_CAP) * 4;
            e = hipFuncSetAttribute(c->k <= 32 ? (const void *) tq_fill_kernel<uint32_t> : (const void *) tq_fill_kernel<uint64_t>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_fill);
            if (e == hipSuccess) {
                KScope ks(c, "tq_fill_kernel", ls);
                if (c->k <= 32)
                    COMMET_LAUNCH(tq_fill_kernel<uint32_t>, dim3(ql.n_pieces), dim3(256), lds_fill, ls, rs->view(), c->k, t, ql.sbits,
                                  ql.n_slices, ql.n_pieces, rpr, ql.d_tstart, ql.d_qaddr, ql.d_qwho);
                else
                    COMMET_LAUNCH(tq_fill_kernel<uint64_t>, dim3(ql.n_pieces), dim3(256), lds_fill, ls, rs->view(), c->k, t, ql.sbits,
                                  ql.n_slices, ql.n_pieces, rpr, ql.d_tstart, ql.d_qaddr, ql.d_qwho);
                e = hipGetLastError();
            }
        }
    }
    if (e == hipSuccess) e = hipEventRecord(c->ev_list, ls);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_list, 0);   // whatever the caller queues next (the probe) finds the list complete
    if (e != hipSuccess) {   // no room for the list (or a launch failed): this set keeps the gather kernels
        (void) hipGetLastError();
        ql.release();
        ql.failed = true;
        return 1;
    }
    ql.built = true;
    ql.bytes = (entries + 1) * 8 + ql.n_records * 6 + entries * 6;
    ql.last_use = ++c->ql_clock;
    c->ql_bytes += ql.bytes;
    if (c->ql_bytes > c->ql_budget) (void) shrink_query_lists(c, c->ql_budget, false);   // least recently used first; never one of this job
    return 0;
}

// the scan's result bytes (one per record of the set's query list); no room = this set keeps the gather kernels
int ensure_query_results(commet_ctx *c, const commet_readset *rs)
{
    const uint64_t need = rs->ql.n_records;
    if (c->qres_cap >= need && c->d_qres) return 0;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return 1;
    (void) dm_free(c->d_qres);
    c->d_qres = nullptr, c->qres_cap = 0;
    hipError_t e = dm_malloc((void **) &c->d_qres, std::max<uint64_t>(need, 1));
    if (e == hipErrorOutOfMemory) {   // (the caller holds ql_mu) give back the lists of sets outside this job and try once more
        (void) hipGetLastError();
        if (shrink_query_lists(c, 0, false)) e = dm_malloc((void **) &c->d_qres, std::max<uint64_t>(need, 1));
    }
    if (e != hipSuccess) {
        (void) hipGetLastError();
        rs->ql.failed = true;
        return 1;
    }
    c->qres_cap = need;
    return 0;
}

// what build_query_list and ensure_query_results will ask for, for this context's (k, t): block sizes in bytes (upper bounds: every
// first-hit window of every read a record); false = the set does not qualify for the tiled search whatever the group
bool query_list_blocks(const commet_ctx *c, const commet_readset *rs, uint64_t out[6])
{
    const int t = t_eff(c, rs);
    const int64_t fhw = (int64_t) rs->max_len - (int64_t) t * c->k + 1;
    if (c->k <= TQ_SBITS || c->k > TQ_MAX_K || fhw < 1 || fhw > TQ_MAX_WIN || rs->max_len >= TQ_MAX_LEN || rs->n_reads < (1ull << 20) || rs->n_reads >= (1ull << 32)) return false;
    const uint64_t records = rs->fhw_total;                   // (an upper bound: windows with a non-ACGT base make no record)
    if (records >= (1ull << 32)) return false;
    int sbits = TQ_SBITS;
    if (c->tq_sbits) sbits = std::max(c->k - 10, std::min(c->k - 1, c->tq_sbits));
    const uint64_t entries = ((uint64_t) 1 << (c->k - sbits)) * ((rs->n_reads + TQ_PIECE - 1) / TQ_PIECE);
    out[0] = (entries + 1) * 8, out[1] = records * 4, out[2] = records * 2, out[3] = entries * 4, out[4] = entries * 2;
    out[5] = c->qres_cap >= records ? 0 : records;            // the context's result buffer, one byte per record
    return true;
}

// one pass of rs over the g <= 2 chunk filters in slots slot0 .. slot0 + g - 1 (g == 2: slots 0, 1 with interleaved A planes)
// job_tag_words != 0 (g == 2): the two filters belong to two jobs; job j's found flags go to d_tags + j * job_tag_words (zeroed by the caller)
int launch_search_tiled(commet_ctx *c, const commet_readset *rs, int g, int slot0, const uint64_t *d_sel, uint64_t *d_tags,
                        unsigned long long *d_counters, uint32_t cstride, uint64_t job_tag_words = 0)
{
    if (rs->n_reads == 0) return 0;
    const commet_readset::QueryList &q = rs->ql;
    if (c->qres_cap < q.n_records) return fail("internal error: tiled search without its result buffer");
    QueryListView v;
    v.tile_off = q.d_tile_off, v.qaddr = q.d_qaddr, v.qwho = q.d_qwho, v.tstart = q.d_tstart, v.tlen = q.d_tlen;
    v.n_slices = q.n_slices, v.n_pieces = q.n_pieces, v.sbits = q.sbits;
    FilterGroupView fg;
    fg.slot0 = c->slot_ptr(slot0);
    fg.il_a = g == 1 ? c->slot_ptr(slot0) : c->il_a;   // one filter: its own plane A (stride 1)
    fg.slot_words = 4 * c->plane_words;
    fg.plane_words = c->plane_words;
    fg.g = g;
    // The probe is bound by L2 gathers, the replay by L2-MISSING requests and bookkeeping: different walls.  The set is cut
    // into `parts` runs of pieces; part i's probe and replay go to stream i % 2, every probe waiting for the probe before it
    // (one slice sweep at a time keeps the slice's filter words in L2), so the replay of part i runs beside the probe of
    // part i + 1.  With per-kernel timing on (durations must add up) or a small set: one part, one stream.
    const int mw = mask_words(c, rs);
    const int t = t_eff(c, rs);
    uint32_t parts = (c->kclock.on || q.n_pieces < 4096) ? 1u : (uint32_t) std::max(1, std::min(16, c->tq_parts));
    const unsigned wpx = c->tq_wpx;
    hipEvent_t ev_probe = c->ev_fork, ev_done = c->ev_join;
    for (uint32_t pi = 0; pi < parts; ++pi) {
        const uint32_t p0 = (uint32_t) ((uint64_t) q.n_pieces * pi / parts), p1 = (uint32_t) ((uint64_t) q.n_pieces * (pi + 1) / parts);
        hipStream_t st = (pi & 1u) ? c->aux_stream : c->stream;
        if (pi) HIP_OK(hipStreamWaitEvent(st, ev_probe, 0));      // behind the previous part's probe (and so behind the filter build)
        {
            KScope ks(c, "tq_probe_kernel", st);
            if (g == 1) COMMET_LAUNCH(tq_probe_kernel<1>, dim3(8 * wpx), dim3(256), 0, st, v, fg.il_a, c->d_qres, p0, p1);
            else COMMET_LAUNCH(tq_probe_kernel<2>, dim3(8 * wpx), dim3(256), 0, st, v, fg.il_a, c->d_qres, p0, p1);
        }
        HIP_OK(hipGetLastError());
        if (pi + 1 < parts) HIP_OK(hipEventRecord(ev_probe, st));
        {
            KScope ks(c, "tq_replay_kernel", st);
            const dim3 grid(p1 - p0), block(TQ_PIECE);
#define COMMET_TQ_REPLAY(W, GS, MW) COMMET_LAUNCH((tq_replay_kernel<W, GS, MW>), grid, block, 0, st, rs->view(), v, c->d_qres, fg, c->k, t, d_sel, d_tags, d_counters, cstride, p0, (uint32_t) std::min<int>(c->tq_hit_cap, TQ_HIT_CAP), job_tag_words)
#define COMMET_TQ_REPLAY_MW(W, GS)                 \
    do {                                           \
        if (mw == 2) COMMET_TQ_REPLAY(W, GS, 2);   \
        else if (mw == 3) COMMET_TQ_REPLAY(W, GS, 3); \
        else if (mw == 4) COMMET_TQ_REPLAY(W, GS, 4); \
        else if (mw == 6) COMMET_TQ_REPLAY(W, GS, 6); \
        else COMMET_TQ_REPLAY(W, GS, 8);           \
    } while (0)
            if (c->k <= 32) {
                if (g == 1) COMMET_TQ_REPLAY_MW(uint32_t, 1);
                else COMMET_TQ_REPLAY_MW(uint32_t, 2);
            } else {
                if (g == 1) COMMET_TQ_REPLAY_MW(uint64_t, 1);
                else COMMET_TQ_REPLAY_MW(uint64_t, 2);
            }
#undef COMMET_TQ_REPLAY_MW
#undef COMMET_TQ_REPLAY
        }
        HIP_OK(hipGetLastError());
    }
    if (parts > 1) {   // the second stream's replays join the main stream (the even parts are on it already)
        HIP_OK(hipEventRecord(ev_done, c->aux_stream));
        HIP_OK(hipStreamWaitEvent(c->stream, ev_done, 0));
    }
    HIP_OK(hipGetLastError());
    return 0;
}

// words per bit-sliced entry (32 chunk filters per word) for a job of n_chunks chunks; 0 = the job takes the slot path
int slice_words(const commet_ctx *c, uint64_t n_chunks)
{
    if (c->slice_mode == 1 || c->count_probes) return 0;
    if (c->k < SLICE_MIN_K || c->k > SLICE_MAX_K || n_chunks == 0) return 0;
    if (c->slice_mode == 0 && n_chunks < 8) return 0;
    int gw = c->slice_gw ? c->slice_gw : n_chunks > 128 ? 8 : n_chunks > 64 ? 4 : n_chunks > 32 ? 2 : 1;
    while (gw > 1 && (((uint64_t) 16 * gw) << c->k) > (1ull << 30)) gw /= 2;   // the four tables: at most 1 GiB
    return gw;
}

int ensure_slice_buffers(commet_ctx *c, int gw, uint64_t n_chunks)
{
    const uint64_t G = 32ull * gw;
    const uint64_t stage_words = (G * 4) << (c->k - 5), table_words = ((uint64_t) 4 * gw) << c->k;
    if (c->slice_stage_words < stage_words || c->slice_table_words < table_words || c->slice_chunks_cap < n_chunks) {
        HIP_OK(hipStreamSynchronize(c->stream));
        if (c->slice_stage_words < stage_words) {
            (void) dm_free(c->slice_stage);
            c->slice_stage = nullptr, c->slice_stage_words = 0;
            HIP_OK(dev_alloc(c, (void **) &c->slice_stage, stage_words * 4, true));
            c->slice_stage_words = stage_words;
        }
        if (c->slice_table_words < table_words) {
            (void) dm_free(c->slice_tables);
            c->slice_tables = nullptr, c->slice_table_words = 0;
            HIP_OK(dev_alloc(c, (void **) &c->slice_tables, table_words * 4, true));
            c->slice_table_words = table_words;
        }
        if (c->slice_chunks_cap < n_chunks) {
            (void) dm_free(c->d_slice_chunks);
            c->d_slice_chunks = nullptr, c->slice_chunks_cap = 0;
            HIP_OK(dev_alloc(c, (void **) &c->d_slice_chunks, n_chunks * sizeof(SliceChunk), true));
            c->slice_chunks_cap = n_chunks;
        }
    }
    return 0;
}

// filters of chunks [ci, ci + g) of the plan -> bit-sliced tables (slice_search.hpp)
int launch_slice_build(commet_ctx *c, const commet_readset *rs, const uint64_t *d_sel, uint64_t ci, int g, int gw,
                       uint32_t *tables = nullptr, uint32_t row_words = 0, uint32_t col0 = 0)
{
    if (!tables) tables = c->slice_tables, row_words = (uint32_t) gw, col0 = 0;   // the table of one group (search_sliced_kernel)
    const int tile_bits = std::min(c->k, SLICE_TILE_BITS);
    const uint32_t tiles = 1u << (c->k - tile_bits);
    const size_t lds = (size_t) 4 << (tile_bits - 5);
    HIP_OK(hipFuncSetAttribute((const void *) slice_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    {
        KScope ks(c, "slice_build_kernel", c->stream);
        COMMET_LAUNCH(slice_build_kernel, dim3(4 * tiles, (unsigned) g), dim3(1024), lds, c->stream, rs->view(), d_sel,
                           c->d_slice_chunks + ci, c->k, tile_bits, tiles, c->slice_stage);
    }
    HIP_OK(hipGetLastError());
    const unsigned grid = (unsigned) ((((uint64_t) 4 << (c->k - 5)) + 255) / 256);
    {
        KScope ks(c, "slice_transpose_kernel", c->stream);
        switch (gw) {
        case 1: COMMET_LAUNCH(slice_transpose_kernel<1>, dim3(grid), dim3(256), 0, c->stream, c->slice_stage, c->k, g, tables, row_words, col0); break;
        case 2: COMMET_LAUNCH(slice_transpose_kernel<2>, dim3(grid), dim3(256), 0, c->stream, c->slice_stage, c->k, g, tables, row_words, col0); break;
        case 4: COMMET_LAUNCH(slice_transpose_kernel<4>, dim3(grid), dim3(256), 0, c->stream, c->slice_stage, c->k, g, tables, row_words, col0); break;
        default: COMMET_LAUNCH(slice_transpose_kernel<8>, dim3(grid), dim3(256), 0, c->stream, c->slice_stage, c->k, g, tables, row_words, col0); break;
        }
    }
    HIP_OK(hipGetLastError());
    return 0;
}

int launch_search_sliced(commet_ctx *c, const commet_readset *rs, int g, int gw, const uint64_t *d_sel, uint64_t *d_tags,
                         unsigned long long *d_counters, uint32_t cstride, uint32_t block_stride = 1)
{
    if (rs->n_reads == 0) return 0;
    if ((rs->n_reads + 255) / 256 >= (1ull << 24)) return fail("search launch too large (>= 2^32 reads in one set)");
    const uint64_t blocks = (rs->n_reads + 255) / 256;
    const dim3 grid((unsigned) ((blocks + block_stride - 1) / block_stride)), block(256);
    KScope ks(c, "search_sliced_kernel", c->stream);
    switch (gw) {
    case 1: COMMET_LAUNCH(search_sliced_kernel<1>, grid, block, 0, c->stream, rs->view(), c->slice_tables, c->k, t_eff(c, rs), g, d_sel, d_tags, d_counters, cstride, block_stride); break;
    case 2: COMMET_LAUNCH(search_sliced_kernel<2>, grid, block, 0, c->stream, rs->view(), c->slice_tables, c->k, t_eff(c, rs), g, d_sel, d_tags, d_counters, cstride, block_stride); break;
    case 4: COMMET_LAUNCH(search_sliced_kernel<4>, grid, block, 0, c->stream, rs->view(), c->slice_tables, c->k, t_eff(c, rs), g, d_sel, d_tags, d_counters, cstride, block_stride); break;
    default: COMMET_LAUNCH(search_sliced_kernel<8>, grid, block, 0, c->stream, rs->view(), c->slice_tables, c->k, t_eff(c, rs), g, d_sel, d_tags, d_counters, cstride, block_stride); break;
    }
    HIP_OK(hipGetLastError());
    return 0;
}

// ---- wide rows (slice_search.hpp): every chunk filter of the job — or as many as the table budget allows — in one table ----
struct WidePlan {
    uint32_t nw = 0;              // words per row that hold chunks (a multiple of WIDE_GROUP_WORDS); 0 = no wide pass
    uint32_t rw = 0;              // row stride in words (a multiple of 32: rows start on 128-byte lines)
    uint64_t chunks_per_pass = 0;
    int lpr = 0, np = 0;          // lanes per read, 16-byte pieces per lane
};

WidePlan wide_plan(const commet_ctx *c, uint64_t n_chunks, int slice_gw)
{
    WidePlan w;
    if (!slice_gw || c->slice_wide == 1) return w;
    if (c->slice_wide == 0 && n_chunks <= 256) return w;             // one table of the narrow kind holds them all
    const uint64_t groups = (n_chunks + 255) / 256;
    // four tables of 2^k rows: 16 bytes per row word and key; at most a third of what is free now, and 48 GiB
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return w;
    const uint64_t have = (uint64_t) c->wide_table_words * 4 + dm_filed_bytes(c->device);   // (what the context already holds, and what the library keeps for reuse, counts as free)
    const uint64_t budget = std::min<uint64_t>(((uint64_t) free_b + have) / 3, 48ull << 30);
    uint64_t cap = std::min<uint64_t>(WIDE_MAX_ROW_WORDS, budget / (16ull << c->k));
    if (c->wide_cap_words) cap = std::min<uint64_t>(cap, c->wide_cap_words);
    cap = cap / WIDE_GROUP_WORDS * WIDE_GROUP_WORDS;
    if (cap < WIDE_GROUP_WORDS) return w;
    const uint64_t passes = (groups * WIDE_GROUP_WORDS + cap - 1) / cap;
    const uint64_t groups_per_pass = (groups + passes - 1) / passes;
    w.nw = (uint32_t) (groups_per_pass * WIDE_GROUP_WORDS);
    w.rw = (w.nw + 31u) & ~31u;   // rows start on 128-byte lines: a row of 1312 bytes is 11 lines, never 12
    w.chunks_per_pass = groups_per_pass * 256;
    const uint32_t pieces = w.nw / 4;
    w.lpr = pieces <= 8 ? 8 : pieces <= 16 ? 16 : pieces <= 32 ? 32 : 64;
    w.np = pieces <= 64 ? 1 : 2;
    return w;
}

int ensure_wide_tables(commet_ctx *c, const WidePlan &w)
{
    const uint64_t words = ((uint64_t) 4 * w.rw) << c->k;
    if (c->wide_table_words >= words) return 0;
    HIP_OK(hipStreamSynchronize(c->stream));
    (void) dm_free(c->wide_tables);
    c->wide_tables = nullptr, c->wide_table_words = 0;
    if (dev_alloc(c, (void **) &c->wide_tables, words * 4, true) != hipSuccess) {
        (void) hipGetLastError();
        return 1;                                                    // the caller falls back to the narrow tables
    }
    c->wide_table_words = words;
    return 0;
}

int launch_search_wide(commet_ctx *c, const commet_readset *rs, const WidePlan &w, int g, const uint64_t *d_sel, uint64_t *d_tags,
                       unsigned long long *d_counters, uint32_t cstride)
{
    if (rs->n_reads == 0) return 0;
    const uint64_t reads_per_block = 4ull * (64 / w.lpr);
    const uint64_t blocks = (rs->n_reads + reads_per_block - 1) / reads_per_block;
    if (blocks >= (1ull << 31)) return fail("search launch too large");
    const dim3 grid((unsigned) blocks), block(256);
    const int t = t_eff(c, rs);
    KScope ks(c, "search_wide_kernel", c->stream);
#define COMMET_WIDE(LPR, NP) COMMET_LAUNCH((search_wide_kernel<LPR, NP>), grid, block, 0, c->stream, rs->view(), c->wide_tables, c->k, t, g, w.nw, w.rw, d_sel, d_tags, d_counters, cstride)
    if (w.np == 2) COMMET_WIDE(64, 2);
    else if (w.lpr == 64) COMMET_WIDE(64, 1);
    else if (w.lpr == 32) COMMET_WIDE(32, 1);
    else if (w.lpr == 16) COMMET_WIDE(16, 1);
    else COMMET_WIDE(8, 1);
#undef COMMET_WIDE
    HIP_OK(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

// (here rather than in cache.hpp: they need the tiled search's geometry)
uint64_t commet_readset_cache_estimate(commet_ctx *c, const commet_readset *rs)
{
    uint64_t b[6];
    if (rs->ctx != c || !rs->finalized || !query_list_blocks(c, rs, b)) return 0;
    return b[0] + b[1] + b[2] + b[3] + b[4];
}

int commet_readset_reserve_cache(commet_ctx *c, const commet_readset *rs)
{
    if (rs->ctx != c) return fail("read set belongs to another context");
    if (!rs->finalized) return fail("read set not finalized");
    HIP_OK(hipSetDevice(c->device));
    uint64_t b[6];
    {
        std::lock_guard<std::mutex> lk(c->ql_mu);
        const bool still = rs->ql_reserved.load() && rs->ql_reserved_at.load() == g_devmem.trims.load();
        if (rs->ql.built || still || !g_devmem.enabled() || !query_list_blocks(c, rs, b)) return 0;
    }
    for (int i = 0; i < 6; ++i)
        if (b[i] && dm_reserve((size_t) b[i]) != hipSuccess) return 0;      // no room: the set keeps the cap's rule
    rs->ql_reserved_at.store(g_devmem.trims.load());
    rs->ql_reserved.store(true);
    return 0;
}

}  // extern "C"
