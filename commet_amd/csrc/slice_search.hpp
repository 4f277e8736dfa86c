// slice_search.hpp — the many-small-chunks regime (small k): chunk filters bit-sliced across a machine word.
//
// At k = 21 the reference's filter is 1 MiB and a chunk holds 244 140 k-mers (index_and_search.cpp:73), so a set of
// 20 M x 150 bp reads is indexed in 10 421 chunks and every search set is re-scanned once per chunk
// (index_and_search.cpp:255-277) — BASELINE configs[4].  A read's result is the OR over the chunks of a per-chunk
// predicate, and the filter layout is free, so here the filters of a GROUP of G = 32 * GW chunks are stored
// bit-sliced: entry `key` of plane p is a G-bit word whose bit c is chunk c's bit at `key`,
//       T_p[key * GW + (c >> 5)]  bit (c & 31),
// and ONE load answers "which of the G chunks have this key" — the search set is scanned once per group, not once
// per chunk, and the four-lane test of a window is three ANDs of G-bit words that stop as soon as no chunk is left.
// Plane A keeps the strand-paired addressing psi_a (kernels.hpp): the entries of a key and of its reverse-complement
// partner are neighbours, so one load of 2 * GW words serves both strands.
//
//   slice_build_kernel      one workgroup per (chunk, plane, tile of 2^20 bits): the chunk's keys of that plane are
//                           OR-ed into an LDS tile, the tile goes to the chunk's own bit-plane in a staging buffer
//                           (replaces BloomFilter::feed applied by index_reads.h:51-59 to one chunk)
//   slice_transpose_kernel  staging bit-planes of the group's chunks -> bit-sliced tables (32 x 32 bit transposes)
//   search_sliced_kernel    lane per read.  Word-parallel pass: for every window that can be a first hit or (t >= 2) a
//                           second one, the G-bit masks of chunks in which the window is a full four-lane hit, per
//                           strand; a chunk in which the read can be found must show such hits in two different
//                           windows of one strand, the first of them among the first-hit windows (one window if
//                           t = 1).  Chunks that pass — true sharing, plus ~1e-4 of the rest by chance — are then
//                           replayed one by one with the reference's exact control flow (search_reads.h:45-83) in
//                           increasing chunk order; the first chunk that finds the read tags it, exactly as the
//                           reference's chunk loop would.
#pragma once

#include "kernels.hpp"

namespace commet {

constexpr int SLICE_TILE_BITS = 20;                      // LDS tile of the per-chunk build: 2^20 bits = 128 KiB
constexpr int SLICE_MIN_K = 12, SLICE_MAX_K = 24;

struct SliceChunk {
    uint64_t first;   // first read of the chunk's range
    uint64_t count;   // reads in the range (selected or not); 0 = empty chunk
};

// stage[((c * 4 + plane) << (k - 5)) + word]: chunk c's plane as a plain bit array (plane A at psi_a addresses)
__global__ __launch_bounds__(1024) void slice_build_kernel(ReadsView rv, const uint64_t *__restrict__ sel,
                                                           const SliceChunk *__restrict__ chunks, int k, int tile_bits,
                                                           uint32_t tiles_per_plane, uint32_t *__restrict__ stage)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t tile[];
    const uint32_t c = blockIdx.y;
    const uint32_t plane = blockIdx.x / tiles_per_plane, ti = blockIdx.x % tiles_per_plane;
    const uint32_t tile_words = 1u << (tile_bits - 5);
    for (uint32_t i = threadIdx.x; i < tile_words; i += 1024) tile[i] = 0;
    __syncthreads();
    const SliceChunk ch = chunks[c];
    const int sh = 32 - k;
    const uint32_t tmask = (1u << tile_bits) - 1u;
    for (uint64_t r = ch.first + threadIdx.x; r < ch.first + ch.count; r += 1024) {
        if (sel && !((sel[r >> 6] >> (r & 63)) & 1ull)) continue;
        uint64_t t0;
        uint32_t len;
        read_extent(rv, r, t0, len);
        const uint32_t *p = rv.planes + 3 * t0;
        uint32_t wh = 0, wl = 0, run = 0;
        for (uint32_t w = 0; w * 32u < len; ++w) {
            const uint32_t hi = p[3 * w], lo = p[3 * w + 1], va = p[3 * w + 2];
            const uint32_t nb = min(32u, len - w * 32u);
            for (uint32_t j = 0; j < nb; ++j) {
                wh = (wh >> 1) | (((hi >> j) & 1u) << (k - 1));
                wl = (wl >> 1) | (((lo >> j) & 1u) << (k - 1));
                run = ((va >> j) & 1u) ? run + 1 : 0;
                if (run < (uint32_t) k) continue;
                const uint32_t ka = __brev(wh) >> sh, kb = __brev(wl) >> sh;
                const uint32_t key = plane == 0 ? psi_a<uint32_t>(ka, k) : plane == 1 ? kb : plane == 2 ? (ka ^ kb) : (ka | kb);
                if ((key >> tile_bits) == ti) atomicOr(&tile[(key & tmask) >> 5], 1u << (key & 31u));
            }
        }
    }
    __syncthreads();
    uint32_t *dst = stage + (((uint64_t) c * 4 + plane) << (k - 5)) + (uint64_t) ti * tile_words;
    for (uint32_t i = threadIdx.x; i < tile_words; i += 1024) dst[i] = tile[i];
}

// out[b] bit j = in[j] bit b
__device__ __forceinline__ void transpose32(uint32_t (&x)[32])
{
    uint32_t m = 0x0000FFFFu;
#pragma unroll
    for (uint32_t j = 16; j; j >>= 1, m ^= m << j) {
#pragma unroll
        for (uint32_t q = 0; q < 32; q = (q + j + 1) & ~j) {
            const uint32_t t = ((x[q] >> j) ^ x[q + j]) & m;
            x[q] ^= t << j;
            x[q + j] ^= t;
        }
    }
}

// tables[(((plane << k) + key) * GW) + cw] bit j = stage plane of chunk 32 * cw + j, bit key (0 for chunks >= g)
template <int GW>
__global__ __launch_bounds__(256) void slice_transpose_kernel(const uint32_t *__restrict__ stage, int k, int g,
                                                              uint32_t *__restrict__ tables)
{
    const uint64_t pw = 1ull << (k - 5);
    const uint64_t idx = blockIdx.x * 256ull + threadIdx.x;
    if (idx >= 4 * pw) return;
    const uint64_t plane = idx / pw, i = idx % pw;
#pragma unroll
    for (int cw = 0; cw < GW; ++cw) {
        uint32_t x[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int c = cw * 32 + j;
            x[j] = c < g ? stage[(((uint64_t) c * 4 + plane) << (k - 5)) + i] : 0u;
        }
        transpose32(x);
        uint32_t *dst = tables + (((plane << k) + 32 * i) * GW) + cw;
#pragma unroll
        for (int b = 0; b < 32; ++b) dst[(uint64_t) b * GW] = x[b];
    }
}

template <int GW> struct SliceWord {
    uint32_t x[GW];
    __device__ __forceinline__ void load(const uint32_t *q)
    {
        if constexpr (GW == 1) x[0] = q[0];
        else if constexpr (GW == 2) {
            const uint2 v = *(const uint2 *) q;
            x[0] = v.x, x[1] = v.y;
        } else {
#pragma unroll
            for (int i = 0; i < GW; i += 4) {
                const uint4 v = *(const uint4 *) (q + i);
                x[i] = v.x, x[i + 1] = v.y, x[i + 2] = v.z, x[i + 3] = v.w;
            }
        }
    }
    __device__ __forceinline__ bool any() const
    {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < GW; ++i) o |= x[i];
        return o != 0;
    }
    __device__ __forceinline__ void and_with(const SliceWord &b)
    {
#pragma unroll
        for (int i = 0; i < GW; ++i) x[i] &= b.x[i];
    }
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int i = 0; i < GW; ++i) x[i] = 0;
    }
};

template <int GW>
__global__ __launch_bounds__(256) void search_sliced_kernel(ReadsView rv, const uint32_t *__restrict__ tables, int k, int t, int g,
                                                            const uint64_t *__restrict__ sel, uint64_t *__restrict__ tags,
                                                            unsigned long long *__restrict__ counters, uint32_t cstride)
{
    const uint64_t r = blockIdx.x * 256ull + threadIdx.x;
    const uint64_t word = r >> 6;
    const int lane = threadIdx.x & 63;
    const bool in_range = (word << 6) < rv.n;
    uint64_t selw = ~0ull, tagw = 0;
    if (in_range) {
        if (sel) selw = sel[word];
        if (tags) tagw = tags[word];
    }
    const bool active = (r < rv.n) && ((selw >> lane) & 1ull) && !((tagw >> lane) & 1ull);
    const uint32_t *TA = tables, *TB = tables + ((1ull << k) * GW), *TC = tables + ((2ull << k) * GW), *TD = tables + ((3ull << k) * GW);
    bool found = false;
    int found_chunk = -1;
    if (active) {
        uint64_t t0;
        uint32_t len;
        read_extent(rv, r, t0, len);
        const uint32_t *p = rv.planes + 3 * t0;
        const int sh = 32 - k;
        const uint32_t mask = (1u << k) - 1u;
        const int last = (int) len - 1;
        const int pe = last - (t - 1) * k;                     // last window that can be a strand's FIRST hit
        const int lim = t >= 2 ? last - (t - 2) * k : last;    // last window that can be its second hit
        SliceWord<GW> once_f, once_r, cand;
        once_f.clear(), once_r.clear(), cand.clear();
        // (1) word-parallel pass over the windows ending at or before lim
        {
            uint32_t wh = 0, wl = 0, run = 0;
            for (uint32_t w = 0; (int) (w * 32u) <= lim; ++w) {
                const uint32_t hi = p[3 * w], lo = p[3 * w + 1], va = p[3 * w + 2];
                const uint32_t nb = (uint32_t) min(32, lim - (int) (w * 32u) + 1);
                for (uint32_t j = 0; j < nb; ++j) {
                    wh = (wh >> 1) | (((hi >> j) & 1u) << (k - 1));
                    wl = (wl >> 1) | (((lo >> j) & 1u) << (k - 1));
                    run = ((va >> j) & 1u) ? run + 1 : 0;
                    if (run < (uint32_t) k) continue;
                    const int q = (int) (32u * w + j);
                    const uint32_t ka = __brev(wh) >> sh, kb = __brev(wl) >> sh;
                    bool selfp;
                    const uint32_t addr = psi_a<uint32_t>(ka, k, selfp);
                    SliceWord<GW> mf, mr, x;
                    if constexpr (GW <= 2) {
                        // the pair (addr & ~1, addr | 1) is 2 * GW contiguous words: one load for both strands
                        SliceWord<2 * GW> pr;
                        pr.load(TA + (uint64_t) (addr & ~1u) * GW);
#pragma unroll
                        for (int i = 0; i < GW; ++i) {
                            mf.x[i] = pr.x[(addr & 1u) * GW + i];
                            mr.x[i] = selfp ? mf.x[i] : pr.x[((addr & 1u) ^ 1u) * GW + i];
                        }
                    } else {
                        mf.load(TA + (uint64_t) addr * GW);
                        if (selfp) mr = mf;
                        else mr.load(TA + (uint64_t) (addr ^ 1u) * GW);
                    }
                    if (mf.any()) {
                        x.load(TB + (uint64_t) kb * GW), mf.and_with(x);
                        if (mf.any()) {
                            x.load(TC + (uint64_t) (ka ^ kb) * GW), mf.and_with(x);
                            if (mf.any()) x.load(TD + (uint64_t) (ka | kb) * GW), mf.and_with(x);
                        }
                    }
                    if (mr.any()) {
                        const uint32_t ra = ~wh & mask, rb = ~wl & mask;
                        x.load(TB + (uint64_t) rb * GW), mr.and_with(x);
                        if (mr.any()) {
                            x.load(TC + (uint64_t) (ra ^ rb) * GW), mr.and_with(x);
                            if (mr.any()) x.load(TD + (uint64_t) (ra | rb) * GW), mr.and_with(x);
                        }
                    }
                    if (t >= 2) {
#pragma unroll
                        for (int i = 0; i < GW; ++i) cand.x[i] |= (once_f.x[i] & mf.x[i]) | (once_r.x[i] & mr.x[i]);
                    }
                    if (q <= pe) {
#pragma unroll
                        for (int i = 0; i < GW; ++i) once_f.x[i] |= mf.x[i], once_r.x[i] |= mr.x[i];
                    }
                }
            }
            if (t < 2) {
#pragma unroll
                for (int i = 0; i < GW; ++i) cand.x[i] = once_f.x[i] | once_r.x[i];
            }
        }
        // (2) exact replay of the candidate chunks, in chunk order (search_reads.h:45-83 on chunk c's bits)
#pragma unroll 1
        for (int cw = 0; cw < GW && !found; ++cw) {
            uint32_t m = 0;
#pragma unroll
            for (int i = 0; i < GW; ++i)
                if (i == cw) m = cand.x[i];   // (a select per word keeps cand in registers)
            while (m && !found) {
                const int cb = __ffs((int) m) - 1;
                m &= m - 1u;
                const int c = cw * 32 + cb;
                if (c >= g) break;
                auto bit = [&](const uint32_t *T, uint32_t key) -> bool { return (T[(uint64_t) key * GW + cw] >> cb) & 1u; };
                for (int strand = 0; strand < 2 && !found; ++strand) {
                    uint32_t wh = 0, wl = 0, run = 0;
                    int seen = 0;
                    bool dead = false;
                    for (uint32_t w = 0; w * 32u < len && !found && !dead; ++w) {
                        const uint32_t hi = p[3 * w], lo = p[3 * w + 1], va = p[3 * w + 2];
                        const uint32_t nb = min(32u, len - w * 32u);
                        for (uint32_t j = 0; j < nb; ++j) {
                            wh = (wh >> 1) | (((hi >> j) & 1u) << (k - 1));
                            wl = (wl >> 1) | (((lo >> j) & 1u) << (k - 1));
                            run = ((va >> j) & 1u) ? run + 1 : 0;
                            // exact pruning (see search_kernel): the missing hits no longer fit behind this window
                            if ((int) (32u * w + j) + (t - seen - 1) * k > last) {
                                dead = true;
                                break;
                            }
                            if (run < (uint32_t) k) continue;
                            uint32_t ka, kb;
                            if (strand == 0) ka = __brev(wh) >> sh, kb = __brev(wl) >> sh;
                            else ka = ~wh & mask, kb = ~wl & mask;
                            if (bit(TA, psi_a<uint32_t>(ka, k)) && bit(TB, kb) && bit(TC, ka ^ kb) && bit(TD, ka | kb)) {
                                ++seen;
                                run = 0;                       // hash.clear(), search_reads.h:60
                                if (seen >= t) {
                                    found = true;
                                    break;
                                }
                            }
                        }
                    }
                }
                if (found) found_chunk = c;
            }
        }
    }
    const uint64_t fb = __ballot(found);
    if (lane == 0 && in_range && tags && fb) tags[word] = tagw | fb;
    if (counters && found) atomicAdd(&counters[(uint64_t) found_chunk * cstride + 1], 1ull);   // found reads only: rare
}

}  // namespace commet
