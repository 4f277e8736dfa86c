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

// tables[(((plane << k) + key) * row_words) + col0 + cw] bit j = stage plane of chunk 32 * cw + j, bit key (0 for chunks >= g).
// row_words = GW, col0 = 0: the table of one group of 32 * GW chunks; wide rows (below): the group's GW words sit at
// column col0 of rows of row_words words that hold many groups side by side.
template <int GW>
__global__ __launch_bounds__(256) void slice_transpose_kernel(const uint32_t *__restrict__ stage, int k, int g,
                                                              uint32_t *__restrict__ tables, uint32_t row_words, uint32_t col0)
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
        uint32_t *dst = tables + (((plane << k) + 32 * i) * row_words) + col0 + cw;
#pragma unroll
        for (int b = 0; b < 32; ++b) dst[(uint64_t) b * row_words] = x[b];
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
__global__ __launch_bounds__(256) COMMET_SGPRS void search_sliced_kernel(ReadsView rv, const uint32_t *__restrict__ tables, int k, int t, int g,
                                                            const uint64_t *__restrict__ sel, uint64_t *__restrict__ tags,
                                                            unsigned long long *__restrict__ counters, uint32_t cstride,
                                                            uint32_t block_stride)
{
    // block_stride > 1: only every block_stride-th block of 256 reads is launched (the sampling probe of the auto mode,
    // capi.hip: its selection bitmap has reads in those blocks only)
    const uint64_t r = (uint64_t) blockIdx.x * block_stride * 256ull + threadIdx.x;
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

// ---------------------------------------------------------------------------------------------------------------------
// Wide rows: ALL chunk filters of the job (or as many as the table budget allows) side by side in one table.
//
// search_sliced_kernel is bound by the rate of L2-missing requests (one 32-byte entry of a 64-byte sector per request:
// 700 GB of HBM traffic per pass at configs[4], 41 passes of 20 M reads, profiles/r03_c5_before), and two things it counts
// on do not hold at k = 21 with 10 421 chunk filters: the short circuit a -> b -> c -> d prunes nothing at entry granularity
// (an entry of 256 chunks is non-zero after A, after A & B and mostly after A & B & C), and plane D hardly filters at all —
// keyd = keya | keyb has three one bits in four, so index and query keys crowd into the same few heavy keys and a random
// window passes plane D with probability 0.68, not 0.11.  A window is therefore a four-lane hit in a chunk that shares
// nothing with the read with probability 9e-4, "two hit windows" is true for ~130 chunks per read, and the exact replay
// of those candidates was most of the kernel's requests.
//
// Here a key's entry is a ROW of up to 16 384 chunk bits (row_words words; 10 421 chunks = 328 words = 1.3 KB; the four
// tables of configs[4] are 11 GB of the 288 GB), and a group of LPR lanes works on ONE read: lane l owns the 16-byte pieces
// l, l + LPR, ... of every row.
//   (1) row pass — per window, both strands: the rows of planes A, B, C (plane D is left to the replay: it would add a
//       third to the bytes and removes a third of the candidates) are ANDed, six coalesced row fetches of which every
//       byte is used; no divergence, no short circuit, the set is scanned once.  Candidate rule, per chunk bit and
//       strand: the windows are cut into blocks of k; t non-overlapping hits lie in t different blocks, and hit j of the
//       reference's greedy scan ends at or before last - (t - j) k.  With J = min(t, 3): a chunk can find the read on a
//       strand only if J different blocks hold a hit among the windows ending at or before last - (t - J) k (a 2-bit
//       saturating block counter per chunk bit: three masks per strand).  ~2 candidates per read at configs[4].
//   (2) exact replay of the candidate chunks, smallest first, by the group with lane = window: every lane tests one
//       window's four bits in chunk c, the ballot is the strand's hit mask, and the reference's greedy rule
//       (search_reads.h:45-83: a hit, then the next complete window k bases on, t hits tag the read) is walked on the
//       mask, forward strand first.  The first chunk that finds the read tags it — the reference's chunk order.
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t WIDE_MAX_ROW_WORDS = 512;             // 16 384 chunks per pass: LPR = 64 lanes x NP = 2 pieces of 4 words
constexpr uint32_t WIDE_GROUP_WORDS = 8;                 // rows are filled in groups of 256 chunks (slice_build + transpose<8>)

__device__ __forceinline__ uint4 and3(const uint4 &a, const uint4 &b, const uint4 &c)
{
    return make_uint4(a.x & b.x & c.x, a.y & b.y & c.y, a.z & b.z & c.z, a.w & b.w & c.w);
}
__device__ __forceinline__ void or_into(uint4 &d, const uint4 &a) { d.x |= a.x, d.y |= a.y, d.z |= a.z, d.w |= a.w; }
// one more block with a hit for the chunk bits set in cur: 2-bit saturating counter (c1 c0), cur cleared
__device__ __forceinline__ void wide_fold(uint32_t &cur, uint32_t &c0, uint32_t &c1)
{
    const uint32_t add = cur & ~(c0 & c1);
    c1 |= c0 & add;
    c0 ^= add;
    cur = 0;
}
__device__ __forceinline__ void wide_fold(uint4 &cur, uint4 &c0, uint4 &c1)
{
    wide_fold(cur.x, c0.x, c1.x), wide_fold(cur.y, c0.y, c1.y), wide_fold(cur.z, c0.z, c1.z), wide_fold(cur.w, c0.w, c1.w);
}
__device__ __forceinline__ uint4 wide_at_least(const uint4 &c0, const uint4 &c1, int J)
{
    if (J <= 1) return make_uint4(c0.x | c1.x, c0.y | c1.y, c0.z | c1.z, c0.w | c1.w);
    if (J == 2) return c1;
    return make_uint4(c0.x & c1.x, c0.y & c1.y, c0.z & c1.z, c0.w & c1.w);
}
__device__ __forceinline__ uint32_t wide_word(const uint4 &v, uint32_t j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

template <int LPR, int NP>
__global__ __launch_bounds__(256) void search_wide_kernel(ReadsView rv, const uint32_t *__restrict__ tables, int k, int t, int g,
                                                          uint32_t nw, uint32_t rw, const uint64_t *__restrict__ sel,
                                                          uint64_t *__restrict__ tags, unsigned long long *__restrict__ counters,
                                                          uint32_t cstride)
{
    constexpr int RPW = 64 / LPR;                         // reads per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / LPR, sl = lane % LPR;
    const uint64_t r = ((uint64_t) blockIdx.x * 4 + wave) * RPW + grp;
    bool active = r < rv.n;
    if (active && sel) active = (sel[r >> 6] >> (r & 63)) & 1ull;
    if (active && tags) active = !((tags[r >> 6] >> (r & 63)) & 1ull);
    const uint64_t plane_stride = ((uint64_t) rw) << k;   // words per table
    const uint32_t *TA = tables, *TB = TA + plane_stride, *TC = TB + plane_stride, *TD = TC + plane_stride;
    uint64_t t0 = 0;
    uint32_t len = 0;
    if (active) read_extent(rv, r, t0, len);
    const uint32_t *p = rv.planes + 3 * t0;
    const int sh = 32 - k;
    const uint32_t mask = (1u << k) - 1u;
    const int last = (int) len - 1;
    const int J = min(t, 3);
    const int lim = active ? last - (t - J) * k : -1;     // hit J of a strand's scan ends at or before lim
    uint4 cand_f[NP], cand_r[NP];
    bool have[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) have[i] = 4u * (uint32_t) (sl + LPR * i) < nw;
    // (1) row pass over the windows ending at or before lim; every lane of the group rolls the same window
    {
        uint4 cur_f[NP], cur_r[NP], f0[NP], f1[NP], r0[NP], r1[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) cur_f[i] = cur_r[i] = f0[i] = f1[i] = r0[i] = r1[i] = make_uint4(0, 0, 0, 0);
        uint32_t wh = 0, wl = 0, run = 0;
        int in_block = 0;                                  // window positions since the block began
        for (uint32_t w = 0; (int) (w * 32u) <= lim; ++w) {
            const uint32_t hi = p[3 * w], lo = p[3 * w + 1], va = p[3 * w + 2];
            const uint32_t nb = (uint32_t) min(32, lim - (int) (w * 32u) + 1);
            for (uint32_t j = 0; j < nb; ++j) {
                wh = (wh >> 1) | (((hi >> j) & 1u) << (k - 1));
                wl = (wl >> 1) | (((lo >> j) & 1u) << (k - 1));
                run = ((va >> j) & 1u) ? run + 1 : 0;
                if ((int) (32u * w + j) < k - 1) continue;              // not a window position yet
                if (run >= (uint32_t) k) {
                    const uint32_t ka = __brev(wh) >> sh, kb = __brev(wl) >> sh;
                    const uint32_t ra = ~wh & mask, rb = ~wl & mask;
                    bool selfp;
                    const uint32_t addr = psi_a<uint32_t>(ka, k, selfp);
                    const uint4 *af = (const uint4 *) (TA + (uint64_t) addr * rw), *ar = (const uint4 *) (TA + (uint64_t) (selfp ? addr : addr ^ 1u) * rw);
                    const uint4 *bf = (const uint4 *) (TB + (uint64_t) kb * rw), *br = (const uint4 *) (TB + (uint64_t) rb * rw);
                    const uint4 *cf = (const uint4 *) (TC + (uint64_t) (ka ^ kb) * rw), *cr = (const uint4 *) (TC + (uint64_t) (ra ^ rb) * rw);
                    uint4 x[NP][6];
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        if (!have[i]) continue;
                        const int pc = sl + LPR * i;
                        x[i][0] = af[pc], x[i][1] = bf[pc], x[i][2] = cf[pc];
                        x[i][3] = ar[pc], x[i][4] = br[pc], x[i][5] = cr[pc];
                    }
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        if (!have[i]) continue;
                        or_into(cur_f[i], and3(x[i][0], x[i][1], x[i][2]));
                        or_into(cur_r[i], and3(x[i][3], x[i][4], x[i][5]));
                    }
                }
                if (++in_block == k) {                      // the block is complete
                    in_block = 0;
#pragma unroll
                    for (int i = 0; i < NP; ++i) wide_fold(cur_f[i], f0[i], f1[i]), wide_fold(cur_r[i], r0[i], r1[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            wide_fold(cur_f[i], f0[i], f1[i]), wide_fold(cur_r[i], r0[i], r1[i]);   // the last, partial block
            cand_f[i] = wide_at_least(f0[i], f1[i], J), cand_r[i] = wide_at_least(r0[i], r1[i], J);
            if (!have[i]) cand_f[i] = cand_r[i] = make_uint4(0, 0, 0, 0);
        }
    }
    // (2) exact replay of the candidate chunks, smallest chunk first.  The loops are uniform over the wave (ballots and
    // shuffles inside); a group without work runs them with its predicate off.
    int found_chunk = -1;
    for (;;) {
        uint32_t best = 0xFFFFFFFFu;                       // this lane's smallest candidate chunk (its words ascend with i)
        if (found_chunk < 0) {
#pragma unroll
            for (int i = NP - 1; i >= 0; --i) {
                const uint32_t w0 = 4u * (uint32_t) (sl + LPR * i);
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const uint32_t m = wide_word(cand_f[i], (uint32_t) j) | wide_word(cand_r[i], (uint32_t) j);
                    if (m) best = (w0 + (uint32_t) j) * 32u + (uint32_t) __ffs((int) m) - 1u;
                }
            }
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) best = min(best, (uint32_t) __shfl_xor((int) best, o, LPR));   // the group's smallest
        const bool go = best != 0xFFFFFFFFu && (int) best < g;
        if (!__any(best != 0xFFFFFFFFu)) break;
        const uint32_t cw = best >> 5, cb = best & 31u;
        // the owner of word cw reads the chunk's strand flags and drops the bits (columns past the last chunk hold zeros,
        // so best >= g cannot happen; such a bit would be dropped here as well)
        bool mine_f = false, mine_r = false;
        if (best != 0xFFFFFFFFu) {
            const uint32_t pc = cw >> 2, wj = cw & 3u, keep = ~(1u << cb);
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if (pc == (uint32_t) (sl + LPR * i)) {
                    mine_f = (wide_word(cand_f[i], wj) >> cb) & 1u, mine_r = (wide_word(cand_r[i], wj) >> cb) & 1u;
                    if (wj == 0) cand_f[i].x &= keep, cand_r[i].x &= keep;
                    else if (wj == 1) cand_f[i].y &= keep, cand_r[i].y &= keep;
                    else if (wj == 2) cand_f[i].z &= keep, cand_r[i].z &= keep;
                    else cand_f[i].w &= keep, cand_r[i].w &= keep;
                }
        }
        uint64_t bf = __ballot(mine_f), br = __ballot(mine_r);
        if constexpr (LPR < 64) bf = (bf >> (grp * LPR)) & ((1ull << LPR) - 1ull), br = (br >> (grp * LPR)) & ((1ull << LPR) - 1ull);
        bool found = false;
        for (int strand = 0; strand < 2; ++strand) {
            const bool flagged = go && (strand ? br != 0 : bf != 0);   // a strand without J hit blocks cannot find the read
            int seen = 0, next_ok = 0;
            bool dead = false;
            for (int qb = k - 1; __any(flagged && !found && !dead && qb <= last && qb + (t - seen - 1) * k <= last); qb += LPR) {
                const int q = qb + sl;
                bool hit = false;
                if (flagged && !found && !dead && q <= last) {
                    ItemWords<uint32_t> it;
                    it.load(p, (uint32_t) q >> 5);
                    uint32_t wh, wl;
                    if (it.window((uint32_t) q & 31u, k, mask, wh, wl)) {
                        uint32_t ka, kb;
                        if (strand == 0) ka = __brev(wh) >> sh, kb = __brev(wl) >> sh;
                        else ka = ~wh & mask, kb = ~wl & mask;
                        const uint32_t va = TA[(uint64_t) psi_a<uint32_t>(ka, k) * rw + cw], vb = TB[(uint64_t) kb * rw + cw];
                        const uint32_t vc = TC[(uint64_t) (ka ^ kb) * rw + cw], vd = TD[(uint64_t) (ka | kb) * rw + cw];
                        hit = ((va & vb & vc & vd) >> cb) & 1u;
                    }
                }
                uint64_t m = __ballot(hit);
                if constexpr (LPR < 64) m = (m >> (grp * LPR)) & ((1ull << LPR) - 1ull);
                while (m && !found && !dead) {
                    const int qq = qb + (__ffsll((long long) m) - 1);
                    m &= m - 1ull;
                    if (qq < next_ok) continue;
                    if (qq + (t - seen - 1) * k > last) {   // the missing hits no longer fit behind this window (exact, see search_kernel)
                        dead = true;
                        break;
                    }
                    ++seen;
                    next_ok = qq + k;                       // hash.clear(), search_reads.h:60: the next complete window ends k bases on
                    if (seen >= t) found = true;
                }
            }
        }
        if (go && found) found_chunk = (int) best;
    }
    if (found_chunk >= 0 && sl == 0) {
        if (tags) atomicOr((unsigned long long *) &tags[r >> 6], 1ull << (r & 63));
        if (counters) atomicAdd(&counters[(uint64_t) found_chunk * cstride + 1], 1ull);   // found reads only: rare
    }
}

}  // namespace commet
