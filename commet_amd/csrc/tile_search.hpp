// tile_search.hpp — the search of a large set with its lane-a gathers served from L2 instead of HBM.
//
// search_group_kernel (kernels.hpp) sits on the chip's random-request ceiling: ~54 G gathers/s that miss L2, whatever
// the table (profiles/r01_membench_random_access.jsonl).  The same gathers run at 150-230 G/s when the workgroups of
// one XCD work, at any one time, on a window of the table that its 4 MiB L2 holds (tools/windowbench.py,
// profiles/r02_windowbench.jsonl).  The lane-a addresses of a read set do not depend on the index set, only on
// (k, t): so they are computed ONCE per resident set and kept sorted by address slice (the "query list"), and a scan
// becomes
//     probe   : slice by slice, the eight XCDs each sweeping their own eighth of the slices: stream the slice's
//               addresses, gather the lane-a word(s) of the group's 1 or 2 chunk filters (both strands sit in one word,
//               psi_a), leave one result byte per address;
//     replay  : piece by piece (1024 consecutive reads): collect the piece's results from every slice into LDS masks,
//               bit = first-hit window, then walk the reference's control flow on the set bits only, probing planes
//               B, C, D on lane-a hits (the sparse replay of search_group_kernel, unchanged in what it decides).
// The outcome is bit-identical to search_reads.h:45-83 applied chunk by chunk (what the tests check); only the order
// in which the filter words are fetched has changed.
//
// Query list of a set (built by tq_count / tq_fill, cached in the commet_readset):
//   slice  = address >> sbits of plane A's psi_a address space         n_slices = 2^(k - sbits)
//   piece  = 1024 consecutive reads                                    n_pieces = ceil(n / 1024)
//   tile (slice, piece) = records [tile_off[slice * n_pieces + piece], tile_off[... + 1])  — slice-major, so a slice's
//   records are contiguous (probe) and a piece finds its share of every slice (replay)
//   qaddr[i] = address within the slice | self-paired flag << 31;   qwho[i] = read within piece | window << 10
#pragma once

#include "kernels.hpp"

namespace commet {

constexpr uint32_t TQ_PIECE = 1024;       // reads per piece = threads of the replay workgroup
constexpr int      TQ_MAX_WIN = 96;       // first-hit windows per read (three mask words)

struct QueryListView {
    const unsigned long long *tile_off;   // n_slices * n_pieces + 1
    const uint32_t *qaddr, *qwho;
    uint32_t n_slices, n_pieces;
    int sbits;
};

// first-hit windows of read r: complete windows ending at q in [k-1, pe], pe = len-1-(t-1)k; f(win = q-(k-1), psi address, self-paired)
template <typename F>
__device__ __forceinline__ void tq_for_each_window(const ReadsView &rv, uint64_t r, int k, int t, F &&f)
{
    uint64_t t0;
    uint32_t len;
    read_extent(rv, r, t0, len);
    const uint32_t *p = rv.planes + 3 * t0;
    const int pe = (int) len - 1 - (t - 1) * k;
    const int sh = 32 - k;
    uint32_t wh = 0, run = 0;
    for (uint32_t w = 0; (int) (w * 32u) <= pe; ++w) {
        const uint32_t hi = p[3 * w], va = p[3 * w + 2];
        const uint32_t nb = (uint32_t) min(32, pe - (int) (w * 32u) + 1);
        for (uint32_t j = 0; j < nb; ++j) {
            wh = (wh >> 1) | (((hi >> j) & 1u) << (k - 1));
            run = ((va >> j) & 1u) ? run + 1 : 0;
            if (run < (uint32_t) k) continue;
            bool selfp;
            const uint32_t addr = psi_a<uint32_t>(__brev(wh) >> sh, k, selfp);
            f(32u * w + j - (uint32_t) (k - 1), addr, selfp);
        }
    }
}

// cnt[slice * n_pieces + piece] = records of the tile.  One workgroup per piece.
__global__ __launch_bounds__(256) void tq_count_kernel(ReadsView rv, int k, int t, int sbits, uint32_t n_slices, uint32_t n_pieces,
                                                       unsigned long long *__restrict__ cnt)
{
    extern __shared__ uint32_t h[];   // n_slices
    for (uint32_t i = threadIdx.x; i < n_slices; i += 256) h[i] = 0;
    __syncthreads();
    const uint64_t r0 = (uint64_t) blockIdx.x * TQ_PIECE;
    for (uint32_t i = threadIdx.x; i < TQ_PIECE; i += 256) {
        const uint64_t r = r0 + i;
        if (r >= rv.n) break;
        tq_for_each_window(rv, r, k, t, [&](uint32_t, uint32_t addr, bool) { atomicAdd(&h[addr >> sbits], 1u); });
    }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < n_slices; s += 256) cnt[(uint64_t) s * n_pieces + blockIdx.x] = h[s];
}

// in-place exclusive scan of a[0 .. n) in three steps: per-block scan (4096 entries per block) + block totals
__global__ __launch_bounds__(1024) void tq_scan_blocks_kernel(unsigned long long *__restrict__ a, uint64_t n, unsigned long long *__restrict__ totals)
{
    __shared__ unsigned long long wsum[16];
    const uint64_t base = (uint64_t) blockIdx.x * 4096 + threadIdx.x * 4ull;
    unsigned long long v[4], s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = base + i < n ? a[base + i] : 0ull;
        s += v[i];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long inc = s;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long x = __shfl_up(inc, o, 64);
        if (lane >= o) inc += x;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned long long wbase = 0, tot = 0;
    for (int i = 0; i < 16; ++i) {
        if (i < wave) wbase += wsum[i];
        tot += wsum[i];
    }
    unsigned long long ex = wbase + inc - s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (base + i < n) a[base + i] = ex;
        ex += v[i];
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void tq_scan_totals_kernel(unsigned long long *__restrict__ totals, uint32_t n_blocks,
                                                              unsigned long long *__restrict__ grand_total)
{
    // n_blocks is small (a few thousand): one workgroup, serial over strips of 1024
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 1024) {
        const uint32_t i = b0 + threadIdx.x;
        const unsigned long long v = i < n_blocks ? totals[i] : 0ull;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        unsigned long long inc = v;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long x = __shfl_up(inc, o, 64);
            if (lane >= o) inc += x;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned long long wbase = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) wbase += wsum[w];
            tot += wsum[w];
        }
        const unsigned long long c = carry;
        if (i < n_blocks) totals[i] = c + wbase + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *grand_total = carry;
}

__global__ __launch_bounds__(1024) void tq_scan_add_kernel(unsigned long long *__restrict__ a, uint64_t n, const unsigned long long *__restrict__ totals,
                                                           const unsigned long long *__restrict__ grand_total)
{
    const uint64_t base = (uint64_t) blockIdx.x * 4096 + threadIdx.x * 4ull;
    const unsigned long long add = totals[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (base + i < n) a[base + i] += add;
    if (blockIdx.x == 0 && threadIdx.x == 0) a[n] = *grand_total;   // closes the table
}

// writes the records of piece blockIdx.x into its tiles (tile_off = the scanned counts)
__global__ __launch_bounds__(256) void tq_fill_kernel(ReadsView rv, int k, int t, int sbits, uint32_t n_slices, uint32_t n_pieces,
                                                      const unsigned long long *__restrict__ tile_off, uint32_t *__restrict__ qaddr,
                                                      uint32_t *__restrict__ qwho)
{
    extern __shared__ uint32_t cur[];   // n_slices: records of the tile written so far
    for (uint32_t i = threadIdx.x; i < n_slices; i += 256) cur[i] = 0;
    __syncthreads();
    const uint64_t r0 = (uint64_t) blockIdx.x * TQ_PIECE;
    const uint32_t smask = (1u << sbits) - 1u;
    for (uint32_t i = threadIdx.x; i < TQ_PIECE; i += 256) {
        const uint64_t r = r0 + i;
        if (r >= rv.n) break;
        tq_for_each_window(rv, r, k, t, [&](uint32_t win, uint32_t addr, bool selfp) {
            const uint32_t s = addr >> sbits;
            const unsigned long long at = tile_off[(uint64_t) s * n_pieces + blockIdx.x] + atomicAdd(&cur[s], 1u);
            qaddr[at] = (addr & smask) | (selfp ? 0x80000000u : 0u);
            qwho[at] = i | (win << 10);
        });
    }
}

// probe: result byte of record i = lane-a bits of its window, bit c = chunk c forward strand, bit GS + c = reverse strand.
// Persistent grid of 8 * WPX workgroups: workgroup b belongs to XCD b % 8 and sweeps slices [x * S / 8, (x + 1) * S / 8)
// together with the other workgroups of that XCD.  planes_a: the group's A planes, word-interleaved with stride GS.
template <int GS>
__global__ __launch_bounds__(256) void tq_probe_kernel(QueryListView ql, const uint32_t *__restrict__ planes_a, uint8_t *__restrict__ qres)
{
    const uint32_t x = blockIdx.x % 8, j = blockIdx.x / 8, wpx = gridDim.x / 8;
    const uint32_t s_lo = (uint32_t) ((uint64_t) x * ql.n_slices / 8), s_hi = (uint32_t) ((uint64_t) (x + 1) * ql.n_slices / 8);
    for (uint32_t s = s_lo; s < s_hi; ++s) {
        const unsigned long long a = ql.tile_off[(uint64_t) s * ql.n_pieces], e = ql.tile_off[(uint64_t) (s + 1) * ql.n_pieces];
        const uint32_t *base = planes_a + (((uint64_t) s << ql.sbits) >> 5) * GS;
        for (unsigned long long i = a + (unsigned long long) j * 256 + threadIdx.x; i < e; i += (unsigned long long) wpx * 256) {
            const uint32_t q = ql.qaddr[i];
            const uint32_t addr = q & 0x7FFFFFFFu, bit = addr & 31u;
            const bool selfp = q >> 31;
            uint32_t res = 0;
            if constexpr (GS == 1) {
                const uint32_t wd = base[addr >> 5];
                const uint32_t fb = (wd >> bit) & 1u;
                res = fb | ((selfp ? fb : (wd >> (bit ^ 1u)) & 1u) << 1);
            } else {
                const uint2 wd = *(const uint2 *) (base + (uint64_t) (addr >> 5) * 2);
                const uint32_t f0 = (wd.x >> bit) & 1u, f1 = (wd.y >> bit) & 1u;
                const uint32_t r0 = selfp ? f0 : (wd.x >> (bit ^ 1u)) & 1u, r1 = selfp ? f1 : (wd.y >> (bit ^ 1u)) & 1u;
                res = f0 | (f1 << 1) | (r0 << 2) | (r1 << 3);
            }
            qres[i] = (uint8_t) res;
        }
    }
}

// replay: one workgroup per piece, one thread per read.
template <int GS, int MW>
__global__ __launch_bounds__(TQ_PIECE) void tq_replay_kernel(ReadsView rv, QueryListView ql, const uint8_t *__restrict__ qres,
                                                             FilterGroupView fg, int k, int t, const uint64_t *__restrict__ sel,
                                                             uint64_t *__restrict__ tags, unsigned long long *__restrict__ counters,
                                                             uint32_t cstride)
{
    __shared__ uint32_t masks[GS * 2 * MW * TQ_PIECE];   // [chunk][strand][word][read]
    auto mask_at = [&](int c, int strand, int h, uint32_t rd) -> uint32_t & { return masks[(((c * 2 + strand) * MW) + h) * TQ_PIECE + rd]; };
    for (uint32_t i = threadIdx.x; i < GS * 2 * MW * TQ_PIECE; i += TQ_PIECE) masks[i] = 0;
    __syncthreads();
    // (1) the piece's results, slice by slice: wave w takes slices w, w + 16, ...
    {
        const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (uint32_t s = wave; s < ql.n_slices; s += TQ_PIECE / 64) {
            const unsigned long long a = ql.tile_off[(uint64_t) s * ql.n_pieces + blockIdx.x], e = ql.tile_off[(uint64_t) s * ql.n_pieces + blockIdx.x + 1];
            for (unsigned long long i = a + lane; i < e; i += 64) {
                const uint32_t res = qres[i];
                if (!res) continue;
                const uint32_t who = ql.qwho[i], rd = who & 1023u, win = who >> 10;
#pragma unroll
                for (int c = 0; c < GS; ++c) {
                    if ((res >> c) & 1u) atomicOr(&mask_at(c, 0, (int) (win >> 5), rd), 1u << (win & 31u));
                    if ((res >> (GS + c)) & 1u) atomicOr(&mask_at(c, 1, (int) (win >> 5), rd), 1u << (win & 31u));
                }
            }
        }
    }
    __syncthreads();
    // (2) the sparse replay of search_group8_kernel on the LDS masks (search_reads.h:45-83 on the set bits only)
    const uint64_t r = (uint64_t) blockIdx.x * TQ_PIECE + threadIdx.x;
    const uint64_t word = r >> 6;
    const int lane = threadIdx.x & 63;
    const bool in_range = (word << 6) < rv.n;
    uint64_t selw = ~0ull, tagw = 0;
    if (in_range) {
        if (sel) selw = sel[word];
        if (tags) tagw = tags[word];
    }
    const bool active = (r < rv.n) && ((selw >> lane) & 1ull) && !((tagw >> lane) & 1ull);
    bool found = false;
    int found_chunk = -1;
    if (active) {
        uint64_t t0;
        uint32_t len;
        read_extent(rv, r, t0, len);
        const uint32_t *p = rv.planes + 3 * t0;
        const int sh = 32 - k;
        const uint32_t mask = k == 32 ? ~0u : ((1u << k) - 1u);
        const int last = (int) len - 1;
        const int pe = last - (t - 1) * k;
        const int q0 = k - 1;
        for (int i = 0; i < fg.g && !found; ++i) {
            const uint32_t *pb = fg.slot0 + (uint64_t) i * fg.slot_words + fg.plane_words;
            const uint32_t *pc = pb + fg.plane_words;
            const uint32_t *pd = pc + fg.plane_words;
            for (int strand = 0; strand < 2 && !found; ++strand) {
                int seen = 0, next_ok = 0;
                bool dead = false;
                auto probe_bcd = [&](uint32_t wh, uint32_t wl) -> bool {
                    uint32_t ka, kb;
                    if (strand == 0) ka = __brev(wh) >> sh, kb = __brev(wl) >> sh;
                    else ka = ~wh & mask, kb = ~wl & mask;
                    return test_bit<uint32_t>(pb, kb) && test_bit<uint32_t>(pc, ka ^ kb) && test_bit<uint32_t>(pd, ka | kb);
                };
                for (int h = 0; h < MW && !found && !dead; ++h) {
                    uint32_t m = mask_at(i, strand, h, threadIdx.x);
                    while (m && !found && !dead) {
                        const uint32_t jj = (uint32_t) __ffs((int) m) - 1u;
                        m &= m - 1u;
                        const int q = q0 + 32 * h + (int) jj;
                        if (q < next_ok) continue;
                        if (q + (t - seen - 1) * k > last) {
                            dead = true;
                            break;
                        }
                        ItemWords<uint32_t> it;
                        it.load(p, (uint32_t) q >> 5);
                        uint32_t wh, wl;
                        (void) it.window((uint32_t) q & 31u, k, mask, wh, wl);   // valid: only complete windows are in the list
                        if (probe_bcd(wh, wl)) {
                            ++seen;
                            next_ok = q + k;
                            if (seen >= t) found = true;
                        }
                    }
                }
                if (!found && !dead && seen >= 1) {   // windows behind the first-hit ones, after a first full hit only
                    for (int q = max(pe + 1, next_ok); q <= last && !found; ++q) {
                        if (q + (t - seen - 1) * k > last) break;
                        ItemWords<uint32_t> it;
                        it.load(p, (uint32_t) q >> 5);
                        uint32_t wh, wl;
                        if (!it.window((uint32_t) q & 31u, k, mask, wh, wl)) continue;
                        const uint32_t ka = strand == 0 ? (__brev(wh) >> sh) : (~wh & mask);
                        const uint32_t addr = psi_a<uint32_t>(ka, k);
                        if (!((fg.il_a[(uint64_t) (addr >> 5) * GS + i] >> (addr & 31u)) & 1u)) continue;
                        if (probe_bcd(wh, wl)) {
                            ++seen;
                            q += k - 1;
                            if (seen >= t) found = true;
                        }
                    }
                }
            }
            if (found) found_chunk = i;
        }
    }
    const uint64_t fb = __ballot(found);
    if (lane == 0 && in_range && tags) tags[word] = tagw | fb;
    if (counters) {
        for (int i = 0; i < fg.g; ++i) {
            const uint64_t sc = __ballot(active && (found_chunk < 0 || found_chunk >= i));
            const uint64_t fd = __ballot(found_chunk == i);
            if (lane == 0) {
                if (sc) atomicAdd(&counters[(uint64_t) i * cstride + 0], (unsigned long long) __popcll(sc));
                if (fd) atomicAdd(&counters[(uint64_t) i * cstride + 1], (unsigned long long) __popcll(fd));
            }
        }
    }
}

}  // namespace commet
