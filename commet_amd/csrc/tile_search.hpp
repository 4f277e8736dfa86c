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
//   qaddr[i] = address within the slice | self-paired flag << 31;   qwho[i] = read within piece | window << 8 (16 bits)
#pragma once

#include "kernels.hpp"
#include "index_part.hpp"   // block_scan

#ifndef COMMET_TQ_ABLATE
#define COMMET_TQ_ABLATE 0   // timing ablations exist only in builds made with -DCOMMET_TQ_ABLATE=<mask> (512: no gather, 1024: no replay, 262144 / 524288: plane-B probes / tail gathers folded into an L2-resident window)
#endif

namespace commet {

#ifndef TQ_SWEEP_U
#define TQ_SWEEP_U 2     // replay, step (2): candidates per thread and round of the balanced sweep (1 / 2 / 4: 4.52 / 4.54 / 4.59 ms)
#endif
#ifndef TQ_COLLECT_U
#define TQ_COLLECT_U 8   // replay, step (1): flat records per lane whose loads are in flight together (1 / 2 / 4 / 8: 5.20 / 4.82 / 4.68 / 4.63 ms)
#endif
#ifndef COMMET_TQ_PIECE
#define COMMET_TQ_PIECE 256
#endif
#ifndef TQ_PASS_LIST
#define TQ_PASS_LIST 1   // replay: full hits of the light scans posted as a list and ORed into the masks array behind the sweep (0: a second mask array)
#endif
#ifndef TQ_HIT_CAP
#define TQ_HIT_CAP 1024  // ... of at most this many hits per piece; more: every scan of the piece walks its own candidates
#endif
constexpr uint32_t TQ_PIECE = COMMET_TQ_PIECE;   // reads per piece = threads of the replay workgroup (1024: 9.5 ms, 512: 7.1, 256: 6.8 on configs[1] in round 2; round 5: 256 / 128 / 64: 4.54 / 4.87 / 5.51)
constexpr uint32_t TQ_MAX_LEN = 8000;     // reads of a set that takes the tiled search are shorter: 256 x (len / 32 + 2) triples of a piece < 2^16 (rd_ext)
constexpr int      TQ_MAX_WIN = 255;      // first-hit windows per read (up to eight mask words; qwho has eight bits for the window, and a tile — one slice's share
                                          // of a piece of 256 reads, all of it when the reads are poly-A — must stay below 2^16 records: tlen is 16 bits)

struct QueryListView {
    const unsigned long long *tile_off;   // n_slices * n_pieces + 1
    const uint32_t *qaddr;
    const uint16_t *qwho;                 // read within piece (8 bits) | window << 8 (8 bits)
    const uint32_t *tstart;               // piece-major copy of the tile bounds for the replay: tstart[piece * n_slices + slice]
    const uint16_t *tlen;                 //   = first record / number of records of tile (slice, piece)
    uint32_t n_slices, n_pieces;
    int sbits;
};

// first-hit windows of read r: complete windows ending at q in [k-1, pe], pe = len-1-(t-1)k; f(win = q-(k-1), psi address, self-paired)
// (W = uint32_t for k <= 32, uint64_t above: the address then has up to 34 bits, its slice and its place in the slice
// still fit 32 bits each)
template <typename W, typename F>
__device__ __forceinline__ void tq_for_each_window(const ReadsView &rv, uint64_t r, int k, int t, F &&f)
{
    using T = KeyTraits<W>;
    uint64_t t0;
    uint32_t len;
    read_extent(rv, r, t0, len);
    const uint32_t *p = rv.planes + 3 * t0;
    const int pe = (int) len - 1 - (t - 1) * k;
    const int sh = T::BITS - k;
    W wh = 0;
    uint32_t run = 0;
    for (uint32_t w = 0; (int) (w * 32u) <= pe; ++w) {
        const uint32_t hi = p[3 * w], va = p[3 * w + 2];
        const uint32_t nb = (uint32_t) min(32, pe - (int) (w * 32u) + 1);
        for (uint32_t j = 0; j < nb; ++j) {
            wh = (wh >> 1) | ((W) ((hi >> j) & 1u) << (k - 1));
            run = ((va >> j) & 1u) ? run + 1 : 0;
            if (run < (uint32_t) k) continue;
            bool selfp;
            const W addr = psi_a<W>(T::brev(wh) >> sh, k, selfp);
            f(32u * w + j - (uint32_t) (k - 1), addr, selfp);
        }
    }
}

// cnt[slice * n_pieces + piece] = records of the tile.  One workgroup per piece.
template <typename W>
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
        tq_for_each_window<W>(rv, r, k, t, [&](uint32_t, W addr, bool) { atomicAdd(&h[(uint32_t) (addr >> sbits)], 1u); });
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

// ---------------------------------------------------------------------------
// Reads of a RAGGED set in order of their first-hit window counts (round 6).  The gather kernels give a read to a thread and walk its
// windows one round trip at a time: a workgroup lives as long as its longest read, its other lanes' request slots idle meanwhile — on
// a set of 50-150-bp reads a pass cost 73 ps per window and read against 50 on 100-bp reads (tools/group_bench.py), and no dealing
// of reads INSIDE a workgroup helps (measured: its registers stay allocated until its last wave is through).  With the reads listed
// in 32 classes of window counts (most first; inside a class in read order), a launch in list form (ActiveList) hands every
// workgroup reads of one class.  A counting sort: per-block class counts, a scan in class-major order (the scan kernels above), the
// fill.  ids[n] = n closes the list (ActiveList::n).  Depends on (k, t) and the set only: cached with the set.
// ---------------------------------------------------------------------------
constexpr uint32_t LO_BLOCK = 1024, LO_CLASSES = 32;
__device__ __forceinline__ uint32_t lo_class(const ReadsView &rv, uint64_t r, uint32_t tk)
{
    const uint64_t len = rv.goff[r + 1] - rv.goff[r];
    const uint64_t wins = len >= tk ? len - tk + 1 : 0;
    return LO_CLASSES - 1u - (uint32_t) min(wins / 8, (uint64_t) (LO_CLASSES - 1));
}

__global__ __launch_bounds__(256) void lo_count_kernel(ReadsView rv, uint32_t tk, uint64_t n_blocks, unsigned long long *__restrict__ cnt)
{
    __shared__ uint32_t h[LO_CLASSES];
    if (threadIdx.x < LO_CLASSES) h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t r0 = (uint64_t) blockIdx.x * LO_BLOCK;
    for (uint32_t i = threadIdx.x; i < LO_BLOCK; i += 256)
        if (r0 + i < rv.n) atomicAdd(&h[lo_class(rv, r0 + i, tk)], 1u);
    __syncthreads();
    if (threadIdx.x < LO_CLASSES) cnt[(uint64_t) threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(256) void lo_fill_kernel(ReadsView rv, uint32_t tk, uint64_t n_blocks, const unsigned long long *__restrict__ off,
                                                      uint32_t *__restrict__ ids)
{
    __shared__ uint32_t h[LO_CLASSES];
    __shared__ unsigned long long base[LO_CLASSES];
    if (threadIdx.x < LO_CLASSES) {
        h[threadIdx.x] = 0;
        base[threadIdx.x] = off[(uint64_t) threadIdx.x * n_blocks + blockIdx.x];
    }
    __syncthreads();
    const uint64_t r0 = (uint64_t) blockIdx.x * LO_BLOCK;
    // (thread x takes reads x, x + 256, ...: inside a class the block's reads come out in four interleaved runs — order inside a class
    // decides nothing)
    for (uint32_t i = threadIdx.x; i < LO_BLOCK; i += 256) {
        const uint64_t r = r0 + i;
        if (r >= rv.n) break;
        const uint32_t c = lo_class(rv, r, tk);
        ids[base[c] + atomicAdd(&h[c], 1u)] = (uint32_t) r;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ids[rv.n] = (uint32_t) rv.n;
}

// piece-major copy of the tile bounds (the replay reads all slices of ONE piece: 256 strided 8-byte loads become two short
// contiguous arrays).  Records < 2^32 and tiles < 2^16 records are guaranteed by the host (list <= 4 GiB, 256 reads x 255 windows).
__global__ __launch_bounds__(256) void tq_bounds_kernel(const unsigned long long *__restrict__ tile_off, uint32_t n_slices, uint32_t n_pieces,
                                                        uint32_t *__restrict__ tstart, uint16_t *__restrict__ tlen)
{
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;          // = piece * n_slices + slice
    if (i >= (uint64_t) n_slices * n_pieces) return;
    const uint64_t p = i / n_slices, s = i % n_slices;
    const unsigned long long a = tile_off[s * n_pieces + p], e = tile_off[s * n_pieces + p + 1];
    tstart[i] = (uint32_t) a;
    tlen[i] = (uint16_t) (e - a);
}

// writes the records of piece blockIdx.x into its tiles.  The records of `rpr` reads at a time are sorted by slice in LDS
// (counting sort) and leave as ONE flat sequence: thread j takes sorted records j, j + 256, ...; a record carries its slice, the
// slice's place in the list (tstart, the piece-major copy of the tile bounds: one coalesced load per workgroup) and what earlier
// rounds wrote sit in LDS.  History: written record by record straight from the window loop, each 4-byte store reached HBM on its
// own (18.7 GB written for 3 GB of records, 11 ms per 10 M-read set); sorted in LDS but written slice by slice — a wave per slice,
// each with a dependent 8-byte load of its tile's start and ~24 of 64 lanes busy — 5.6 ms; flat, two rounds per piece 2.5 ms.
#ifndef COMMET_TQ_FILL_CAP
#define COMMET_TQ_FILL_CAP 9600   // a whole piece of 100-bp reads at k = 32, t = 2 (256 x 37 records) in ONE round: 1.94 ms per 10 M-read set against 2.59 with 6144 (two rounds of 128 reads) and 2.58 with 4800; 81 KB of LDS, two workgroups per CU
#endif
constexpr uint32_t TQ_FILL_CAP = COMMET_TQ_FILL_CAP;    // records sorted per round: rpr * (first-hit windows per read) <= TQ_FILL_CAP
template <typename W>
__global__ __launch_bounds__(256) void tq_fill_kernel(ReadsView rv, int k, int t, int sbits, uint32_t n_slices, uint32_t n_pieces,
                                                      uint32_t rpr, const uint32_t *__restrict__ tstart,
                                                      uint32_t *__restrict__ qaddr, uint16_t *__restrict__ qwho)
{
    static_assert(TQ_PIECE <= 256 && TQ_MAX_WIN <= 256, "qwho packs the read in 8 bits and the window in 8");
    extern __shared__ uint32_t fl[];
    uint32_t *cnt = fl, *base = cnt + n_slices, *fill = base + n_slices, *dstb = fill + n_slices;   // n_slices each
    uint32_t *rec_a = dstb + n_slices, *rec_w = rec_a + TQ_FILL_CAP;
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t pw[TQ_PIECE + 1];      // exclusive prefix of the reads' first-hit windows (len - t k + 1: an upper bound on their records)
    // where this piece's share of every slice starts (n_slices <= 1024 consecutive words)
    for (uint32_t i = threadIdx.x; i < n_slices; i += 256) dstb[i] = tstart[(uint64_t) blockIdx.x * n_slices + i];
    const uint64_t r0 = (uint64_t) blockIdx.x * TQ_PIECE;
    const uint32_t smask = (1u << sbits) - 1u;
    // Rounds: as many consecutive reads as keep their first-hit windows within TQ_FILL_CAP records, by the reads' OWN lengths (round 6:
    // cut by the set's longest read, a piece of 50-150-bp reads took three rounds where its 9.7 k records need two, one for 100-bp reads);
    // at most rpr reads (the host's bound for reads of the longest length: the same cut as before on sets of one read length)
    {
        uint32_t w = 0;
        if (threadIdx.x < TQ_PIECE && r0 + threadIdx.x < rv.n) {
            uint64_t t0;
            uint32_t len;
            read_extent(rv, r0 + threadIdx.x, t0, len);
            const uint32_t tk = (uint32_t) t * (uint32_t) k;
            w = len >= tk ? len - tk + 1 : 0;
        }
        uint32_t tot;
        const uint32_t ex = block_scan<256>(w, wsum, &tot);
        if (threadIdx.x < TQ_PIECE) pw[threadIdx.x] = ex;
        if (threadIdx.x == 0) pw[TQ_PIECE] = tot;
        __syncthreads();
    }
    (void) rpr;
    for (uint32_t rr = 0, hi = 0; rr < TQ_PIECE; rr = hi) {
        {   // hi = the largest h in (rr, TQ_PIECE] with pw[h] - pw[rr] <= TQ_FILL_CAP (a single read never exceeds the cap: TQ_MAX_WIN windows)
            uint32_t lo = rr + 1, up = TQ_PIECE;
            const uint32_t lim = pw[rr] + TQ_FILL_CAP;
            while (lo < up) {
                const uint32_t mid = (lo + up + 1) >> 1;
                if (pw[mid] <= lim) lo = mid;
                else up = mid - 1;
            }
            hi = lo;
        }
        for (uint32_t i = threadIdx.x; i < n_slices; i += 256) cnt[i] = 0, fill[i] = 0;
        __syncthreads();
        const uint32_t i = rr + threadIdx.x;
        const bool mine = i < hi && r0 + i < rv.n;
        if (mine) tq_for_each_window<W>(rv, r0 + i, k, t, [&](uint32_t, W addr, bool) { atomicAdd(&cnt[(uint32_t) (addr >> sbits)], 1u); });
        __syncthreads();
        lds_scan<256>(cnt, base, n_slices, wsum);
        if (mine)
            tq_for_each_window<W>(rv, r0 + i, k, t, [&](uint32_t win, W addr, bool selfp) {
                const uint32_t s = (uint32_t) (addr >> sbits);
                const uint32_t at = base[s] + atomicAdd(&fill[s], 1u);
                rec_a[at] = ((uint32_t) addr & smask) | (selfp ? 0x80000000u : 0u);
                rec_w[at] = i | (win << 8) | (s << 16);                  // read (8 bits) | window (8) | slice (<= 10)
            });
        __syncthreads();
        const uint32_t total = base[n_slices - 1] + cnt[n_slices - 1];
        for (uint32_t j = threadIdx.x; j < total; j += 256) {
            const uint32_t w = rec_w[j], s = w >> 16;
            const uint32_t dst = dstb[s] + (j - base[s]);
            qaddr[dst] = rec_a[j];
            qwho[dst] = (uint16_t) (w & 0xFFFFu);
        }
        __syncthreads();
        for (uint32_t s = threadIdx.x; s < n_slices; s += 256) dstb[s] += cnt[s];
        // (the zeroing of cnt / fill at the top of the next round touches other arrays; its barrier orders these updates, too)
    }
}

// probe: result byte of record i = lane-a bits of its window, bit c = chunk c forward strand, bit GS + c = reverse strand.
// Persistent grid of 8 * WPX workgroups: workgroup b belongs to XCD b % 8 and sweeps slices [x * S / 8, (x + 1) * S / 8)
// together with the other workgroups of that XCD.  planes_a: the group's A planes, word-interleaved with stride GS.
template <int GS>
__device__ __forceinline__ uint32_t tq_probe_one(const uint32_t *__restrict__ base, uint32_t q)
{
    const uint32_t addr = q & 0x7FFFFFFFu, bit = addr & 31u;
    const bool selfp = q >> 31;
    if constexpr (GS == 1) {
        const uint32_t wd = base[addr >> 5];
        const uint32_t fb = (wd >> bit) & 1u;
        return fb | ((selfp ? fb : (wd >> (bit ^ 1u)) & 1u) << 1);
    } else {
        const uint2 wd = *(const uint2 *) (base + (uint64_t) (addr >> 5) * 2);
        const uint32_t f0 = (wd.x >> bit) & 1u, f1 = (wd.y >> bit) & 1u;
        const uint32_t r0 = selfp ? f0 : (wd.x >> (bit ^ 1u)) & 1u, r1 = selfp ? f1 : (wd.y >> (bit ^ 1u)) & 1u;
        return f0 | (f1 << 1) | (r0 << 2) | (r1 << 3);
    }
}

// [p0, p1): the pieces whose records are probed (a part of the set: the host runs the replay of one part beside the probe
// of the next, capi.hip launch_search_tiled); a slice's records of consecutive pieces are contiguous.
template <int GS>
__global__ __launch_bounds__(256) void tq_probe_kernel(QueryListView ql, const uint32_t *__restrict__ planes_a, uint8_t *__restrict__ qres,
                                                       uint32_t p0, uint32_t p1)
{
    const uint32_t x = blockIdx.x % 8, j = blockIdx.x / 8, wpx = gridDim.x / 8;
    const uint32_t s_lo = (uint32_t) ((uint64_t) x * ql.n_slices / 8), s_hi = (uint32_t) ((uint64_t) (x + 1) * ql.n_slices / 8);
    for (uint32_t s = s_lo; s < s_hi; ++s) {
        unsigned long long a = ql.tile_off[(uint64_t) s * ql.n_pieces + p0];
        const unsigned long long e = ql.tile_off[(uint64_t) s * ql.n_pieces + p1];
        const uint32_t *base = planes_a + (((uint64_t) s << ql.sbits) >> 5) * GS;
        // whole groups of four records [4m, 4m + 4) inside [a, e): one 16-byte load of addresses, four gathers, one 4-byte
        // store of results.  The address and result streams are read / written once: non-temporal, so that they do not
        // push the slice's filter words out of L2.
        const unsigned long long a4 = (a + 3) & ~3ull, e4 = e & ~3ull;
        if (a4 < e4) {
            const unsigned long long step = (unsigned long long) wpx * 256, m_end = e4 / 4;
            unsigned long long m = a4 / 4 + (unsigned long long) j * 256 + threadIdx.x;
            // (the next group's addresses prefetched while this group's gathers are made: 2.59 against 2.37 ms — the extra registers in flight cost more than the overlap gives)
            for (; m < m_end; m += step) {
                const uint32_t *qp = ql.qaddr + 4 * m;
                const uint32_t q0 = __builtin_nontemporal_load(qp), q1 = __builtin_nontemporal_load(qp + 1),
                               q2 = __builtin_nontemporal_load(qp + 2), q3 = __builtin_nontemporal_load(qp + 3);
                const uint32_t r = tq_probe_one<GS>(base, q0) | (tq_probe_one<GS>(base, q1) << 8) | (tq_probe_one<GS>(base, q2) << 16) |
                                   (tq_probe_one<GS>(base, q3) << 24);
                __builtin_nontemporal_store(r, (uint32_t *) (qres + 4 * m));
            }
        }
        // the up to three records before a4 and after e4 (all of the slice when it has no whole group)
        if (j == 0 && threadIdx.x < 8) {
            unsigned long long i;
            bool mine;
            if (a4 < e4) {
                i = threadIdx.x < 4 ? a + threadIdx.x : e4 + (threadIdx.x - 4);
                mine = threadIdx.x < 4 ? i < a4 : i < e;
            } else {                              // fewer than eight records and no aligned group among them
                i = a + threadIdx.x;
                mine = i < e;
            }
            if (mine) qres[i] = (uint8_t) tq_probe_one<GS>(base, ql.qaddr[i]);
        }
    }
}

// replay: one workgroup per piece, one thread per read.
// (scalar registers capped, COMMET_SGPRS in kernels.hpp: eight workgroups per CU instead of six — the LDS (19.3 KiB) and the 50
// vector registers allow eight, and the kernel is a chain of dependent round trips that lives on workgroups in flight)
template <typename W, int GS, int MW>
__global__ __launch_bounds__(TQ_PIECE) COMMET_SGPRS void tq_replay_kernel(ReadsView rv, QueryListView ql, const uint8_t *__restrict__ qres,
                                                             FilterGroupView fg, int k, int t, const uint64_t *__restrict__ sel,
                                                             uint64_t *__restrict__ tags, unsigned long long *__restrict__ counters,
                                                             uint32_t cstride, uint32_t piece0, uint32_t hit_cap, uint64_t job_tag_words)
{
    // job_tag_words != 0 (GS == 2 only; round 6): the two chunk filters belong to TWO jobs that search this set (two J2 jobs of a reference
    // set, two J3 jobs of a target whose index selections make one chunk each, commet_index_many_and_search): the probe's one gather per
    // record serves both; here job 1 starts afresh at chunk 1 — no read is skipped for what job 0 found — and job j's found flags go to
    // tags + j * job_tag_words (zeroed by the host), its counters to its own chunk's slots.  `tags` is not read then.
    const bool multi = GS == 2 && job_tag_words != 0;
    // hit_cap <= TQ_HIT_CAP: full hits of light scans a piece may post (tests set it to 0: every piece with a hit overflows)
    const uint32_t piece = blockIdx.x + piece0;          // (the launch covers pieces piece0 .. piece0 + gridDim.x - 1)
    __shared__ uint32_t masks[GS * 2 * MW * TQ_PIECE];   // [chunk][strand][word][read]
    auto mask_at = [&](int c, int strand, int h, uint32_t rd) -> uint32_t & { return masks[(((c * 2 + strand) * MW) + h) * TQ_PIECE + rd]; };
    for (uint32_t i = threadIdx.x; i < GS * 2 * MW * TQ_PIECE; i += TQ_PIECE) masks[i] = 0;
    __syncthreads();
#if !(COMMET_TQ_ABLATE & 512)
    // (1) the piece's results.  Wave w takes slices w, w + 16, ... in batches of 64 tiles: every lane fetches the bounds of
    // one tile, the batch's records then form one flat list that the wave walks 64 at a time (tile of a record: by
    // counting the tile ends at or below it) — independent loads, no chain of small dependent ones.
    {
        const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        constexpr uint32_t NWAVE = TQ_PIECE / 64;
        for (uint32_t s0 = wave; s0 < ql.n_slices; s0 += 64 * NWAVE) {
            const uint32_t sl = s0 + lane * NWAVE;                       // this lane's tile of the batch
            unsigned long long ta = 0, te = 0;
            if (sl < ql.n_slices) {
                ta = ql.tstart[(uint64_t) piece * ql.n_slices + sl];           // piece-major: a piece's bounds are contiguous
                te = ta + ql.tlen[(uint64_t) piece * ql.n_slices + sl];
            }
            const uint32_t len = (uint32_t) (te - ta);
            uint32_t inc = len;                                           // inclusive prefix of the tile lengths over the lanes
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t x = __shfl_up(inc, o, 64);
                if ((int) lane >= o) inc += x;
            }
            const uint32_t total = __shfl(inc, 63, 64);
            // TQ_COLLECT_U flat records per lane and round: their result bytes and owner words are independent loads, all issued
            // before the first is looked at — one round trip per round instead of two per record (the loop used to wait for a
            // record's result byte, then, for the one in four that is not zero, for its owner word).  The owner words of the
            // zero results cost no memory traffic: their sectors are fetched for their neighbours anyway.
            for (uint32_t f0 = 0; f0 < total; f0 += 64 * TQ_COLLECT_U) {              // (uniform trip count: every lane takes part in the shuffles)
                unsigned long long ri[TQ_COLLECT_U];
                bool ok[TQ_COLLECT_U];
#pragma unroll
                for (int u = 0; u < TQ_COLLECT_U; ++u) {
                    const uint32_t f = f0 + 64u * (uint32_t) u + lane;
                    // tile of flat record f = number of lanes whose inclusive prefix is <= f (binary search over the 64 prefixes)
                    uint32_t lo = 0, hi = 64;
#pragma unroll
                    for (int step = 0; step < 7; ++step) {
                        const uint32_t mid = min((lo + hi) >> 1, 63u);
                        const uint32_t pm = __shfl(inc, (int) mid, 64);
                        if (lo < hi) {
                            if (pm <= f) lo = mid + 1;
                            else hi = mid;
                        }
                    }
                    const uint32_t src = min(lo, 63u);
                    const uint32_t before_raw = __shfl(inc, (int) (src ? src - 1 : 0), 64);
                    const unsigned long long tbase = __shfl((unsigned long long) ta, (int) src, 64);
                    ok[u] = f < total;
                    ri[u] = ok[u] ? tbase + (f - (src ? before_raw : 0u)) : 0ull;     // (record 0 exists whenever the batch has one)
                }
                uint32_t res[TQ_COLLECT_U], who[TQ_COLLECT_U];
#pragma unroll
                for (int u = 0; u < TQ_COLLECT_U; ++u) {
                    res[u] = __builtin_nontemporal_load(qres + ri[u]);
                    who[u] = __builtin_nontemporal_load(ql.qwho + ri[u]);
                }
#pragma unroll
                for (int u = 0; u < TQ_COLLECT_U; ++u) {
                    if (!ok[u] || !res[u]) continue;
                    const uint32_t rd = who[u] & 255u, win = who[u] >> 8;
#pragma unroll
                    for (int c = 0; c < GS; ++c) {
                        if ((res[u] >> c) & 1u) atomicOr(&mask_at(c, 0, (int) (win >> 5), rd), 1u << (win & 31u));
                        if ((res[u] >> (GS + c)) & 1u) atomicOr(&mask_at(c, 1, (int) (win >> 5), rd), 1u << (win & 31u));
                    }
                }
            }
        }
    }
#endif
    __syncthreads();
    // (2) planes B, C, D for the lane-a candidates, balanced over the workgroup.  A read has ~11 candidates on average but
    // the count varies from lane to lane, and every probe is a dependent HBM round trip; walking them per read leaves
    // most lanes idle in most of ~150 serialised loads per wave.  So the candidates of the whole piece are numbered
    // (prefix sums of the masks' popcounts) and thread f takes candidates f, f + 1024, ...: one plane-B probe per lane and
    // round, all lanes busy.  Survivors (A & B) are collected in a second mask array and go through planes C and D the same
    // way; what is left are the full four-lane hits.
    // The full hits (A & B & C & D) of the light scans are found by whichever thread the sweep hands the candidate to, and are few (chance
    // hits; reads that share sequence are heavy scans and walk their own candidates).  PASS_LIST: they are posted as a list of
    // TQ_HIT_CAP words and, behind the sweep, ORed into the masks array itself, which nobody reads any more by then — the second
    // mask array (12 KiB with three mask words: five workgroups per CU instead of eight, 6.4 against 4.5 ms per configs[1]-sized
    // step on 50-150-bp reads) is gone.  A piece with more hits than the list holds lets every scan walk its own candidates, as
    // heavy scans do: exact, only slower.
    constexpr bool PASS_LIST = TQ_PASS_LIST != 0;
    __shared__ uint32_t pass_arr[PASS_LIST ? 1 : GS * 2 * MW * TQ_PIECE];
    __shared__ uint32_t hits[PASS_LIST ? TQ_HIT_CAP : 1];
    __shared__ uint32_t hit_n;
    uint32_t *const pass = PASS_LIST ? masks : pass_arr;
    // where every read of the piece lies and how long it is: (first triple - the piece's first triple) | length << 16.  A candidate is
    // probed by whichever thread the sweep hands it to, a tail window by whichever thread its number falls on, and on a set of many read
    // lengths the owner's extent is two more loads (goff) in front of every such probe's chain; the piece's 256 extents fit 1 KiB
    // (reads of a set that takes the tiled search have fewer than 2^13 bases: TQ_MAX_WIN first-hit windows)
    __shared__ uint32_t rd_ext[TQ_PIECE];
    __shared__ unsigned long long piece_t0;
    __shared__ uint32_t pre[TQ_PIECE];
    __shared__ uint32_t scan_ws[TQ_PIECE / 64];
    __shared__ unsigned int wg_cnt[2 * GS];
    constexpr int NS = 2 * GS;                           // scan index = chunk * 2 + strand
    auto word_at = [&](uint32_t *arr, int i, int h, uint32_t rd) -> uint32_t & { return arr[((i * MW) + h) * TQ_PIECE + rd]; };
    if (threadIdx.x < 2 * GS) wg_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) hit_n = 0;
    const uint64_t r = (uint64_t) piece * TQ_PIECE + threadIdx.x;
    const uint64_t word = r >> 6;
    const int lane = threadIdx.x & 63;
    const bool in_range = (word << 6) < rv.n;
    uint64_t selw = ~0ull, tagw = 0;
    if (in_range) {
        if (sel) selw = sel[word];
        if (tags && !multi) tagw = tags[word];
    }
    const bool active = (r < rv.n) && ((selw >> lane) & 1ull) && !((tagw >> lane) & 1ull);
    uint64_t my_t0 = 0;
    uint32_t my_len = 0;
    if (r < rv.n) read_extent(rv, r, my_t0, my_len);
    if (threadIdx.x == 0) piece_t0 = my_t0;                          // (the piece's first read exists: the grid covers pieces of the set only)
    using T = KeyTraits<W>;
    const int sh = T::BITS - k;
    const W kmask = (k == T::BITS) ? ~(W) 0 : (((W) 1 << k) - 1);
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int h = 0; h < MW; ++h) {
            if (!active || (i >> 1) >= fg.g) word_at(masks, i, h, threadIdx.x) = 0;   // reads that are not searched have no candidates
            if (!PASS_LIST) word_at(pass, i, h, threadIdx.x) = 0;
        }
    __syncthreads();
    rd_ext[threadIdx.x] = (r < rv.n) ? ((uint32_t) (my_t0 - piece_t0) & 0xFFFFu) | (my_len << 16) : 0u;   // (first read before the sweep's first barrier)
    // A read that shares sequence with the index set has a lane-a bit on (nearly) every window of one strand; the
    // reference leaves it after t hits, i.e. after ~4 t probes.  Such "heavy" scans (more than TQ_HEAVY candidates) keep
    // their masks in registers and their thread walks them itself in step (3), stopping at t; probing all their windows
    // in the balanced sweep would multiply their probes by ten.  The read's other scans (the other strand, the other
    // chunk: a handful of chance candidates) stay in the sweep like anybody's.
    constexpr uint32_t TQ_HEAVY = 20;   // 4 / 12 / 18 / 24 / 32 / 40 -> 8.2 / 6.9 / 6.3 / 6.3 / 6.6 / 8.9 ms on configs[1]
    uint32_t am[NS][MW];
    uint32_t hv = 0;                    // bit i: scan i is heavy
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        uint32_t n = 0;
#pragma unroll
        for (int h = 0; h < MW; ++h) am[i][h] = word_at(masks, i, h, threadIdx.x), n += __popc(am[i][h]);
        if (n > TQ_HEAVY) {
            hv |= 1u << i;
#pragma unroll
            for (int h = 0; h < MW; ++h) word_at(masks, i, h, threadIdx.x) = 0;   // (the sweep's first barrier comes before anyone else reads them)
        }
    }
    const bool heavy = hv != 0;
    // words of another read of the piece around window end q
    const uint32_t *const piece_planes = rv.planes + 3 * piece_t0;
    auto keys_of = [&](uint32_t owner, int strand, int q, W &ka, W &kb) {
        ItemWords<W> it;
        it.load(piece_planes + 3u * (rd_ext[owner] & 0xFFFFu), (uint32_t) q >> 5);
        W wh, wl;
        (void) it.window((uint32_t) q & 31u, k, kmask, wh, wl);           // complete: only complete windows are in the list
        if (strand) ka = ~wh & kmask, kb = ~wl & kmask;
        else ka = T::brev(wh) >> sh, kb = T::brev(wl) >> sh;
    };
    // one balanced sweep over the set bits of `src` (heavy scans are not in it), four candidates per thread and round so that
    // four probes are in flight per lane: word_of(owner, scan, window end) -> {filter word address, bit} of the first plane
    // to test; on_set(owner, scan, mask word, bit, ka, kb) is called for the candidates whose bit is set
    auto sweep = [&](uint32_t *src, auto &&word_of, auto &&on_set) {
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int h = 0; h < MW; ++h) cnt += __popc(word_at(src, i, h, threadIdx.x));
        uint32_t total;
        const uint32_t ex = block_scan<TQ_PIECE>(cnt, scan_ws, &total);
        pre[threadIdx.x] = ex + cnt;                      // inclusive
        __syncthreads();
        constexpr int U = TQ_SWEEP_U;
        for (uint32_t f0 = threadIdx.x; f0 < total; f0 += U * TQ_PIECE) {
            uint32_t owner[U], bitn[U], fw[U], fbit[U];
            W ka[U], kb[U];
            int sc[U], hw[U];
            bool have[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t f = f0 + (uint32_t) u * TQ_PIECE;
                have[u] = f < total;
                owner[u] = 0, bitn[u] = 0, sc[u] = 0, hw[u] = 0;
                if (!have[u]) continue;
                uint32_t lo = 0, hi = TQ_PIECE - 1;       // first read whose inclusive prefix exceeds f
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (pre[mid] <= f) lo = mid + 1;
                    else hi = mid;
                }
                uint32_t local = f - (lo ? pre[lo - 1] : 0u), wd = 0;
                bool got = false;
#pragma unroll
                for (int i = 0; i < NS; ++i)
#pragma unroll
                    for (int h = 0; h < MW; ++h) {
                        if (got) continue;
                        const uint32_t x = word_at(src, i, h, lo);
                        const uint32_t c = __popc(x);
                        if (local < c) sc[u] = i, hw[u] = h, wd = x, got = true;
                        else local -= c;
                    }
                for (uint32_t v = 0; v < local; ++v) wd &= wd - 1u;        // drop the `local` lowest set bits
                owner[u] = lo, bitn[u] = (uint32_t) __ffs((int) wd) - 1u;
            }
            const uint32_t *wp[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                                  // keys: the owners' read words (L1 / L2)
                ka[u] = kb[u] = 0;
                if (have[u]) keys_of(owner[u], sc[u] & 1, (int) (32 * hw[u] + bitn[u]) + (k - 1), ka[u], kb[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {                                  // the filter words: up to four HBM probes in flight
                fw[u] = 0, fbit[u] = 0;
                if (have[u]) {
                    wp[u] = word_of(sc[u], ka[u], kb[u], fbit[u]);
                    fw[u] = *wp[u];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (have[u] && ((fw[u] >> fbit[u]) & 1u)) on_set(owner[u], sc[u], hw[u], bitn[u], ka[u], kb[u]);
        }
        __syncthreads();
    };
    if (!(COMMET_TQ_ABLATE & 1024)) {
        // plane B of every candidate; the one in nine that passes goes through planes C and D at once (two independent
        // loads).  A second balanced sweep for C / D (1.2 K survivors per piece) cost more in prefix sums and barriers
        // than the divergence it avoided.
        sweep(masks,
              [&](int i, W, W kb, uint32_t &bit) -> const uint32_t * {
                  bit = (uint32_t) kb & 31u;
                  if (COMMET_TQ_ABLATE & 262144) return fg.slot0 + (uint64_t) (i >> 1) * fg.slot_words + fg.plane_words + ((kb >> 5) & 0x1FFFFu);   // timing bound: plane B served from L2
                  return fg.slot0 + (uint64_t) (i >> 1) * fg.slot_words + fg.plane_words + (kb >> 5);
              },
              [&](uint32_t owner, int i, int h, uint32_t b, W ka, W kb) {
                  const uint32_t *pc = fg.slot0 + (uint64_t) (i >> 1) * fg.slot_words + 2 * fg.plane_words, *pd = pc + fg.plane_words;
                  const uint32_t vc = pc[(ka ^ kb) >> 5], vd = pd[(ka | kb) >> 5];
                  if ((vc >> ((uint32_t) (ka ^ kb) & 31u)) & (vd >> ((uint32_t) (ka | kb) & 31u)) & 1u) {
                      if (PASS_LIST) {
                          const uint32_t at = atomicAdd(&hit_n, 1u);
                          if (at < hit_cap) hits[at] = owner | ((uint32_t) i << 8) | ((uint32_t) h << 12) | (b << 16);
                      } else {
                          atomicOr(&word_at(pass, i, h, owner), 1u << b);
                      }
                  }
              });
    }
    bool all_self = false;              // PASS_LIST: the list overflowed — every scan walks its own lane-a candidates (am[])
    if (PASS_LIST) {                    // (behind the sweep's last barrier: nobody reads the masks any more)
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int h = 0; h < MW; ++h) word_at(masks, i, h, threadIdx.x) = 0;
        __syncthreads();
        const uint32_t nh = hit_n;
        all_self = nh > hit_cap;
        for (uint32_t e = threadIdx.x; e < min(nh, hit_cap); e += TQ_PIECE) {
            const uint32_t x = hits[e];
            atomicOr(&word_at(masks, (int) ((x >> 8) & 15u), (int) ((x >> 12) & 15u), x & 255u), 1u << (x >> 16));
        }
        __syncthreads();
    }
    // (3) the reference's control flow (search_reads.h:45-83) on the full hits of this thread's read: per chunk, strand 0
    // then strand 1; greedy non-overlapping hits; the windows behind the first-hit ones are probed only for a scan that
    // already has a hit (exact pruning, see search_kernel).  Those tails are fetched by the whole workgroup: the threads
    // that need one post (read, first window end) in LDS, then thread p takes window p % 32 of request p / 32 — one
    // round trip and ~50 instructions per window with every lane busy, where a thread fetching its own 32 windows
    // (a rolling window, four batches of eight loads) kept the other lanes of its wave waiting through
    // ~1500 instructions and four round trips (1.8 ms of this kernel's 6.1 on configs[1]).
    // (PASS_LIST: the hit list is dead by now; the tails' request / answer arrays take its place)
    static_assert(!PASS_LIST || TQ_HIT_CAP >= 2 * TQ_PIECE, "the tails' arrays alias the hit list");
    __shared__ uint32_t tail_arr[PASS_LIST ? 1 : 2 * TQ_PIECE];
    uint32_t *const tail_req = PASS_LIST ? hits : tail_arr, *const tail_bits = tail_req + TQ_PIECE;
    __shared__ uint32_t tail_n;
    int found_chunk = -1;
    bool found_job0 = false, found_job1 = false;       // (two jobs in one scan)
    {
        const uint32_t *p = rv.planes + 3 * my_t0;
        const int last = (int) my_len - 1;
        const int pe = last - (t - 1) * k;
        const int q0 = k - 1;
        const bool scanning = active && !(COMMET_TQ_ABLATE & (1024 | 8192));
        bool found = false;
        for (int i = 0; i < 2 * fg.g; ++i) {   // (uniform)
            if (multi && i == 2) found_job0 = found, found = false;    // chunk 1 opens the second job: nothing carries over
            const int strand = i & 1;
            const uint32_t *pb = fg.slot0 + (uint64_t) (i >> 1) * fg.slot_words + fg.plane_words;
            const uint32_t *pc = pb + fg.plane_words, *pd = pc + fg.plane_words;
            int seen = 0, next_ok = 0;
            bool dead = !scanning || found;
            for (int h = 0; h < MW && !found && !dead; ++h) {
                const bool hscan = ((hv >> i) & 1u) || all_self;
                uint32_t m = pass[((i * MW) + h) * TQ_PIECE + threadIdx.x];    // light scans: full hits (step 2)
                if (hscan) {                                                     // heavy scans: lane-a candidates, probed here
                    m = 0;
#pragma unroll
                    for (int ii = 0; ii < NS; ++ii)
#pragma unroll
                        for (int hh = 0; hh < MW; ++hh)
                            if (ii == i && hh == h) m = am[ii][hh];
                }
                while (m) {
                    const int q = q0 + 32 * h + (__ffs((int) m) - 1);
                    m &= m - 1u;
                    if (q < next_ok) continue;
                    if (q + (t - seen - 1) * k > last) {
                        dead = true;
                        break;
                    }
                    if (hscan) {
                        ItemWords<W> it;
                        it.load(p, (uint32_t) q >> 5);
                        W wh, wl, ka, kb;
                        (void) it.window((uint32_t) q & 31u, k, kmask, wh, wl);
                        if (strand) ka = ~wh & kmask, kb = ~wl & kmask;
                        else ka = T::brev(wh) >> sh, kb = T::brev(wl) >> sh;
                        // three independent loads, one round trip: a heavy scan's candidates are almost all true k-mers of the
                        // index set, the short circuit b -> c -> d would only serialise them
                        const uint32_t vb = pb[kb >> 5], vc = pc[(ka ^ kb) >> 5], vd = pd[(ka | kb) >> 5];
                        if (!((vb >> ((uint32_t) kb & 31u)) & (vc >> ((uint32_t) (ka ^ kb) & 31u)) & (vd >> ((uint32_t) (ka | kb) & 31u)) & 1u)) continue;
                    }
                    ++seen;
                    next_ok = q + k;
                    if (seen >= t) {
                        found = true;
                        break;
                    }
                }
            }
            // windows behind the first-hit ones, 32 at a time, for the scans that have a hit but not yet t of them
            const bool tails_on = !(COMMET_TQ_ABLATE & 32768) && !((COMMET_TQ_ABLATE & 65536) && heavy) && !((COMMET_TQ_ABLATE & 131072) && !heavy);
            for (int qb = max(pe + 1, next_ok);; qb += 32) {   // (uniform trip count: every thread takes part in the barriers)
                const bool want = tails_on && !found && !dead && seen >= 1 && qb <= last && qb + (t - seen - 1) * k <= last;
                if (threadIdx.x == 0) tail_n = 0;
                if (!__syncthreads_or(want)) break;
                if (want) {
                    tail_req[atomicAdd(&tail_n, 1u)] = threadIdx.x | ((uint32_t) qb << 8);
                    tail_bits[threadIdx.x] = 0;
                }
                __syncthreads();
                const uint32_t n_pairs = tail_n * 32u;
                for (uint32_t pr = threadIdx.x; pr < n_pairs; pr += TQ_PIECE) {
                    const uint32_t rq = tail_req[pr >> 5], owner = rq & 255u, w = pr & 31u;
                    const int q = (int) (rq >> 8) + (int) w;
                    const uint32_t oext = rd_ext[owner];
                    if (q >= (int) (oext >> 16)) continue;
                    ItemWords<W> it;
                    it.load(piece_planes + 3u * (oext & 0xFFFFu), (uint32_t) q >> 5);
                    W wh, wl;
                    if (!it.window((uint32_t) q & 31u, k, kmask, wh, wl)) continue;   // a base that is not ACGT: no k-mer here
                    const W ka = strand ? (W) (~wh & kmask) : (W) (T::brev(wh) >> sh);
                    const W addr = psi_a<W>(ka, k);
                    const uint32_t v = (COMMET_TQ_ABLATE & 524288) ? fg.il_a[(uint64_t) ((addr >> 5) & 0x1FFFFu) * GS + (uint32_t) (i >> 1)]   // timing bound: tails served from L2
                                                                   : fg.il_a[(uint64_t) (addr >> 5) * GS + (uint32_t) (i >> 1)];
                    if ((v >> ((uint32_t) addr & 31u)) & 1u) atomicOr(&tail_bits[owner], 1u << w);
                }
                __syncthreads();
                if (want) {
                    uint32_t m = tail_bits[threadIdx.x];
                    while (m && !found) {
                        const int q = qb + (__ffs((int) m) - 1);
                        m &= m - 1u;
                        if (q < next_ok) continue;
                        if (q + (t - seen - 1) * k > last) {
                            dead = true;
                            break;
                        }
                        ItemWords<W> it;
                        it.load(p, (uint32_t) q >> 5);
                        W wh, wl, ka, kb;
                        (void) it.window((uint32_t) q & 31u, k, kmask, wh, wl);
                        if (strand) ka = ~wh & kmask, kb = ~wl & kmask;
                        else ka = T::brev(wh) >> sh, kb = T::brev(wl) >> sh;
                        const uint32_t vb = pb[kb >> 5], vc = pc[(ka ^ kb) >> 5], vd = pd[(ka | kb) >> 5];   // one round trip
                        if ((vb >> ((uint32_t) kb & 31u)) & (vc >> ((uint32_t) (ka ^ kb) & 31u)) & (vd >> ((uint32_t) (ka | kb) & 31u)) & 1u) {
                            ++seen;
                            next_ok = q + k;
                            if (seen >= t) found = true;
                        }
                    }
                }
            }
            if (found && found_chunk < 0) found_chunk = i >> 1;
        }
        found_job1 = found;
    }
    if constexpr (GS == 2) if (multi) {
        const uint64_t fb0 = __ballot(found_job0), fb1 = __ballot(found_job1), sc = __ballot(active);
        if (lane == 0 && in_range && tags) tags[word] = fb0, tags[job_tag_words + word] = fb1;
        __syncthreads();
        if (counters) {
            if (lane == 0) {
                if (sc) atomicAdd(&wg_cnt[0], (unsigned int) __popcll(sc)), atomicAdd(&wg_cnt[2], (unsigned int) __popcll(sc));
                if (fb0) atomicAdd(&wg_cnt[1], (unsigned int) __popcll(fb0));
                if (fb1) atomicAdd(&wg_cnt[3], (unsigned int) __popcll(fb1));
            }
            __syncthreads();
            if (threadIdx.x < 4u && wg_cnt[threadIdx.x])
                atomicAdd(&counters[(uint64_t) (threadIdx.x >> 1) * cstride + (threadIdx.x & 1)], (unsigned long long) wg_cnt[threadIdx.x]);
        }
        return;
    }
    const bool found = found_chunk >= 0;
    const uint64_t fb = __ballot(found);
    if (lane == 0 && in_range && tags) tags[word] = tagw | fb;
    __syncthreads();
    if (counters) {
        // per workgroup in LDS first: every wave adding to the same two global words costs milliseconds of serialised atomics
        for (int i = 0; i < fg.g; ++i) {
            const uint64_t sc = __ballot(active && (found_chunk < 0 || found_chunk >= i));
            const uint64_t fd = __ballot(found_chunk == i);
            if (lane == 0) {
                if (sc) atomicAdd(&wg_cnt[2 * i], (unsigned int) __popcll(sc));
                if (fd) atomicAdd(&wg_cnt[2 * i + 1], (unsigned int) __popcll(fd));
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * (unsigned) fg.g && wg_cnt[threadIdx.x])
            atomicAdd(&counters[(uint64_t) (threadIdx.x >> 1) * cstride + (threadIdx.x & 1)], (unsigned long long) wg_cnt[threadIdx.x]);
    }
}

}  // namespace commet
