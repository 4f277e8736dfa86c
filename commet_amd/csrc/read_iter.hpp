// read_iter.hpp — host-side restatement of the reference's read iteration rules,
// turned from a pull iterator into a *plan* over resident reads.
//
// The reference walks a set through FileManager::get_next_read_to_compare()
// (include/file_manager.h:88-112) on top of FastaFile::get_next_read()
// (include/fasta_file.h:132-183).  Which reads come out, in which order, where
// a pass stops, and how many fetches are counted decide
//   - the reads of every index chunk, incl. the look-ahead read that is
//     dropped when a chunk fills (index_reads.h:49,60)            [SURVEY Q1-Q3]
//   - which reads a search pass scans (a set stops at a file with no selected
//     read, file_manager.h:90-98)                                  [SURVEY Q6]
//   - the loop condition of main() (index_and_search.cpp:255)     [SURVEY Q2]
// The classes below keep the same counters (current_read_pos, first_read,
// _cnt_valid_reads, current_file, nb_seen_reads) but run over read *numbers*.
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

namespace commet {

struct FileSpan {
    uint64_t first;   // set-wide number of the file's first read
    uint64_t count;   // reads in the file (selected or not)
};

inline bool bit_at(const uint8_t *bits, uint64_t i) { return (bits[i >> 3] >> (i & 7)) & 1; }
inline void bit_on(uint8_t *bits, uint64_t i) { bits[i >> 3] |= (uint8_t) (1u << (i & 7)); }

// number of set bits in [lo, hi)
inline uint64_t count_bits(const uint8_t *bits, uint64_t lo, uint64_t hi)
{
    uint64_t n = 0;
    while (lo < hi && (lo & 7)) n += bit_at(bits, lo++);
    while (hi > lo && (hi & 7)) n += bit_at(bits, --hi);
    uint64_t b = lo >> 3;
    const uint64_t be = hi >> 3;
    for (; b + 8 <= be; b += 8) {
        uint64_t w;
        __builtin_memcpy(&w, bits + b, 8);
        n += (uint64_t) __builtin_popcountll(w);
    }
    for (; b < be; ++b) n += (uint64_t) __builtin_popcount(bits[b]);
    return n;
}

// true when bits [0, n) are all set
inline bool all_ones(const uint8_t *bits, uint64_t n)
{
    const uint64_t full = n >> 3;
    uint64_t b = 0;
    for (; b + 8 <= full; b += 8) {
        uint64_t w;
        __builtin_memcpy(&w, bits + b, 8);
        if (w != ~0ull) return false;
    }
    for (; b < full; ++b)
        if (bits[b] != 0xFF) return false;
    for (uint64_t i = full << 3; i < n; ++i)
        if (!bit_at(bits, i)) return false;
    return true;
}

// first set bit at or after `from`, below `end`; returns end if none
inline uint64_t next_set_bit(const uint8_t *bits, uint64_t from, uint64_t end)
{
    while (from < end) {
        const uint32_t byte = bits[from >> 3] >> (from & 7);
        if (byte) {
            const uint64_t r = from + (uint64_t) __builtin_ctz(byte);
            return r < end ? r : end;
        }
        from = (from | 7) + 1;
        // skip zero bytes quickly
        while (from + 64 <= end && (from & 63) == 0) {
            uint64_t w;
            __builtin_memcpy(&w, bits + (from >> 3), 8);
            if (w) break;
            from += 64;
        }
    }
    return end;
}

// One file: FastaFile's iterator state (fasta_file.h:132-183, 258-264).
struct FileCursor {
    FileSpan span{0, 0};
    uint64_t nb_valid = 0;    // _nb_valid_reads = popcount of the input filter
    uint64_t cnt_valid = 0;   // _cnt_valid_reads
    uint64_t pos = 0;         // current_read_pos
    bool     first_read = true;

    void rewind()
    {
        cnt_valid = 0;
        pos = 0;
        first_read = true;
    }
};

class SetIterator {
public:
    static constexpr uint64_t NONE = ~0ull;   // the empty-string sentinel

    // select: set-wide input-filter bits (nullptr = all ones); empty_reads:
    // sorted set-wide numbers of reads whose sequence is empty.
    SetIterator(const std::vector<FileSpan> &files, const uint8_t *select, const std::vector<uint64_t> &empty_reads)
        : select_(select), empty_(empty_reads)
    {
        for (const FileSpan &s : files) {
            FileCursor c;
            c.span = s;
            c.nb_valid = select ? count_bits(select, s.first, s.first + s.count) : s.count;
            files_.push_back(c);
        }
        current_file_ = files_.empty() ? -1 : 0;
    }

    uint64_t total_valid() const   // FileManager::get_total_nb_reads, file_manager.h:268-274
    {
        uint64_t s = 0;
        for (const FileCursor &c : files_) s += c.nb_valid;
        return s;
    }
    uint64_t seen() const { return nb_seen_; }

    void rewind()   // file_manager.h:223-229
    {
        current_file_ = 0;
        nb_seen_ = 0;
        for (FileCursor &c : files_) c.rewind();
    }

    // get_next_read_to_compare with an all-false tag vector (tagged reads are
    // skipped later on the device; skipping never changes which of the other
    // reads are produced).  Returns a set-wide read number or NONE.
    uint64_t next()
    {
        const int nfiles = (int) files_.size();
        if (current_file_ < 0 || current_file_ >= nfiles) {   // reference: out-of-range access (UB)
            ++nb_seen_;
            return NONE;
        }
        uint64_t r = file_next(files_[current_file_]);
        if (r == NONE) {
            ++current_file_;
            if (current_file_ >= nfiles) {
                ++nb_seen_;
                return file_next(files_[current_file_ - 1]);   // file_manager.h:94-95
            }
            r = file_next(files_[current_file_]);
        }
        ++nb_seen_;
        return r;
    }

private:
    bool is_empty_read(uint64_t r) const { return std::binary_search(empty_.begin(), empty_.end(), r); }

    // FastaFile::get_next_read
    uint64_t file_next(FileCursor &c)
    {
        if (c.first_read) c.first_read = false;
        else ++c.pos;
        if (c.cnt_valid < c.nb_valid) {
            if (select_)   // skip unselected reads (:143-152)
                c.pos = next_set_bit(select_, c.span.first + c.pos, c.span.first + c.span.count) - c.span.first;
            if (c.pos < c.span.count) {
                const uint64_t r = c.span.first + c.pos;
                if (!empty_.empty() && is_empty_read(r)) return NONE;   // empty sequence == EOF sentinel (:178-182)
                ++c.cnt_valid;
                return r;
            }
        }
        return NONE;
    }

    std::vector<FileCursor> files_;
    const uint8_t *select_;
    const std::vector<uint64_t> &empty_;
    int current_file_ = -1;
    uint64_t nb_seen_ = 0;
};

struct Chunk {
    uint64_t first = 0, last = 0;   // inclusive range of set-wide read numbers (valid when n_reads > 0)
    uint64_t n_reads = 0;
    uint64_t kmers = 0;
};

struct IndexPlan {
    std::vector<Chunk>   chunks;
    std::vector<uint8_t> indexed_bits;   // reads actually fed to a filter (selected, not dropped)
    uint64_t             indexed_reads = 0;
    uint64_t             kmers = 0;
    bool                 dense = false;  // every read of every chunk's [first, last] is indexed (no selection inside a chunk)
};

// The chunk loop of main() (index_and_search.cpp:255-263) + index_reads
// (index_reads.h:41-63) run over read numbers and per-read k-mer counts.
inline IndexPlan plan_index(const std::vector<FileSpan> &files, const uint8_t *select,
                            const std::vector<uint64_t> &empty_reads, const uint32_t *kcnt, uint64_t n_reads,
                            uint64_t max_kmer)
{
    IndexPlan plan;
    plan.indexed_bits.assign(n_reads / 8 + 1, 0);
    SetIterator it(files, select, empty_reads);
    const uint64_t to_index = it.total_valid();
    while (it.seen() < to_index) {
        Chunk ch;
        uint64_t r = it.next();
        while (r != SetIterator::NONE && ch.kmers < max_kmer) {
            if (ch.n_reads == 0) ch.first = r;
            ch.last = r;
            ++ch.n_reads;
            ch.kmers += kcnt[r];
            bit_on(plan.indexed_bits.data(), r);
            r = it.next();   // look-ahead fetch; dropped if the chunk is full
        }
        plan.indexed_reads += ch.n_reads;
        plan.kmers += ch.kmers;
        plan.chunks.push_back(ch);
    }
    return plan;
}

// Reads a search pass visits when nothing is tagged yet (search_reads.h:41-86).
inline std::vector<uint8_t> plan_search(const std::vector<FileSpan> &files, const uint8_t *select,
                                        const std::vector<uint64_t> &empty_reads, uint64_t n_reads, uint64_t *n_visited)
{
    std::vector<uint8_t> bits(n_reads / 8 + 1, 0);
    SetIterator it(files, select, empty_reads);
    it.rewind();
    uint64_t n = 0;
    for (uint64_t r = it.next(); r != SetIterator::NONE; r = it.next()) {
        bit_on(bits.data(), r);
        ++n;
    }
    if (n_visited) *n_visited = n;
    return bits;
}

// ---------------------------------------------------------------------------
// Fast plans for the common case: no input filter, no empty sequence, no empty
// file.  Then the iterator yields reads 0,1,2,... one per call, and the chunk
// boundaries follow from prefix sums of the per-read k-mer counts:
//   chunk [s, e-1], e = first index with P[e] - P[s] >= max_kmer (or n),
//   read e is the dropped look-ahead, next chunk starts at e+1; the loop of
//   main() ends when e+1 >= n (index_and_search.cpp:255).
// Must agree with plan_index / plan_search exactly (tests/test_host_plan.py).
// ---------------------------------------------------------------------------
inline bool plan_fast_ok(const std::vector<FileSpan> &files, const uint8_t *select,
                         const std::vector<uint64_t> &empty_reads, uint64_t max_kmer)
{
    if (select || !empty_reads.empty() || max_kmer == 0 || files.empty()) return false;
    for (const FileSpan &f : files)
        if (f.count == 0) return false;
    return true;
}

inline void build_kmer_prefix(const uint32_t *kcnt, uint64_t n_reads, std::vector<uint64_t> &prefix)
{
    prefix.resize(n_reads + 1);
    uint64_t s = 0;
    for (uint64_t i = 0; i < n_reads; ++i) {
        prefix[i] = s;
        s += kcnt[i];
    }
    prefix[n_reads] = s;
}

inline IndexPlan plan_index_fast(const std::vector<uint64_t> &prefix, uint64_t n_reads, uint64_t max_kmer)
{
    IndexPlan plan;
    plan.dense = true;   // the dropped look-ahead reads lie between the chunks
    plan.indexed_bits.assign(n_reads / 8 + 1, 0);
    for (uint64_t i = 0; i < n_reads / 8; ++i) plan.indexed_bits[i] = 0xFF;
    for (uint64_t i = (n_reads / 8) * 8; i < n_reads; ++i) bit_on(plan.indexed_bits.data(), i);
    uint64_t s = 0;
    while (s < n_reads) {
        const uint64_t target = prefix[s] + max_kmer;
        // first e in (s, n] with prefix[e] >= target
        const uint64_t *b = prefix.data() + s + 1, *e_ptr = prefix.data() + n_reads + 1;
        const uint64_t *it = std::lower_bound(b, e_ptr, target);
        const uint64_t e = it == e_ptr ? n_reads : (uint64_t) (it - prefix.data());
        Chunk ch;
        ch.first = s;
        ch.last = e - 1;
        ch.n_reads = e - s;
        ch.kmers = prefix[e] - prefix[s];
        plan.chunks.push_back(ch);
        plan.indexed_reads += ch.n_reads;
        plan.kmers += ch.kmers;
        if (e < n_reads) plan.indexed_bits[e >> 3] &= (uint8_t) ~(1u << (e & 7));   // dropped look-ahead read
        s = e + 1;
    }
    return plan;
}

inline std::vector<uint8_t> plan_search_fast(uint64_t n_reads, uint64_t *n_visited)
{
    std::vector<uint8_t> bits(n_reads / 8 + 1, 0);
    for (uint64_t i = 0; i < n_reads / 8; ++i) bits[i] = 0xFF;
    for (uint64_t i = (n_reads / 8) * 8; i < n_reads; ++i) bit_on(bits.data(), i);
    if (n_visited) *n_visited = n_reads;
    return bits;
}

// ---------------------------------------------------------------------------
// Plans with an input filter but no empty sequence: the iterator reduces to
// "next set bit of the current file", plus one end-marker when a file with no
// selected read is entered from the previous file (a leading such file is
// skipped silently), plus the final end-marker.  Same results as SetIterator,
// without its per-read bookkeeping (tests/test_host_plan.py).
// ---------------------------------------------------------------------------
class SelectedIterator {
public:
    static constexpr uint64_t NONE = ~0ull;
    SelectedIterator(const std::vector<FileSpan> &files, const uint8_t *select) : files_(files), select_(select)
    {
        nb_valid_.reserve(files.size());
        for (const FileSpan &f : files) {
            nb_valid_.push_back(count_bits(select, f.first, f.first + f.count));
            total_ += nb_valid_.back();
        }
        if (!files_.empty()) {
            pos_ = files_[0].first;
            end_ = files_[0].first + files_[0].count;
        }
    }
    uint64_t total_valid() const { return total_; }
    const std::vector<uint64_t> &nb_valid() const { return nb_valid_; }
    inline uint64_t next()
    {
        for (;;) {
            if (cf_ >= files_.size()) return NONE;
            if (pos_ < end_) {
                const uint64_t r = next_set_bit(select_, pos_, end_);
                if (r < end_) {
                    pos_ = r + 1;
                    return r;
                }
                pos_ = end_;
            }
            ++cf_;                                    // current file exhausted
            if (cf_ >= files_.size()) return NONE;    // end of the set
            pos_ = files_[cf_].first;
            end_ = pos_ + files_[cf_].count;
            if (nb_valid_[cf_] == 0) {                // entering a file with nothing selected ends the pass / chunk
                pos_ = end_;
                return NONE;
            }
        }
    }

private:
    const std::vector<FileSpan> &files_;
    const uint8_t *select_;
    std::vector<uint64_t> nb_valid_;
    uint64_t total_ = 0;
    size_t cf_ = 0;
    uint64_t pos_ = 0, end_ = 0;
};

// Event-driven form of the two nested loops of main() / index_reads: the stream of
// fetches (selected reads in order, an end-marker when a file with nothing selected
// is entered, the final end-marker) is generated 64 select bits at a time.
inline IndexPlan plan_index_select(const std::vector<FileSpan> &files, const uint8_t *select, const uint32_t *kcnt,
                                   uint64_t n_reads, uint64_t max_kmer)
{
    IndexPlan plan;
    plan.indexed_bits.assign(n_reads / 8 + 1, 0);
    uint8_t *bits = plan.indexed_bits.data();
    std::vector<uint64_t> nb_valid;
    uint64_t to_index = 0;
    for (const FileSpan &f : files) {
        nb_valid.push_back(count_bits(select, f.first, f.first + f.count));
        to_index += nb_valid.back();
    }
    uint64_t seen = 0;
    bool in_chunk = false, stop = false;
    Chunk ch;
    auto close_chunk = [&]() {
        plan.indexed_reads += ch.n_reads;
        plan.kmers += ch.kmers;
        plan.chunks.push_back(ch);
        ch = Chunk();
        in_chunk = false;
    };
    // one fetch of the iterator: r = read number or NONE
    auto fetch = [&](uint64_t r) {
        if (!in_chunk) {                      // outer loop test, then the first fetch of index_reads
            if (seen >= to_index) {
                stop = true;
                return;
            }
            in_chunk = true;
        }
        ++seen;
        if (r != SetIterator::NONE && ch.kmers < max_kmer) {
            if (ch.n_reads == 0) ch.first = r;
            ch.last = r;
            ++ch.n_reads;
            ch.kmers += kcnt[r];
            bit_on(bits, r);
        } else {
            close_chunk();                    // r (if a read) is the dropped look-ahead
        }
    };
    for (size_t fi = 0; fi < files.size() && !stop; ++fi) {
        const uint64_t lo = files[fi].first, hi = lo + files[fi].count;
        if (nb_valid[fi] == 0) {
            if (fi >= 1) fetch(SetIterator::NONE);   // entering a file with nothing selected
            continue;
        }
        uint64_t a = lo;
        while (a < hi && !stop) {
            // up to 64 select bits starting at a (unaligned tail handled by the mask)
            const uint64_t nb = std::min<uint64_t>(64 - (a & 7), hi - a);
            uint64_t w = 0;
            const uint64_t byte0 = a >> 3, nbytes = ((a & 7) + nb + 7) >> 3;
            __builtin_memcpy(&w, select + byte0, nbytes > 8 ? 8 : nbytes);
            w >>= (a & 7);
            if (nb < 64) w &= (1ull << nb) - 1;
            while (w && !stop) {
                const uint64_t r = a + (uint64_t) __builtin_ctzll(w);
                w &= w - 1;
                fetch(r);
            }
            a += nb;
        }
    }
    if (!stop) {
        // the final end-marker(s): index_reads keeps being called while seen < total (never more than once here)
        while (!stop && (in_chunk || seen < to_index)) fetch(SetIterator::NONE);
    }
    return plan;
}

// Plan of a selection when every file holds a selected read and there is no empty sequence: the fetch stream is the
// selected reads in order plus the final end-marker, so file borders play no role and a chunk is "selected reads from
// `first` on until their k-mers reach max_kmer; the next selected read is dropped" (index_reads.h:49-60).
// block_sums[b] = k-mers of the selected reads among reads [b * bs, (b + 1) * bs) (summed on the device, where kcnt
// lives): whole blocks are skipped, only the blocks in which a chunk starts or ends are walked read by read — no
// per-read work over the set on the host (== plan_index_select on the same input, tests/test_host_plan.py).
inline bool plan_blocks_ok(const std::vector<FileSpan> &files, const uint8_t *select, const std::vector<uint64_t> &empty_reads,
                           uint64_t max_kmer)
{
    if (!empty_reads.empty() || files.empty() || max_kmer == 0) return false;
    for (const FileSpan &f : files)
        if (f.count == 0 || (select && next_set_bit(select, f.first, f.first + f.count) >= f.first + f.count)) return false;
    return true;
}

// select == nullptr: every read is selected (the plan is then `dense`).  kcnt_of(q) = k-mers of read q; it is asked
// only for reads of the blocks in which a chunk starts or ends (the library fetches those blocks from the device).
template <typename KcntOf>
inline IndexPlan plan_index_blocks(const uint8_t *select, KcntOf &&kcnt_of, uint64_t n_reads, uint64_t max_kmer,
                                   const uint64_t *block_sums, uint64_t bs)
{
    auto sel_next = [&](uint64_t from, uint64_t end) { return select ? next_set_bit(select, from, end) : std::min(from, end); };
    auto sel_at = [&](uint64_t i) { return !select || bit_at(select, i); };
    auto sel_count = [&](uint64_t lo, uint64_t hi) { return select ? count_bits(select, lo, hi) : hi - lo; };
    IndexPlan plan;
    plan.dense = select == nullptr;
    plan.indexed_bits.assign(n_reads / 8 + 1, 0);
    if (select) __builtin_memcpy(plan.indexed_bits.data(), select, n_reads / 8);
    else __builtin_memset(plan.indexed_bits.data(), 0xFF, n_reads / 8);
    for (uint64_t i = (n_reads / 8) * 8; i < n_reads; ++i)
        if (sel_at(i)) bit_on(plan.indexed_bits.data(), i);
    uint64_t pos = sel_next(0, n_reads);   // first read of the chunk being opened
    while (pos < n_reads) {
        Chunk ch;
        ch.first = pos;
        bool full = false;
        uint64_t r = pos;
        while (r < n_reads && !full) {
            const uint64_t blk = r / bs, blk_end = std::min(n_reads, (blk + 1) * bs);
            if (r == blk * bs && ch.kmers + block_sums[blk] < max_kmer) {   // the whole block fits: its reads are not looked at
                ch.kmers += block_sums[blk];
                r = blk_end;
                continue;
            }
            for (uint64_t q = sel_next(r, blk_end); q < blk_end; q = sel_next(q + 1, blk_end)) {
                ch.kmers += kcnt_of(q);
                if (ch.kmers >= max_kmer) {   // the chunk is full with read q in it
                    full = true;
                    ch.last = q;
                    break;
                }
            }
            r = blk_end;
        }
        if (!full) {   // everything up to the end of the set: the last selected read
            uint64_t e = n_reads;
            while (!sel_at(e - 1)) --e;   // ch.first is selected
            ch.last = e - 1;
        }
        ch.n_reads = sel_count(ch.first, ch.last + 1);
        plan.chunks.push_back(ch);
        plan.indexed_reads += ch.n_reads;
        plan.kmers += ch.kmers;
        if (!full) break;                                        // closed by the final end-marker
        const uint64_t d = sel_next(ch.last + 1, n_reads);       // the look-ahead read that closes a full chunk
        if (d >= n_reads) break;
        plan.indexed_bits[d >> 3] &= (uint8_t) ~(1u << (d & 7));   // fetched, never indexed
        pos = sel_next(d + 1, n_reads);
    }
    return plan;
}

// visited = selected reads of file 0 and of the following files up to (not including) the first file with no
// selected read; whole byte ranges are copied, file edges bit by bit
inline std::vector<uint8_t> plan_search_select(const std::vector<FileSpan> &files, const uint8_t *select, uint64_t n_reads,
                                               uint64_t *n_visited)
{
    std::vector<uint8_t> bits(n_reads / 8 + 1, 0);
    uint64_t n = 0;
    for (size_t i = 0; i < files.size(); ++i) {
        const uint64_t lo = files[i].first, hi = lo + files[i].count;
        const uint64_t nv = count_bits(select, lo, hi);
        if (i >= 1 && nv == 0) break;
        n += nv;
        uint64_t a = lo;
        while (a < hi && (a & 7)) {
            if (bit_at(select, a)) bit_on(bits.data(), a);
            ++a;
        }
        uint64_t b = hi;
        while (b > a && (b & 7)) {
            --b;
            if (bit_at(select, b)) bit_on(bits.data(), b);
        }
        if (b > a) __builtin_memcpy(bits.data() + (a >> 3), select + (a >> 3), (b - a) >> 3);
    }
    if (n_visited) *n_visited = n;
    return bits;
}

}  // namespace commet
