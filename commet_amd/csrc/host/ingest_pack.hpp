// ingest_pack.hpp — host half of the read-set ingest: records -> 2-bit planes, packed by the ingest threads.
//
// The device layout of a read set (kernels.hpp) is, per read, ceil(len / 32) word triples {hi, lo, valid}: bit j of a
// word = base 32 * w + j, hi = 1 for G/T, lo = 1 for C/T, valid = 1 for ACGTacgt (alphabet.h:44-58, hash_key.h:72-88);
// read r starts at triple (goff[r] >> 5) + r.  Packing on the host, in the threads that parse the files, sends 12 bytes
// per 32 bases over PCIe instead of 32 (+ 8 per read of offsets for a device-side packer), and needs no packing kernel.
// A 32-base block is three AVX2 movemasks (scalar code when the CPU has no AVX2).
//
// HIP-free on purpose: the sink that owns the staging buffers and uploads them is a template parameter, so the same
// code runs in libcommet_hip.so (pinned buffers + hipMemcpyAsync, capi.hip) and in the CPU-only checker
// host/ingest_check.cpp that the sanitizer tests build (tests/test_sanitizers.py).
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "fasta_source.hpp"

namespace commet_host {

inline void pack32_scalar(const uint8_t *s, uint32_t n, uint32_t &hi, uint32_t &lo, uint32_t &va)
{
    hi = lo = va = 0;
    for (uint32_t j = 0; j < n; ++j) {
        const uint32_t ch = s[j], u = ch & 0xDFu;
        const uint32_t v = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
        hi |= (v & ((ch >> 2) & 1u)) << j;
        lo |= (v & (((ch >> 1) ^ (ch >> 2)) & 1u)) << j;
        va |= v << j;
    }
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void pack32_avx2(const uint8_t *s, uint32_t &hi, uint32_t &lo, uint32_t &va)
{
    const __m256i x = _mm256_loadu_si256((const __m256i *) s);
    const __m256i u = _mm256_and_si256(x, _mm256_set1_epi8((char) 0xDF));
    const __m256i v = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, _mm256_set1_epi8('A')), _mm256_cmpeq_epi8(u, _mm256_set1_epi8('C'))),
                                      _mm256_or_si256(_mm256_cmpeq_epi8(u, _mm256_set1_epi8('G')), _mm256_cmpeq_epi8(u, _mm256_set1_epi8('T'))));
    const uint32_t vm = (uint32_t) _mm256_movemask_epi8(v);
    const uint32_t b2 = (uint32_t) _mm256_movemask_epi8(_mm256_slli_epi16(x, 5));   // bit 2 of every byte
    const uint32_t b1 = (uint32_t) _mm256_movemask_epi8(_mm256_slli_epi16(x, 6));   // bit 1 of every byte
    hi = b2 & vm;
    lo = (b1 ^ b2) & vm;
    va = vm;
}
inline bool cpu_has_avx2()
{
    static const bool has = __builtin_cpu_supports("avx2");
    return has;
}
#endif

// 32 whole bases at s -> one triple
inline void pack32(const uint8_t *s, uint32_t &hi, uint32_t &lo, uint32_t &va)
{
#if defined(__x86_64__)
    if (cpu_has_avx2()) {
        pack32_avx2(s, hi, lo, va);
        return;
    }
#endif
    pack32_scalar(s, 32, hi, lo, va);
}

// One staging buffer lent by the sink: planes of cap_triples triples, base offsets of cap_reads reads.
struct PackStage {
    uint32_t *planes = nullptr;
    uint64_t *goff = nullptr;
    uint64_t cap_triples = 0, cap_reads = 0;
};

// What a worker learned about its reads (merged into the read set at the end)
struct PackSummary {
    uint32_t min_len = 0xFFFFFFFFu, max_len = 0;
    std::vector<uint64_t> empty_reads;
    void merge(const PackSummary &o)
    {
        min_len = std::min(min_len, o.min_len);
        max_len = std::max(max_len, o.max_len);
        empty_reads.insert(empty_reads.end(), o.empty_reads.begin(), o.empty_reads.end());
    }
};

// Sink concept:
//   bool acquire(int worker, PackStage &st);     a staging buffer whose previous upload has completed
//   bool flush(int worker, const PackStage &st, uint64_t triple0, uint64_t n_triples, uint64_t read0, uint64_t n_reads);
//                                                 planes[0 .. 3 * n_triples) -> device triples triple0 ..,
//                                                 goff[0 .. n_reads) -> device offsets read0 ..; the buffer is the sink's again
template <class Sink>
class PackWriter {
public:
    PackWriter(Sink &sink, int worker) : sink_(sink), worker_(worker) {}
    PackSummary summary;

    // the next read: set-wide number r, first base at set-wide base offset goff
    bool read_begin(uint64_t r, uint64_t goff)
    {
        const uint64_t T = (goff >> 5) + r;
        if (have_ && (T - tbase_ >= st_.cap_triples || rfill_ >= st_.cap_reads || r != rbase_ + rfill_)) {
            if (!flush_all()) return false;
        }
        if (!have_) {
            if (!sink_.acquire(worker_, st_)) return false;
            have_ = true;
            tbase_ = T, tfill_ = 0, rbase_ = r, rfill_ = 0;
        }
        while (tbase_ + tfill_ < T) put(0, 0, 0);   // the unused triple between two reads (at most one) is zeroed
        st_.goff[rfill_++] = goff;
        hi_ = lo_ = va_ = nb_ = 0;
        len_ = 0;
        cur_ = r;
        return ok_;
    }
    // more bases of the current read (one call per sequence line of a multi-line record)
    bool append(const uint8_t *s, size_t n)
    {
        len_ += n;
        // up to the next word boundary: bit by bit
        while (n && nb_) {
            add_base(*s++);
            --n;
        }
        while (n >= 32) {
            uint32_t h, l, v;
            pack32(s, h, l, v);
            if (!put(h, l, v)) return false;
            s += 32, n -= 32;
        }
        if (n && nb_ == 0) {
            // the read's last, partial word (or a line's, in a multi-line record) in one go from a zero-padded copy: zero bytes are no
            // bases, so the bits past n stay clear.  (Base by base this was 16 scalar steps per read of a ragged set, 4 of a 100-bp one.)
            uint8_t buf[32] = {0};
            memcpy(buf, s, n);
            pack32(buf, hi_, lo_, va_);
            nb_ = (uint32_t) n;
            n = 0;
        }
        while (n) {
            add_base(*s++);
            --n;
        }
        return ok_;
    }
    bool read_end()
    {
        if (nb_) {
            if (!put(hi_, lo_, va_)) return false;
            nb_ = 0;
        }
        if (len_ > 0x7FFFFFFFull) return fail("read longer than 2^31-1 bases");
        if (len_ == 0) summary.empty_reads.push_back(cur_);
        summary.min_len = std::min<uint32_t>(summary.min_len, (uint32_t) len_);
        summary.max_len = std::max<uint32_t>(summary.max_len, (uint32_t) len_);
        return ok_;
    }
    bool close() { return !have_ || flush_all(); }
    const std::string &error() const { return err_; }

private:
    bool fail(const char *m)
    {
        err_ = m;
        ok_ = false;
        return false;
    }
    void add_base(uint8_t ch)
    {
        const uint32_t u = ch & 0xDFu;
        const uint32_t v = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
        hi_ |= (v & ((ch >> 2) & 1u)) << nb_;
        lo_ |= (v & (((ch >> 1) ^ (ch >> 2)) & 1u)) << nb_;
        va_ |= v << nb_;
        if (++nb_ == 32) {
            put(hi_, lo_, va_);
            hi_ = lo_ = va_ = nb_ = 0;
        }
    }
    // appends one triple; a full buffer is uploaded first (also in the middle of a read longer than the buffer)
    bool put(uint32_t h, uint32_t l, uint32_t v)
    {
        if (tfill_ == st_.cap_triples) {
            if (!sink_.flush(worker_, st_, tbase_, tfill_, rbase_, rfill_)) return fail("upload failed");
            tbase_ += tfill_, rbase_ += rfill_;
            tfill_ = 0, rfill_ = 0;
            if (!sink_.acquire(worker_, st_)) return fail("no staging buffer");
        }
        uint32_t *d = st_.planes + 3 * tfill_++;
        d[0] = h, d[1] = l, d[2] = v;
        return true;
    }
    bool flush_all()
    {
        have_ = false;
        if (tfill_ == 0 && rfill_ == 0) return ok_;
        if (!sink_.flush(worker_, st_, tbase_, tfill_, rbase_, rfill_)) return fail("upload failed");
        return ok_;
    }

    Sink &sink_;
    int worker_;
    PackStage st_;
    bool have_ = false, ok_ = true;
    uint64_t tbase_ = 0, tfill_ = 0, rbase_ = 0, rfill_ = 0, cur_ = 0, len_ = 0;
    uint32_t hi_ = 0, lo_ = 0, va_ = 0, nb_ = 0;
    std::string err_;
};

// A piece = a run of whole records of one file.  FASTA files are cut at lines starting with '>' (a record boundary by
// the reference's own rule, fasta_file.h:61-68); FASTQ files stay one piece ('@' may also start a quality line).
struct IngestPiece {
    int file = 0;
    ReadFormat fmt = ReadFormat::Fasta;
    const char *d = nullptr;
    size_t n = 0;
    uint64_t n_reads = 0, n_bases = 0;     // pass A
    uint64_t read0 = 0, base0 = 0;         // prefix over the pieces
};

inline void count_piece(IngestPiece &p)
{
    if (p.fmt == ReadFormat::Fastq) {
        p.n_reads = count_fastq_records(p.d, p.n);
        for_each_fastq_record(p.d, p.n, p.n_reads, [&](const char *, size_t len) { p.n_bases += len; });
        return;
    }
    const char *d = p.d;
    const size_t n = p.n;
    size_t i = 0;
    while (i < n && d[i] != '>') {   // bytes before the first header line belong to no record
        const char *nl = (const char *) memchr(d + i, '\n', n - i);
        i = nl ? (size_t) (nl - d) + 1 : n;
    }
    while (i < n) {
        const char *nl = (const char *) memchr(d + i, '\n', n - i);   // header
        size_t j = nl ? (size_t) (nl - d) + 1 : n;
        ++p.n_reads;
        while (j < n && d[j] != '>') {
            nl = (const char *) memchr(d + j, '\n', n - j);
            const size_t e = nl ? (size_t) (nl - d) : n;
            p.n_bases += e - j;
            j = nl ? e + 1 : n;
        }
        i = j;
    }
}

// cuts file f (d, n) into at most `want` pieces of whole FASTA records
inline void split_file(int f, ReadFormat fmt, const char *d, size_t n, int want, std::vector<IngestPiece> &pieces)
{
    if (fmt != ReadFormat::Fasta) want = 1;
    size_t b = 0;
    for (int q = 0; q < want && b < n; ++q) {
        size_t e = (q == want - 1) ? n : std::min(n, (size_t) ((double) n * (q + 1) / want));
        if (e < n) {   // advance to the next line that starts with '>'
            const char *x = d + e;
            for (;;) {
                const char *nl = (const char *) memchr(x, '\n', (size_t) (d + n - x));
                if (!nl || nl + 1 >= d + n) { e = n; break; }
                if (nl[1] == '>') { e = (size_t) (nl + 1 - d); break; }
                x = nl + 1;
            }
        }
        if (e > b) {
            IngestPiece p;
            p.file = f, p.fmt = fmt, p.d = d + b, p.n = e - b;
            pieces.push_back(p);
        }
        b = e;
    }
}

// records of one piece -> writer (records as the reference reads them: fasta_file.h:155-175, fastq_file.h:139-190)
template <class Sink>
bool pack_piece(const IngestPiece &p, PackWriter<Sink> &w)
{
    uint64_t r = p.read0, g = p.base0;
    if (p.fmt == ReadFormat::Fastq) {
        bool ok = true;
        for_each_fastq_record(p.d, p.n, p.n_reads, [&](const char *s, size_t len) {
            if (!ok) return;
            ok = w.read_begin(r, g) && w.append((const uint8_t *) s, len) && w.read_end();
            ++r, g += len;
        });
        return ok;
    }
    const char *d = p.d;
    const size_t n = p.n;
    size_t i = 0;
    while (i < n && d[i] != '>') {
        const char *nl = (const char *) memchr(d + i, '\n', n - i);
        i = nl ? (size_t) (nl - d) + 1 : n;
    }
    while (i < n) {
        const char *nl = (const char *) memchr(d + i, '\n', n - i);   // header line
        size_t j = nl ? (size_t) (nl - d) + 1 : n;
        if (!w.read_begin(r, g)) return false;
        while (j < n && d[j] != '>') {
            nl = (const char *) memchr(d + j, '\n', n - j);
            const size_t e = nl ? (size_t) (nl - d) : n;
            if (e > j && !w.append((const uint8_t *) d + j, e - j)) return false;
            g += e - j;
            j = nl ? e + 1 : n;
        }
        if (!w.read_end()) return false;
        ++r;
        i = j;
    }
    return true;
}

inline int ingest_threads()
{
    const char *e = getenv("COMMET_INGEST_THREADS");
    int t = e ? atoi(e) : 32;   // packing is memory-bound: more threads than cores still help (measured: 16 -> 28 ms, 32 -> 20 ms per GB)
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && t > (int) hw) t = (int) hw;
    return t < 1 ? 1 : t;
}

// runs fn(worker, item) for every item in [0, n) on up to T threads (items are taken from a shared counter)
inline void parallel_items(int T, size_t n, const std::function<void(int, size_t)> &fn)
{
    std::atomic<size_t> next{0};
    const int nt = (int) std::min<size_t>((size_t) std::max(T, 1), std::max<size_t>(n, 1));
    std::vector<std::thread> th;
    auto loop = [&](int t) {
        for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(t, i);
    };
    for (int t = 1; t < nt; ++t) th.emplace_back(loop, t);
    loop(0);
    for (std::thread &x : th) x.join();
}

// Pass A + B over whole files: counts, then packs every piece through `sink` with T workers.
// Fills file_reads (records per file), totals, and the merged summary.  make_sink is called once the totals are known:
//   Sink *make_sink(total_reads, total_bases, workers)   (nullptr = failure)
template <class Sink, class MakeSink>
bool ingest_files(const std::vector<const char *> &data, const std::vector<size_t> &sizes, const std::vector<ReadFormat> &fmts,
                  int T, MakeSink &&make_sink, std::vector<uint64_t> &file_reads, uint64_t &total_reads, uint64_t &total_bases,
                  PackSummary &summary, std::string &err)
{
    std::vector<IngestPiece> pieces;
    for (size_t f = 0; f < data.size(); ++f)
        split_file((int) f, fmts[f], data[f], sizes[f], (fmts[f] == ReadFormat::Fasta && sizes[f] > (8u << 20)) ? T * 4 : 1, pieces);
    parallel_items(T, pieces.size(), [&](int, size_t i) { count_piece(pieces[i]); });
    total_reads = total_bases = 0;
    file_reads.assign(data.size(), 0);
    for (IngestPiece &p : pieces) {
        p.read0 = total_reads, p.base0 = total_bases;
        total_reads += p.n_reads, total_bases += p.n_bases;
        file_reads[p.file] += p.n_reads;
    }
    const int nt = (int) std::min<size_t>((size_t) T, std::max<size_t>(pieces.size(), 1));
    Sink *sink = make_sink(total_reads, total_bases, nt);
    if (!sink) {
        err = "cannot set up the upload of the read set";
        return false;
    }
    std::mutex mu;
    std::atomic<bool> failed{false};
    std::vector<PackWriter<Sink>> writers;
    writers.reserve(nt);
    for (int t = 0; t < nt; ++t) writers.emplace_back(*sink, t);
    parallel_items(nt, pieces.size(), [&](int t, size_t i) {
        if (failed) return;
        if (!pack_piece(pieces[i], writers[t])) {
            std::lock_guard<std::mutex> lk(mu);
            if (err.empty()) err = writers[t].error().empty() ? "read set ingest failed" : writers[t].error();
            failed = true;
        }
    });
    for (int t = 0; t < nt; ++t) {
        if (!failed && !writers[t].close()) {
            err = writers[t].error().empty() ? "upload failed" : writers[t].error();
            failed = true;
        }
        summary.merge(writers[t].summary);
    }
    std::sort(summary.empty_reads.begin(), summary.empty_reads.end());
    return !failed;
}

// The same for reads already in memory as (bases, offsets[n + 1]): reads [0, n) become set-wide reads read0 + i at base
// offsets base0 + offsets[i].
template <class Sink>
bool ingest_arrays(const uint8_t *bases, const uint64_t *offsets, uint64_t n, uint64_t read0, uint64_t base0, int T, Sink &sink,
                   PackSummary &summary, std::string &err)
{
    const uint64_t per = 1u << 16;                       // reads per work item
    const size_t items = (size_t) ((n + per - 1) / per);
    const int nt = (int) std::min<size_t>((size_t) std::max(T, 1), std::max<size_t>(items, 1));
    std::mutex mu;
    std::atomic<bool> failed{false};
    std::vector<PackWriter<Sink>> writers;
    writers.reserve(nt);
    for (int t = 0; t < nt; ++t) writers.emplace_back(sink, t);
    parallel_items(nt, items, [&](int t, size_t it) {
        if (failed) return;
        PackWriter<Sink> &w = writers[t];
        const uint64_t a = it * per, b = std::min<uint64_t>(n, a + per);
        for (uint64_t i = a; i < b; ++i) {
            if (offsets[i + 1] < offsets[i]) {
                std::lock_guard<std::mutex> lk(mu);
                err = "offsets must be non-decreasing";
                failed = true;
                return;
            }
            if (!(w.read_begin(read0 + i, base0 + offsets[i]) && w.append(bases + offsets[i], (size_t) (offsets[i + 1] - offsets[i])) && w.read_end())) {
                std::lock_guard<std::mutex> lk(mu);
                if (err.empty()) err = w.error().empty() ? "read set ingest failed" : w.error();
                failed = true;
                return;
            }
        }
    });
    for (int t = 0; t < nt; ++t) {
        if (!failed && !writers[t].close()) {
            err = writers[t].error().empty() ? "upload failed" : writers[t].error();
            failed = true;
        }
        summary.merge(writers[t].summary);
    }
    std::sort(summary.empty_reads.begin(), summary.empty_reads.end());
    return !failed;
}

}  // namespace commet_host
