// fasta_source.hpp — FASTA -> pinned staging batches.
// Replaces the pull parser of the reference (FastaFile, include/fasta_file.h:38-314:
// one getline per line, plus a counting pre-pass per open) by a memchr scanner
// over the mapped file that writes sequence bytes straight into the pinned
// staging buffers of the HIP library.  Record rules are the reference's:
//   - a record starts at a line whose first char is '>' (fasta_file.h:61-68)
//   - its sequence is every following line up to the next such line, '\n'
//     stripped, empty lines skipped, all other bytes kept ('\r', IUPAC, ...)
//     (fasta_file.h:155-175)
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/commet_hip.h"

namespace commet_host {

class MappedFile {
public:
    ~MappedFile() { close_file(); }
    bool open_file(const std::string &path)
    {
        close_file();
        fd_ = open(path.c_str(), O_RDONLY);
        if (fd_ < 0) return false;
        struct stat sb;
        if (fstat(fd_, &sb) != 0) return false;
        size_ = (size_t) sb.st_size;
        if (size_) {
            void *p = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
            if (p == MAP_FAILED) return false;
            data_ = (const char *) p;
            madvise((void *) data_, size_, MADV_SEQUENTIAL);
        }
        return true;
    }
    void close_file()
    {
        if (data_) munmap((void *) data_, size_);
        if (fd_ >= 0) close(fd_);
        data_ = nullptr;
        size_ = 0;
        fd_ = -1;
    }
    const char *data() const { return data_; }
    size_t size() const { return size_; }

private:
    int fd_ = -1;
    const char *data_ = nullptr;
    size_t size_ = 0;
};

// number of lines that start with '>'
inline uint64_t count_fasta_records(const char *d, size_t n)
{
    uint64_t cnt = 0;
    size_t i = 0;
    while (i < n) {
        if (d[i] == '>') ++cnt;
        const char *nl = (const char *) memchr(d + i, '\n', n - i);
        if (!nl) break;
        i = (size_t) (nl - d) + 1;
    }
    return cnt;
}

// Streams every record of the mapped FASTA into rs. Returns 0 on success.
inline int stream_fasta(commet_readset *rs, const char *d, size_t n, std::string &err)
{
    size_t i = 0;
    // bytes before the first '>' line belong to no record
    while (i < n && d[i] != '>') {
        const char *nl = (const char *) memchr(d + i, '\n', n - i);
        i = nl ? (size_t) (nl - d) + 1 : n;
    }
    uint8_t *hb = nullptr;
    uint64_t *ho = nullptr;
    uint64_t bcap = 0, rcap = 0, used = 0, nreads = 0;
    bool have = false;
    while (i < n) {
        // header line
        const char *nl = (const char *) memchr(d + i, '\n', n - i);
        size_t j = nl ? (size_t) (nl - d) + 1 : n;
        // extent of the record's sequence lines
        const size_t seq_begin = j;
        while (j < n && d[j] != '>') {
            nl = (const char *) memchr(d + j, '\n', n - j);
            j = nl ? (size_t) (nl - d) + 1 : n;
        }
        const size_t seq_end = j;
        const uint64_t upper = seq_end - seq_begin;   // >= sequence length
        if (have && (nreads >= rcap || used + upper > bcap)) {
            if (commet_readset_stage_commit(rs, nreads)) { err = commet_last_error(); return 1; }
            have = false;
        }
        if (!have) {
            if (commet_readset_stage_acquire(rs, &hb, &bcap, &ho, &rcap)) { err = commet_last_error(); return 1; }
            have = true;
            used = 0;
            nreads = 0;
            ho[0] = 0;
            if (upper > bcap) { err = "a read does not fit the staging buffer"; return 1; }
        }
        size_t p = seq_begin;
        while (p < seq_end) {
            nl = (const char *) memchr(d + p, '\n', seq_end - p);
            const size_t e = nl ? (size_t) (nl - d) : seq_end;
            memcpy(hb + used, d + p, e - p);
            used += e - p;
            p = nl ? e + 1 : seq_end;
        }
        ho[++nreads] = used;
        i = seq_end;
    }
    if (have && commet_readset_stage_commit(rs, nreads)) { err = commet_last_error(); return 1; }
    return 0;
}

// ---------------------------------------------------------------------------
// FASTQ and gzip inputs.  Format sniffing as the reference does it
// (file_manager.h:125-157): first byte '>' = FASTA, '@' = FASTQ, anything else is
// handed to zlib and sniffed again after inflation.
//   FASTQ record (fastq_file.h:139-190): header line (blank lines before it are
//   skipped), the NEXT line verbatim is the sequence, then the '+' line and the
//   quality line (blank lines before each skipped); #records = non-empty lines / 4
//   (fastq_file.h:60-67).
// ---------------------------------------------------------------------------
enum class ReadFormat { Fasta, Fastq, Unknown };

inline ReadFormat sniff_format(const char *d, size_t n)
{
    if (n && d[0] == '>') return ReadFormat::Fasta;
    if (n && d[0] == '@') return ReadFormat::Fastq;
    return ReadFormat::Unknown;
}

// whole-file inflate (gzopen also passes plain files through, like the reference's gz path)
inline bool inflate_file(const std::string &path, std::vector<char> &out)
{
    gzFile f = gzopen(path.c_str(), "r");
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    out.clear();
    std::vector<char> buf(8 << 20);
    for (;;) {
        const int got = gzread(f, buf.data(), (unsigned) buf.size());
        if (got < 0) {
            gzclose(f);
            return false;
        }
        if (got == 0) break;
        out.insert(out.end(), buf.data(), buf.data() + got);
    }
    gzclose(f);
    return true;
}

// A read file held in memory: mapped when plain, inflated when gzipped.
class ReadFileData {
public:
    bool open_file(const std::string &path)
    {
        if (!map_.open_file(path)) return false;
        d_ = map_.data();
        n_ = map_.size();
        fmt_ = sniff_format(d_, n_);
        if (fmt_ == ReadFormat::Unknown) {
            if (!inflate_file(path, inflated_)) return false;
            map_.close_file();
            d_ = inflated_.data();
            n_ = inflated_.size();
            fmt_ = sniff_format(d_, n_);
            gz_ = true;
        }
        return true;
    }
    const char *data() const { return d_; }
    size_t size() const { return n_; }
    ReadFormat format() const { return fmt_; }
    bool gzipped() const { return gz_; }

private:
    MappedFile map_;
    std::vector<char> inflated_;
    const char *d_ = nullptr;
    size_t n_ = 0;
    ReadFormat fmt_ = ReadFormat::Unknown;
    bool gz_ = false;
};

inline uint64_t count_fastq_records(const char *d, size_t n)
{
    uint64_t lines = 0;
    size_t i = 0;
    while (i < n) {
        const char *nl = (const char *) memchr(d + i, '\n', n - i);
        const size_t e = nl ? (size_t) (nl - d) : n;
        if (e > i) ++lines;
        i = nl ? e + 1 : n;
    }
    return lines / 4;
}

inline uint64_t count_records(ReadFormat fmt, const char *d, size_t n)
{
    return fmt == ReadFormat::Fastq ? count_fastq_records(d, n) : count_fasta_records(d, n);
}

// next line [b, e) starting at i; returns false at end of data
inline bool next_line(const char *d, size_t n, size_t &i, size_t &b, size_t &e)
{
    if (i >= n) return false;
    const char *nl = (const char *) memchr(d + i, '\n', n - i);
    b = i;
    e = nl ? (size_t) (nl - d) : n;
    i = nl ? e + 1 : n;
    return true;
}

// Calls f(seq_begin, seq_len) for the first `limit` FASTQ records.
template <typename F>
inline void for_each_fastq_record(const char *d, size_t n, uint64_t limit, F &&f)
{
    size_t i = 0, b, e;
    for (uint64_t r = 0; r < limit; ++r) {
        bool ok;
        while ((ok = next_line(d, n, i, b, e)) && e == b) {}   // header, skipping blank lines
        if (!ok) return;
        size_t sb = 0, se = 0;
        if (!next_line(d, n, i, sb, se)) sb = se = 0;          // sequence line, verbatim
        while ((ok = next_line(d, n, i, b, e)) && e == b) {}   // '+'
        if (ok) while ((ok = next_line(d, n, i, b, e)) && e == b) {}   // quality
        f(d + sb, se - sb);
    }
}

// Streams every record of a FASTQ buffer into rs.
inline int stream_fastq(commet_readset *rs, const char *d, size_t n, uint64_t n_records, std::string &err)
{
    uint8_t *hb = nullptr;
    uint64_t *ho = nullptr;
    uint64_t bcap = 0, rcap = 0, used = 0, nreads = 0;
    bool have = false;
    int rc = 0;
    for_each_fastq_record(d, n, n_records, [&](const char *s, size_t len) {
        if (rc) return;
        if (have && (nreads >= rcap || used + len > bcap)) {
            if (commet_readset_stage_commit(rs, nreads)) { err = commet_last_error(); rc = 1; return; }
            have = false;
        }
        if (!have) {
            if (commet_readset_stage_acquire(rs, &hb, &bcap, &ho, &rcap)) { err = commet_last_error(); rc = 1; return; }
            have = true;
            used = 0;
            nreads = 0;
            ho[0] = 0;
            if (len > bcap) { err = "a read does not fit the staging buffer"; rc = 1; return; }
        }
        memcpy(hb + used, s, len);
        used += len;
        ho[++nreads] = used;
    });
    if (!rc && have && commet_readset_stage_commit(rs, nreads)) { err = commet_last_error(); rc = 1; }
    return rc;
}

inline int stream_records(commet_readset *rs, ReadFormat fmt, const char *d, size_t n, uint64_t n_records, std::string &err)
{
    return fmt == ReadFormat::Fastq ? stream_fastq(rs, d, n, n_records, err) : stream_fasta(rs, d, n, err);
}

}  // namespace commet_host
