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

#include <cstdint>
#include <cstring>
#include <string>

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

}  // namespace commet_host
