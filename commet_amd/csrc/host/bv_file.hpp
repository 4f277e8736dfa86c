// bv_file.hpp — the `.bv` bit-vector file format, byte for byte
// (reference: BooleanVector::print / ::read, include/boolean_vector.h:302-414):
//   "<comment>\n#<N>\n" followed by N/8+1 raw bytes, read i = byte i/8, mask 1<<(i%8);
//   written O_CREAT|O_TRUNC with mode 0600.
#pragma once

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

namespace commet_host {

struct BitVector {
    uint64_t size = 0;              // bits
    std::vector<uint8_t> bytes;     // size/8 + 1
    std::string comment;

    void init_false(uint64_t n)
    {
        size = n;
        bytes.assign(n / 8 + 1, 0);
    }
    void init_true(uint64_t n)      // boolean_vector.h:148-164: padding bits cleared
    {
        size = n;
        bytes.assign(n / 8 + 1, 0xFF);
        for (uint64_t i = n; i < bytes.size() * 8; ++i) bytes[i >> 3] &= (uint8_t) ~(1u << (i & 7));
    }
    bool get(uint64_t i) const { return (bytes[i >> 3] >> (i & 7)) & 1; }
    void set(uint64_t i) { bytes[i >> 3] |= (uint8_t) (1u << (i & 7)); }
    uint64_t nb_one() const         // boolean_vector.h:236-264: all bytes, capped at size
    {
        uint64_t r = 0;
        for (uint8_t b : bytes) r += (uint64_t) __builtin_popcount(b);
        return r > size ? size : r;
    }
};

inline bool write_bv(const std::string &path, const BitVector &bv)
{
    const int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, (mode_t) 0600);
    if (fd == -1) {
        std::cerr << "Error opening file " << path << " -> exit\n";
        return false;
    }
    const std::string head = bv.comment + "\n#" + std::to_string(bv.size) + "\n";
    bool ok = write(fd, head.data(), head.size()) == (ssize_t) head.size();
    ok = ok && write(fd, bv.bytes.data(), bv.bytes.size()) == (ssize_t) bv.bytes.size();
    close(fd);
    if (!ok) std::cerr << "Error writing last byte of " << path << " -> exit\n";
    return ok;
}

inline bool read_bv(const std::string &path, BitVector &bv)
{
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd == -1) {
        std::cerr << "Error opening file " << path << " -> exit\n";
        return false;
    }
    struct stat sb;
    if (fstat(fd, &sb) == -1) {
        std::cerr << "Error getting statistics from file " << path << " -> exit\n";
        close(fd);
        return false;
    }
    std::string data((size_t) sb.st_size, '\0');
    size_t got = 0;
    while (got < data.size()) {
        const ssize_t r = read(fd, &data[got], data.size() - got);
        if (r <= 0) break;
        got += (size_t) r;
    }
    close(fd);
    size_t i = 0;
    while (i < data.size() && data[i] != '#') ++i;          // boolean_vector.h:384-387
    bv.comment = data.substr(0, i ? i - 1 : 0);
    ++i;
    std::string num;
    while (i < data.size() && data[i] != '\n') num += data[i++];
    ++i;
    if (num.empty()) {
        std::cerr << "Error, boolean vector does not contain its size\n";
        return false;
    }
    bv.init_false((uint64_t) atoi(num.c_str()));            // boolean_vector.h:398 (atoi)
    if (i < data.size()) memcpy(bv.bytes.data(), data.data() + i, std::min(bv.bytes.size(), data.size() - i));
    return true;
}

}  // namespace commet_host
