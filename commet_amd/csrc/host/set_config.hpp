// set_config.hpp — the set-config grammar `name:file[,bv];file[,bv]...`
// (reference: read_sets, include/set_parser.h:46-102; README:92-108).
//   - one set per non-empty line; tag = text before the first ':' (NOT trimmed),
//     "SET<n>" when the line has no ':' (n counts non-empty lines)
//   - files split on ';', optional ",bv" after the first ','; only ' ' is
//     trimmed, at both ends (set_parser.h:32-40); the last piece is always kept
//   - sets come back ordered by tag (std::map), a repeated tag replaces the
//     earlier line
#pragma once

#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <vector>

namespace commet_host {

struct SetEntry {
    std::string file;
    std::string bv;   // empty when absent
};

using SetMap = std::map<std::string, std::vector<SetEntry>>;

inline void trim_spaces(std::string &s)
{
    size_t b = 0, e = s.size();
    while (b < e && s[b] == ' ') ++b;
    while (e > b && s[e - 1] == ' ') --e;
    s = s.substr(b, e - b);
}

inline SetEntry parse_entry(std::string item)
{
    SetEntry en;
    trim_spaces(item);
    const size_t comma = item.find(',');
    if (comma != std::string::npos) {
        en.bv = item.substr(comma + 1);
        trim_spaces(en.bv);
        item = item.substr(0, comma);
        trim_spaces(item);
    }
    en.file = item;
    return en;
}

// returns false (after the reference's message) when the file cannot be read
inline bool read_sets(const std::string &path, SetMap &sets)
{
    sets.clear();
    std::ifstream in(path.c_str());
    if (!in.good()) {
        std::cerr << "Cannot read file " << path << "\n";
        return false;
    }
    int nb_sets = 0;
    std::string line;
    while (in.good()) {
        std::getline(in, line);
        if (line.empty()) continue;
        ++nb_sets;
        std::string tag;
        const size_t colon = line.find(':');
        if (colon != std::string::npos) {
            tag = line.substr(0, colon);
            line = line.substr(colon + 1);
        } else {
            tag = "SET" + std::to_string(nb_sets);
        }
        std::vector<SetEntry> entries;
        size_t semi;
        while (!line.empty() && (semi = line.find(';')) != std::string::npos) {
            entries.push_back(parse_entry(line.substr(0, semi)));
            line = line.substr(semi + 1);
        }
        entries.push_back(parse_entry(line));
        sets[tag] = entries;
    }
    return true;
}

}  // namespace commet_host
