// index_and_search — drop-in replacement of Commet's tool of the same name
// (reference: src/index_and_search.cpp:56-422), driving the MI355X path through
// the C ABI of include/commet_hip.h.  Same flags, same set-config grammar, same
// stdout banners, same OUT/<file>_in_<set>.bv bytes and LOG/<s>_in_<i>.log
// format, so Commet.py (Commet.py:197,220,233) can call it unchanged.
//
// Host side = this file: argv, set-configs, FASTA / FASTQ (plain or gzip) -> pinned staging batches
// (fasta_source.hpp), filter / output .bv files (bv_file.hpp).  Everything
// between "reads are resident" and "tag bits are back" runs in the library.
//
// Extra (non-reference) controls: env COMMET_DEVICE=<n> picks the GPU.

#include <dirent.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <sys/un.h>
#include <sys/wait.h>
#include <cerrno>
#include <csignal>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <future>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/commet_hip.h"
#include "bv_file.hpp"
#include "fasta_source.hpp"
#include "set_config.hpp"

using namespace commet_host;

static const std::string version = "2.1";   // index_and_search.cpp:44 (interface version we mirror)

// The tool's exits become exceptions so that one process can serve many invocations (resident mode below).
struct ToolExit {
    int code;
};
#define TOOL_EXIT(c) throw ToolExit{c}

// ---- resident mode ---------------------------------------------------------------------------------------------
// Commet.py starts one index_and_search process per job (Commet.py:197,220,233): N^2 - 1 HIP start-ups, and every set is
// parsed and uploaded again each time it takes part in a job (2(N - 1) + ... times).  `index_and_search --serve SOCKET`
// keeps ONE process with its contexts and the read sets it has loaded (keyed by path, size and mtime of their files) in
// HBM; an ordinary invocation with COMMET_SERVER=SOCKET in its environment forwards argv and its working directory,
// and prints what comes back: same stdout / stderr bytes, same files, same exit code, Commet.py unchanged.
struct CachedSet {
    commet_readset *rs = nullptr;
    std::vector<uint64_t> file_reads;
    uint64_t bytes = 0, last_use = 0;
    bool in_use = false;
};
struct ServerState {
    std::map<std::pair<int, int>, commet_ctx *> ctxs;                            // (k, t) -> context
    std::map<std::pair<commet_ctx *, std::string>, CachedSet> sets;              // (context, files key) -> resident set
    uint64_t tick = 0, hits = 0, loads = 0, evictions = 0, requests = 0;
    uint64_t cached_bytes = 0, budget_bytes = 0;                                 // budget: COMMET_SERVER_CACHE_GB (default 96)
    // makes room for `need` more bytes of resident sets: by the budget on their estimated footprint, and by what the
    // device really has left (a set's footprint grows after it was cached: per-read counts, the tiled search's query list)
    void evict_until(uint64_t need, commet_ctx *ctx)
    {
        for (;;) {
            uint64_t free_b = ~0ull, total_b = 0;
            if (ctx) (void) commet_device_memory(ctx, &free_b, &total_b);
            const bool tight = ctx && free_b < std::max<uint64_t>(4 * need, total_b / 4);
            // what the sets hold now: their packed reads (estimated when they were cached) + the query lists the tiled
            // search has made for them since (commet_readset_cache_bytes: several times the set itself)
            uint64_t held = cached_bytes;
            for (auto &kv : sets) held += commet_readset_cache_bytes(kv.second.rs);
            if (held + need <= budget_bytes && !tight) return;
            auto victim = sets.end();
            for (auto it = sets.begin(); it != sets.end(); ++it)
                if (!it->second.in_use && (victim == sets.end() || it->second.last_use < victim->second.last_use)) victim = it;
            if (victim == sets.end()) return;
            commet_readset_destroy(victim->second.rs);
            cached_bytes -= victim->second.bytes;
            sets.erase(victim);
            ++evictions;
        }
    }
};
static ServerState *g_server = nullptr;

static void print_usage()
{
    // index_and_search.cpp:407-422
    std::cerr << "\nindex_and_search, version " << version << "\n";
    std::cerr << "Usage : ./index_and_search -i <file> -s <file> [options]\n";
    std::cerr << "Mandatory:\n";
    std::cerr << "\t -i <file>: A file containing the list of files to index - MANDATORY\n";
    std::cerr << "\t -s <file>: A file containing the list of files to search - MANDATORY\n";
    std::cerr << "\t            Each line of the file corresponds to a set of files to search\n";
    std::cerr << "Options:\n";
    std::cerr << "\t -l </.../>: ABSOLUTE path to log folder\n";
    std::cerr << "\t -o </.../>: ABSOLUTE path to output folder\n";
    std::cerr << "\t -k <value>: Size of k-mers (value of k). [default=33]\n";
    std::cerr << "\t -t <value>: Number of shared k-mers. [default=2]\n";
    std::cerr << "\t -f: Full comparison of index set and the first searched set [default=false]\n";
    std::cerr << "\t -h: Prints this message\n";
    std::cerr << "\t -v: Prints the version number\n";
}

static void ensure_dir(const std::string &p)
{
    struct stat info;   // index_and_search.cpp:178-191
    if (stat(p.c_str(), &info) != 0) mkdir(p.c_str(), S_IRWXU | S_IRGRP | S_IXGRP);
    else if (!(info.st_mode & S_IFDIR)) {
        std::cerr << "Error: " << p << " already exists and is not a directory\n";
        TOOL_EXIT(1);
    }
}

struct LoadedFile {
    std::string name;
    uint64_t nb_reads = 0;
    BitVector filter;     // ReadFile::bv
};

struct LoadedSet {
    std::string nickname;
    std::vector<LoadedFile> files;
    commet_readset *rs = nullptr;
    std::vector<uint8_t> select;   // set-wide input-filter bits
    bool any_bv = false;
    uint64_t n_reads = 0;
    CachedSet *cached = nullptr;   // resident mode: the set belongs to the server's cache

    LoadedSet() = default;
    LoadedSet(const LoadedSet &) = delete;
    LoadedSet &operator=(const LoadedSet &) = delete;
    LoadedSet(LoadedSet &&o) noexcept { *this = std::move(o); }
    LoadedSet &operator=(LoadedSet &&o) noexcept
    {
        if (this != &o) {
            release();
            nickname = std::move(o.nickname), files = std::move(o.files), select = std::move(o.select);
            rs = o.rs, any_bv = o.any_bv, n_reads = o.n_reads, cached = o.cached;
            o.rs = nullptr, o.cached = nullptr;
        }
        return *this;
    }
    // a loaded set is given back: destroyed, or (resident mode) left in the cache for the next invocation.  Also on the
    // way out of a request that fails (TOOL_EXIT unwinds through here): a set that was loaded but never reached the
    // cache would otherwise keep its gigabytes of HBM for the life of the server.
    void release()
    {
        if (cached) cached->in_use = false;
        else if (rs) commet_readset_destroy(rs);
        rs = nullptr;
        cached = nullptr;
    }
    ~LoadedSet() { release(); }
};

static void build_select(LoadedSet &out);

static void release_set(LoadedSet &ls) { ls.release(); }

// FileManager::addFile for every entry of one set (file_manager.h:117-216),
// FastaFile ctors (fasta_file.h:49-116), then streams the reads to HBM.
static void load_set(commet_ctx *ctx, const std::string &nickname, const std::vector<SetEntry> &entries, LoadedSet &out)
{
    out.nickname = nickname;
    // resident mode: a set whose files (path, size, mtime) are the ones of a cached set is not opened again
    std::string files_key;
    CachedSet *hit = nullptr;
    if (g_server) {
        bool all = true;
        for (const SetEntry &en : entries) {
            struct stat sb;
            char rp[PATH_MAX];
            if (stat(en.file.c_str(), &sb) != 0 || !realpath(en.file.c_str(), rp)) {
                all = false;
                break;
            }
            files_key += std::string(rp) + "|" + std::to_string((unsigned long long) sb.st_size) + "|" +
                         std::to_string((long long) sb.st_mtim.tv_sec) + "." + std::to_string((long) sb.st_mtim.tv_nsec) + ";";
        }
        if (!all) files_key.clear();
        else {
            auto it = g_server->sets.find(std::make_pair(ctx, files_key));
            if (it != g_server->sets.end() && !it->second.in_use) hit = &it->second;
        }
    }
    if (hit) {
        std::vector<bool> given;
        for (const SetEntry &en : entries) {
            if (en.bv.empty()) std::cout << "open " << en.file << "\n";
            else std::cout << "open " << en.file << "," << en.bv << "\n";
            LoadedFile lf;
            lf.name = en.file;
            if (!en.bv.empty()) {
                if (!read_bv(en.bv, lf.filter)) TOOL_EXIT(1);
                out.any_bv = true;
            }
            out.files.push_back(lf);
            given.push_back(!en.bv.empty());
        }
        out.rs = hit->rs;
        out.cached = hit;
        hit->in_use = true;
        hit->last_use = ++g_server->tick;
        ++g_server->hits;
        out.n_reads = commet_readset_num_reads(out.rs);
        for (size_t i = 0; i < out.files.size(); ++i) {
            LoadedFile &lf = out.files[i];
            lf.nb_reads = hit->file_reads[i];
            if (!given[i]) lf.filter.init_true(lf.nb_reads);
            else if (lf.nb_reads != lf.filter.size) {   // fasta_file.h:108-111
                std::cerr << "Number of reads in " << lf.name << " and boolean vector size are not equal -> quit\n";
                TOOL_EXIT(1);
            }
        }
        build_select(out);
        return;
    }
    std::vector<std::unique_ptr<ReadFileData>> maps;
    std::vector<bool> bv_given;
    // the files of a set are opened (mapped, or inflated when gzipped: one zlib stream each) by one thread per file;
    // messages and error handling stay in file order
    std::vector<std::future<std::unique_ptr<ReadFileData>>> opening;
    for (const SetEntry &en : entries) {
        const std::string path = en.file;
        opening.push_back(std::async(std::launch::async, [path]() {
            std::unique_ptr<ReadFileData> f(new ReadFileData);
            if (!f->open_file(path)) f.reset();
            return f;
        }));
    }
    size_t entry_no = 0;
    for (const SetEntry &en : entries) {
        if (en.bv.empty()) std::cout << "open " << en.file << "\n";
        else std::cout << "open " << en.file << "," << en.bv << "\n";
        std::unique_ptr<ReadFileData> mf = opening[entry_no++].get();
        if (!mf) {
            if (en.bv.empty()) {
                std::cerr << "Cannot open file file " << en.file << " -> ignore\n";   // file_manager.h:121-123
                std::cerr << "Cannot open file " << en.file << " -> ignore\n";        // gz path, :144-147
                TOOL_EXIT(1);
            }
            std::cerr << "Cannot open file " << en.file << " -> ignore\n";            // file_manager.h:177-180
            continue;
        }
        if (mf->format() == ReadFormat::Unknown) {   // neither '>' nor '@', plain or gzipped (file_manager.h:154-156)
            std::cerr << "Unknown format: " << en.file << " -> ignore\n";
            TOOL_EXIT(1);
        }
        LoadedFile lf;
        lf.name = en.file;
        if (!en.bv.empty()) {
            if (!read_bv(en.bv, lf.filter)) TOOL_EXIT(1);
            out.any_bv = true;
        }
        out.files.push_back(lf);
        bv_given.push_back(!en.bv.empty());
        maps.push_back(std::move(mf));
    }
    // parse + upload all files of the set (several host threads, pinned staging, hipMemcpyAsync + packing kernel)
    std::vector<const char *> data;
    std::vector<uint64_t> sizes;
    for (const std::unique_ptr<ReadFileData> &mf : maps) {
        data.push_back(mf->data());
        sizes.push_back(mf->size());
    }
    out.rs = commet_readset_from_buffers(ctx, data.data(), sizes.data(), (int) maps.size());
    if (!out.rs) {
        std::cerr << "Error: " << commet_last_error() << "\n";
        TOOL_EXIT(1);
    }
    if (commet_readset_finalize(out.rs)) {
        std::cerr << "Error: " << commet_last_error() << "\n";
        TOOL_EXIT(1);
    }
    out.n_reads = commet_readset_num_reads(out.rs);
    if (g_server && !files_key.empty() && maps.size() == entries.size() &&
        g_server->sets.find(std::make_pair(ctx, files_key)) == g_server->sets.end()) {
        // the set stays resident for later invocations (bytes: planes + offsets + counts + bitmaps, roughly)
        uint64_t bases = 0;
        for (uint64_t sz : sizes) bases += sz;
        CachedSet cs;
        cs.rs = out.rs;
        cs.bytes = bases / 2 + out.n_reads * 40 + (1u << 20);
        for (size_t i = 0; i < out.files.size(); ++i) cs.file_reads.push_back(commet_readset_file_reads(out.rs, i));
        g_server->evict_until(cs.bytes, ctx);
        cs.in_use = true;
        cs.last_use = ++g_server->tick;
        g_server->cached_bytes += cs.bytes;
        ++g_server->loads;
        out.cached = &(g_server->sets[std::make_pair(ctx, files_key)] = cs);
    }
    for (size_t i = 0; i < out.files.size(); ++i) {
        LoadedFile &lf = out.files[i];
        lf.nb_reads = commet_readset_file_reads(out.rs, i);
        if (!bv_given[i]) lf.filter.init_true(lf.nb_reads);
        else if (lf.nb_reads != lf.filter.size) {   // fasta_file.h:108-111
            std::cerr << "Number of reads in " << lf.name << " and boolean vector size are not equal -> quit\n";
            TOOL_EXIT(1);
        }
    }
    build_select(out);
}

// set-wide select bits = the per-file filters, concatenated
static void build_select(LoadedSet &out)
{
    out.select.assign(out.n_reads / 8 + 1, 0);
    if (!out.any_bv) {   // every read of every file: whole bytes at once
        std::fill(out.select.begin(), out.select.begin() + out.n_reads / 8, (uint8_t) 0xFF);
        for (uint64_t i = (out.n_reads / 8) * 8; i < out.n_reads; ++i) out.select[i >> 3] |= (uint8_t) (1u << (i & 7));
        return;
    }
    uint64_t pos = 0;
    for (const LoadedFile &lf : out.files) {
        for (uint64_t i = 0; i < lf.nb_reads; ++i)
            if (lf.filter.get(i)) out.select[(pos + i) >> 3] |= (uint8_t) (1u << ((pos + i) & 7));
        pos += lf.nb_reads;
    }
}

static int run_tool(int argc, char **argv)
{
    std::string search_file_list, index_file_list;
    int kmer_size = 33;   // index_and_search.cpp:71-72
    int min_hits = 2;
    std::string log_path = ".", out_path = ".";
    bool full = false;

    if (argc == 1) {
        print_usage();
        return 0;
    }
    int arg_pos = 1;
    while (arg_pos < argc) {   // index_and_search.cpp:85-172
        const std::string flag = argv[arg_pos];
        auto need_arg = [&]() {
            ++arg_pos;
            if (arg_pos >= argc) {
                std::cerr << "Error, flag " << argv[arg_pos - 1] << " needs an argument\n";
                print_usage();
                TOOL_EXIT(1);
            }
        };
        if (flag == "-i") {
            need_arg();
            if (!index_file_list.empty()) std::cerr << "index files already given (-i) -> ignore";
            else index_file_list = argv[arg_pos];
        } else if (flag == "-s") {
            need_arg();
            if (!search_file_list.empty()) std::cerr << "search files already given (-s) -> ignore";
            else search_file_list = argv[arg_pos];
        } else if (flag == "-l") {
            need_arg();
            log_path = argv[arg_pos];
        } else if (flag == "-o") {
            need_arg();
            out_path = argv[arg_pos];
        } else if (flag == "-k") {
            need_arg();
            kmer_size = atoi(argv[arg_pos]);
            std::cout << "k-mer size (-k) = " << kmer_size << "\n";
        } else if (flag == "-t") {
            need_arg();
            min_hits = atoi(argv[arg_pos]);
            std::cout << "min hits (-t) = " << min_hits << "\n";
        } else if (flag == "-f") {
            full = true;
        } else if (flag == "-h") {
            print_usage();
            return 0;
        } else if (flag == "-v") {
            std::cout << "\nindex_and_search version " << version << "\n";
            return 0;
        } else {
            std::cerr << "Unknown option " << flag << "\n";
            print_usage();
            return 0;
        }
        ++arg_pos;
    }
    ensure_dir(log_path);
    ensure_dir(out_path);

    const auto start_time = std::chrono::steady_clock::now();

    SetMap index_sets, search_sets;
    if (!read_sets(index_file_list, index_sets)) TOOL_EXIT(1);
    if (index_sets.size() != 1) {   // index_and_search.cpp:197-200
        std::cerr << "Only one set of files is allowed for indexing\n";
        TOOL_EXIT(1);
    }

    // COMMET_INGEST_VERBOSE: wall time of the tool's phases on stderr
    const bool phase_verbose = getenv("COMMET_INGEST_VERBOSE") != nullptr;
    auto phase_t = std::chrono::steady_clock::now();
    auto phase = [&](const char *what) {
        const auto now = std::chrono::steady_clock::now();
        if (phase_verbose) fprintf(stderr, "[tool] %-24s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - phase_t).count());
        phase_t = now;
    };
    const char *dev_env = getenv("COMMET_DEVICE");
    commet_ctx *ctx = nullptr;
    if (g_server) {   // resident mode: one context per (k, t), kept
        commet_ctx *&slot = g_server->ctxs[std::make_pair(kmer_size, min_hits)];
        if (!slot) slot = commet_create(dev_env ? atoi(dev_env) : 0, kmer_size, min_hits);
        ctx = slot;
        if (!ctx) g_server->ctxs.erase(std::make_pair(kmer_size, min_hits));
    } else {
        ctx = commet_create(dev_env ? atoi(dev_env) : 0, kmer_size, min_hits);
    }
    phase("context (HIP start-up)");
    if (!ctx) {
        std::cerr << commet_last_error() << "\n";
        TOOL_EXIT(1);
    }

    LoadedSet index_set;
    load_set(ctx, index_sets.begin()->first, index_sets.begin()->second, index_set);

    if (!read_sets(search_file_list, search_sets)) TOOL_EXIT(1);
    // -f: only the first search set is opened (index_and_search.cpp:231-233)
    std::vector<LoadedSet> searches(full && !search_sets.empty() ? 1 : search_sets.size());
    {
        size_t s = 0;
        for (SetMap::iterator it = search_sets.begin(); it != search_sets.end() && s < searches.size(); ++it, ++s)
            load_set(ctx, it->first, it->second, searches[s]);
    }
    phase("load sets");
    if (searches.empty()) {
        // the reference dereferences search_sets[0] here (index_and_search.cpp:247): undefined
        std::cerr << "Error: no set to search\n";
        TOOL_EXIT(1);
    }

    // the chunk loop (index_and_search.cpp:241-277) on the device
    const int ns = (int) searches.size();
    std::vector<const commet_readset *> rs(ns);
    std::vector<const uint8_t *> sel(ns);
    std::vector<std::vector<uint8_t>> tags(ns);
    std::vector<uint8_t *> tag_ptr(ns);
    for (int s = 0; s < ns; ++s) {
        rs[s] = searches[s].rs;
        sel[s] = searches[s].any_bv ? searches[s].select.data() : nullptr;
        tags[s].assign(searches[s].n_reads / 8 + 1, 0);
        tag_ptr[s] = tags[s].data();
    }
    std::vector<commet_pair_stats> stats(ns);
    commet_job_info info;
    if (commet_index_and_search(ctx, index_set.rs, index_set.any_bv ? index_set.select.data() : nullptr, ns, rs.data(),
                                sel.data(), tag_ptr.data(), stats.data(), &info)) {
        std::cerr << "Error: " << commet_last_error() << "\n";
        TOOL_EXIT(1);
    }

    phase("index_and_search");
    // per-chunk banners (index_and_search.cpp:267-269)
    for (uint64_t c = 0; c < info.n_chunks; ++c)
        for (int s = 0; s < ns; ++s) {
            std::cout << "\n------------------------------------------------------------------\n";
            std::cout << "finding reads from {" << searches[s].nickname << "} present in raw {" << index_set.nickname
                      << "}\n";
            std::cout << "------------------------------------------------------------------\n";
        }

    const float index_s = (float) (info.index_ms / 1000.0);
    for (int s = 0; s < ns; ++s) {   // index_and_search.cpp:278-301
        const float search_s = (float) (stats[s].search_ms / 1000.0);
        auto total_s = [&]() {
            return (float) std::chrono::duration<double>(std::chrono::steady_clock::now() - start_time).count();
        };
        std::cout << "\n------------------------------------------------------------------\n";
        std::cout << "Reads from {" << searches[s].nickname << "} present in raw {" << index_set.nickname << "}\n";
        std::cout << "------------------------------------------------------------------\n";
        std::cout << "Index  time: " << index_s << " s\n";
        std::cout << "Search time: " << search_s << " s\n";
        std::cout << "Total  time: " << total_s() << " s\n";
        std::cout << "[indexed " << stats[s].indexed << ", searched " << stats[s].searched << ", shared "
                  << stats[s].shared << "]\n";
        const std::string fname = log_path + "/" + searches[s].nickname + "_in_" + index_set.nickname + ".log";
        std::ofstream log_file(fname.c_str());
        if (!log_file.good()) {
            std::cerr << "Cannot open log file : " << fname << "\n";
            TOOL_EXIT(1);
        }
        log_file << "Index  time: " << index_s << " s\n";
        log_file << "Search time: " << search_s << " s\n";
        log_file << "Total  time: " << total_s() << " s\n";
        log_file << "[indexed " << stats[s].indexed << ", searched " << stats[s].searched << ", shared "
                 << stats[s].shared << "]\n";
        log_file.close();
    }

    // FileManager::save_bv (file_manager.h:245-252): one .bv per file of `set`, bits taken from set-wide tags
    auto save_bv = [&](const LoadedSet &set, const std::vector<uint8_t> &set_tags, const std::string &suffix) {
        uint64_t pos = 0;
        for (const LoadedFile &lf : set.files) {
            BitVector bv;
            bv.init_false(lf.nb_reads);
            for (uint64_t i = 0; i < lf.nb_reads; ++i)
                if ((set_tags[(pos + i) >> 3] >> ((pos + i) & 7)) & 1) bv.set(i);
            pos += lf.nb_reads;
            const std::string base = lf.name.substr(lf.name.rfind("/") + 1);
            bv.comment = lf.name + " in " + suffix;
            if (!write_bv(out_path + "/" + base + "_in_" + suffix + ".bv", bv)) TOOL_EXIT(1);
        }
    };

    if (full) {
        // Full comparison on the first search set (index_and_search.cpp:304-391): A = index set, B = search set.
        //   pass 1 (above)  B in A                                   -> T1
        //   pass 2          A in (B restricted to T1)                -> T2, written as <A files>_in_<B>.bv
        //   pass 3          (B restricted to T1) in (A restricted to T2) -> written as <B files>_in_<A>.bv
        LoadedSet &A = index_set, &B = searches[0];
        auto popcount = [](const std::vector<uint8_t> &bits, uint64_t n) {
            uint64_t c = 0;
            for (uint64_t i = 0; i < n; ++i) c += (bits[i >> 3] >> (i & 7)) & 1;
            return c;
        };
        const uint64_t nb_reads_A = popcount(A.select, A.n_reads), nb_reads_B = popcount(B.select, B.n_reads);
        const std::vector<uint8_t> T1 = tags[0];
        auto one_pass = [&](const LoadedSet &idx, const uint8_t *idx_sel, const LoadedSet &srch, const uint8_t *srch_sel,
                            std::vector<uint8_t> &out_tags, const std::string &banner, const std::string &log_name,
                            uint64_t denom, bool save_now, const std::string &save_suffix) {
            std::ofstream log_file(log_name.c_str());
            if (!log_file.good()) {
                std::cerr << "Cannot open log file " << log_name << " -> exit\n";
                TOOL_EXIT(1);
            }
            std::cout << "\n------------------------------------------------------------------\n";
            std::cout << banner << "\n";
            std::cout << "------------------------------------------------------------------\n";
            const auto t0 = std::chrono::steady_clock::now();
            const commet_readset *q = srch.rs;
            out_tags.assign(srch.n_reads / 8 + 1, 0);
            uint8_t *tp = out_tags.data();
            commet_pair_stats st;
            commet_job_info inf;
            if (commet_index_and_search(ctx, idx.rs, idx_sel, 1, &q, &srch_sel, &tp, &st, &inf)) {
                std::cerr << "Error: " << commet_last_error() << "\n";
                TOOL_EXIT(1);
            }
            if (save_now) save_bv(srch, out_tags, save_suffix);
            const float tot = (float) std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            for (std::ostream *os : {(std::ostream *) &std::cout, (std::ostream *) &log_file}) {
                *os << "Index  time: " << (float) (inf.index_ms / 1000.0) << " s\n";
                *os << "Search time: " << (float) (st.search_ms / 1000.0) << " s\n";
                *os << "Total  time: " << tot << " s\n";
                *os << "[indexed " << st.indexed << ", searched " << st.searched << ", shared " << st.shared << "]\n"
                    << 100 * (float) st.shared / (float) denom << "%\n";
            }
        };
        std::vector<uint8_t> T2, T3;
        one_pass(B, T1.data(), A, A.select.data(), T2,
                 "finding reads from {" + A.nickname + "} present in {raw {" + B.nickname + "} present in raw {" + A.nickname + "}}",
                 log_path + "/" + A.nickname + "_in_" + B.nickname + ".log", nb_reads_A, true, B.nickname);
        one_pass(A, T2.data(), B, T1.data(), T3,
                 "finding reads from {" + B.nickname + "} present in {raw {" + A.nickname + "} present in {raw {" + B.nickname +
                     "} present in raw {" + A.nickname + "}}}",
                 log_path + "/" + B.nickname + "_in_" + A.nickname + ".log", nb_reads_B, true, A.nickname);
        tags[0] = T3;
    }

    // save_bv (index_and_search.cpp:397-399)
    for (int s = 0; s < ns; ++s) save_bv(searches[s], tags[s], index_set.nickname);
    phase("logs + .bv files");

    // (ending the process without this teardown was tried: the driver then reclaims the memory while the next job's
    // HIP start-up waits for it — same wall time per job)
    for (LoadedSet &ls : searches) release_set(ls);
    release_set(index_set);
    if (!g_server) commet_destroy(ctx);
    phase("teardown");
    return 0;
}

// ---- resident mode: protocol ------------------------------------------------------------------------------------
// request : u32 n, then n strings (u32 length + bytes): working directory, argv[0], argv[1], ...
// reply   : i32 exit code, u64 + bytes of stdout, u64 + bytes of stderr
static bool read_all(int fd, void *buf, size_t n)
{
    char *p = (char *) buf;
    while (n) {
        const ssize_t r = read(fd, p, n);
        if (r <= 0) return false;
        p += r, n -= (size_t) r;
    }
    return true;
}
static bool write_all(int fd, const void *buf, size_t n)
{
    const char *p = (const char *) buf;
    while (n) {
        const ssize_t r = write(fd, p, n);
        if (r <= 0) return false;
        p += r, n -= (size_t) r;
    }
    return true;
}

static int serve(const char *sock_path)
{
    signal(SIGPIPE, SIG_IGN);
    ServerState state;
    const char *gb = getenv("COMMET_SERVER_CACHE_GB");
    state.budget_bytes = (uint64_t) (gb ? atof(gb) : 96.0) * (1ull << 30);
    g_server = &state;
    const int ls = socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un addr;
    memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX;
    if (ls < 0 || strlen(sock_path) >= sizeof addr.sun_path) {
        std::cerr << "Error: cannot create the server socket " << sock_path << "\n";
        return 1;
    }
    strcpy(addr.sun_path, sock_path);
    unlink(sock_path);
    if (bind(ls, (sockaddr *) &addr, sizeof addr) != 0 || listen(ls, 16) != 0) {
        std::cerr << "Error: cannot listen on " << sock_path << "\n";
        return 1;
    }
    char home[PATH_MAX];
    if (!getcwd(home, sizeof home)) strcpy(home, "/");
    std::cerr << "index_and_search: serving on " << sock_path << "\n";
    for (;;) {
        const int fd = accept(ls, nullptr, nullptr);
        if (fd < 0) {
            if (errno != EINTR) usleep(20000);   // (EMFILE and the like do not go away by asking again at once)
            continue;
        }
        uint32_t n = 0;
        std::vector<std::string> strs;
        bool ok = read_all(fd, &n, 4) && n >= 2 && n < 4096;
        for (uint32_t i = 0; ok && i < n; ++i) {
            uint32_t len = 0;
            ok = read_all(fd, &len, 4) && len < (1u << 20);
            if (!ok) break;
            std::string x(len, '\0');
            ok = len == 0 || read_all(fd, &x[0], len);
            strs.push_back(x);
        }
        if (!ok) {
            close(fd);
            continue;
        }
        int code = 0;
        std::ostringstream out, err;
        if (strs.size() == 3 && strs[2] == "--server-stop") {
            close(fd);
            break;
        }
        std::streambuf *old_out = std::cout.rdbuf(out.rdbuf()), *old_err = std::cerr.rdbuf(err.rdbuf());
        if (strs.size() == 3 && strs[2] == "--server-stats") {
            std::cout << "requests " << state.requests << ", sets resident " << state.sets.size() << " (" << state.cached_bytes / (1 << 20)
                      << " MiB), cache hits " << state.hits << ", loads " << state.loads << ", evictions " << state.evictions
                      << ", contexts " << state.ctxs.size() << "\n";
        } else {
            ++state.requests;
            std::vector<char *> av;
            for (size_t i = 1; i < strs.size(); ++i) av.push_back(&strs[i][0]);
            if (chdir(strs[0].c_str()) != 0) {
                std::cerr << "Error: cannot enter " << strs[0] << "\n";
                code = 1;
            } else {
                try {
                    code = run_tool((int) av.size(), av.data());
                } catch (const ToolExit &e) {
                    code = e.code;
                } catch (const std::exception &e) {
                    std::cerr << "Error: " << e.what() << "\n";
                    code = 1;
                }
            }
            for (auto &kv : state.sets) kv.second.in_use = false;   // (a request that failed half-way gives its sets back here)
            if (chdir(home) != 0) {}
        }
        std::cout.rdbuf(old_out);
        std::cerr.rdbuf(old_err);
        const std::string so = out.str(), se = err.str();
        const int32_t c32 = code;
        const uint64_t lo = so.size(), le = se.size();
        (void) (write_all(fd, &c32, 4) && write_all(fd, &lo, 8) && write_all(fd, so.data(), lo) && write_all(fd, &le, 8) &&
                write_all(fd, se.data(), le));
        close(fd);
    }
    for (auto &kv : state.sets) commet_readset_destroy(kv.second.rs);
    for (auto &kv : state.ctxs) commet_destroy(kv.second);
    close(ls);
    unlink(sock_path);
    g_server = nullptr;
    return 0;
}


// ---- resident mode on several GPUs --------------------------------------------------------------------------------------------
// `index_and_search --serve SOCKET --devices N` (N = a number, or "all" = the render nodes under /dev/dri; COMMET_SERVER_DEVICES
// does the same): the process forks one single-device server per GPU BEFORE anything in it has touched the GPU — child d serves
// SOCKET.d with the loop above on device d (COMMET_SERVER_DEVICE_LIST="0,0,1": the device of every child, for rehearsals on fewer
// GPUs) — and becomes a ROUTER that never makes a HIP call: it accepts the clients' requests on SOCKET, reads the job's two set
// configs (-i / -s, relative to the client's directory) and hands the request to the device on which most of the job's files are
// resident already; a job none of whose files it has seen goes to the device with the fewest requests in flight, then the fewest
// files.  One thread per client: jobs on different devices run at the same time — what Commet.py does with --sge, and what any
// caller that starts jobs in parallel gets; per device they queue, as on one GPU.  The reply's bytes are relayed unchanged.
// Commet.py's local mode starts its jobs one after the other (os.system): it gains the node's MEMORY from this (N x 96 GB of
// resident sets), not its kernels; the N x N driver (commet_amd/matrix.py) is the way to use all GPUs for one matrix.
struct Router {
    std::string sock;
    std::vector<std::string> child_sock;
    std::vector<pid_t> child_pid;
    std::mutex mu;
    std::map<std::string, std::set<int>> where;       // file (absolute path) -> devices it has been routed to
    std::vector<uint64_t> inflight, files_on, served;
    uint64_t requests = 0, routed_by_residency = 0;

    static std::string absolute(const std::string &cwd, const std::string &p) { return (!p.empty() && p[0] == '/') ? p : cwd + "/" + p; }

    // the files of the job's index and search sets (what a request's -i / -s name), as far as they can be read here
    static std::vector<std::string> job_files(const std::vector<std::string> &strs)
    {
        std::vector<std::string> out;
        for (size_t i = 2; i + 1 < strs.size(); ++i) {
            if (strs[i] != "-i" && strs[i] != "-s") continue;
            std::ifstream in(absolute(strs[0], strs[i + 1]).c_str());
            std::string line;
            while (in.good() && std::getline(in, line)) {
                const size_t colon = line.find(':');
                if (colon != std::string::npos) line = line.substr(colon + 1);
                std::stringstream ss(line);
                std::string item;
                while (std::getline(ss, item, ';')) {
                    const size_t comma = item.find(',');
                    if (comma != std::string::npos) item = item.substr(0, comma);
                    trim_spaces(item);
                    if (!item.empty()) out.push_back(absolute(strs[0], item));
                }
            }
        }
        return out;
    }

    int pick(const std::vector<std::string> &files)
    {
        std::lock_guard<std::mutex> lk(mu);
        const int n = (int) child_sock.size();
        std::vector<int> have(n, 0);
        for (const std::string &f : files) {
            auto it = where.find(f);
            if (it != where.end())
                for (int d : it->second) ++have[d];
        }
        int best = 0;
        for (int d = 1; d < n; ++d) {
            if (have[d] != have[best]) {
                if (have[d] > have[best]) best = d;
            } else if (inflight[d] != inflight[best]) {
                if (inflight[d] < inflight[best]) best = d;
            } else if (files_on[d] < files_on[best]) {
                best = d;
            }
        }
        if (have[best] > 0) ++routed_by_residency;
        for (const std::string &f : files)
            if (where[f].insert(best).second) ++files_on[best];
        ++inflight[best], ++served[best], ++requests;
        return best;
    }
};

static bool send_request(int fd, const std::vector<std::string> &strs)
{
    const uint32_t n = (uint32_t) strs.size();
    bool ok = write_all(fd, &n, 4);
    for (const std::string &x : strs) {
        const uint32_t len = (uint32_t) x.size();
        ok = ok && write_all(fd, &len, 4) && (len == 0 || write_all(fd, x.data(), len));
    }
    return ok;
}

static int connect_to(const std::string &path)
{
    const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un addr;
    memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX;
    if (fd < 0 || path.size() >= sizeof addr.sun_path) return -1;
    strcpy(addr.sun_path, path.c_str());
    if (connect(fd, (sockaddr *) &addr, sizeof addr) != 0) {
        close(fd);
        return -1;
    }
    return fd;
}

// one request through child `d`: the reply (exit code, stdout, stderr) as raw bytes
static bool ask_child(const Router &r, int d, const std::vector<std::string> &strs, int32_t &code, std::string &so, std::string &se)
{
    const int fd = connect_to(r.child_sock[(size_t) d]);
    if (fd < 0) return false;
    uint64_t lo = 0, le = 0;
    bool ok = send_request(fd, strs) && read_all(fd, &code, 4) && read_all(fd, &lo, 8) && lo < (1ull << 32);
    if (ok) so.resize(lo), ok = lo == 0 || read_all(fd, &so[0], lo);
    ok = ok && read_all(fd, &le, 8) && le < (1ull << 32);
    if (ok) se.resize(le), ok = le == 0 || read_all(fd, &se[0], le);
    close(fd);
    return ok;
}

static int count_render_nodes()
{
    int n = 0;
    if (DIR *d = opendir("/dev/dri")) {
        while (dirent *e = readdir(d))
            if (!strncmp(e->d_name, "renderD", 7)) ++n;
        closedir(d);
    }
    return n;
}

static int serve_multi(const char *sock_path, int n_dev)
{
    signal(SIGPIPE, SIG_IGN);
    Router r;
    r.sock = sock_path;
    std::vector<int> dev_of((size_t) n_dev);
    for (int d = 0; d < n_dev; ++d) dev_of[(size_t) d] = d;
    if (const char *lst = getenv("COMMET_SERVER_DEVICE_LIST")) {      // rehearsals: which device every child takes
        std::stringstream ss(lst);
        std::string item;
        for (int d = 0; d < n_dev && std::getline(ss, item, ','); ++d) dev_of[(size_t) d] = atoi(item.c_str());
    }
    // the children first: nothing in this process has touched the GPU yet (no HIP call is ever made by the router)
    for (int d = 0; d < n_dev; ++d) {
        const std::string cs = std::string(sock_path) + "." + std::to_string(d);
        const pid_t pid = fork();
        if (pid < 0) {
            std::cerr << "Error: cannot start the server of device " << d << "\n";
            for (pid_t p : r.child_pid) kill(p, SIGTERM);
            return 1;
        }
        if (pid == 0) {
            setenv("COMMET_DEVICE", std::to_string(dev_of[(size_t) d]).c_str(), 1);
            _exit(serve(cs.c_str()));
        }
        r.child_sock.push_back(cs), r.child_pid.push_back(pid);
    }
    r.inflight.assign((size_t) n_dev, 0), r.files_on.assign((size_t) n_dev, 0), r.served.assign((size_t) n_dev, 0);
    const int ls = socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un addr;
    memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX;
    bool up = ls >= 0 && strlen(sock_path) < sizeof addr.sun_path;
    if (up) {
        strcpy(addr.sun_path, sock_path);
        unlink(sock_path);
        // (the clients' socket appears once every child listens on its own: a client that finds it may send at once)
        for (int d = 0; d < n_dev && up; ++d) {
            struct stat sb;
            int tries = 0;
            while (stat(r.child_sock[(size_t) d].c_str(), &sb) != 0 && tries++ < 1200) {
                int st = 0;
                if (waitpid(r.child_pid[(size_t) d], &st, WNOHANG) > 0) { up = false; break; }
                usleep(50000);
            }
            up = up && stat(r.child_sock[(size_t) d].c_str(), &sb) == 0;
        }
        up = up && bind(ls, (sockaddr *) &addr, sizeof addr) == 0 && listen(ls, 64) == 0;
    }
    if (!up) {
        std::cerr << "Error: cannot serve on " << sock_path << " with " << n_dev << " device(s)\n";
        for (pid_t p : r.child_pid) kill(p, SIGTERM);
        return 1;
    }
    std::cerr << "index_and_search: serving on " << sock_path << " with " << n_dev << " device servers\n";
    std::atomic<int> active{0};                                     // client threads under way (detached: a server lives for days)
    bool stop = false;
    while (!stop) {
        const int fd = accept(ls, nullptr, nullptr);
        if (fd < 0) {
            if (errno != EINTR) usleep(20000);
            continue;
        }
        uint32_t n = 0;
        std::vector<std::string> strs;
        bool ok = read_all(fd, &n, 4) && n >= 2 && n < 4096;
        for (uint32_t i = 0; ok && i < n; ++i) {
            uint32_t len = 0;
            ok = read_all(fd, &len, 4) && len < (1u << 20);
            if (!ok) break;
            std::string x(len, '\0');
            ok = len == 0 || read_all(fd, &x[0], len);
            strs.push_back(x);
        }
        if (!ok) {
            close(fd);
            continue;
        }
        if (strs.size() == 3 && strs[2] == "--server-stop") {
            close(fd);
            stop = true;
            break;
        }
        if (strs.size() == 3 && strs[2] == "--server-stats") {      // the children's lines, then the router's own
            std::string so, se;
            for (int d = 0; d < n_dev; ++d) {
                int32_t c = 0;
                std::string o, e;
                if (ask_child(r, d, strs, c, o, e)) so += "device server " + std::to_string(d) + ": " + o;
            }
            {
                std::lock_guard<std::mutex> lk(r.mu);
                std::ostringstream os;
                os << "router: requests " << r.requests << ", routed to a device that held files of the job " << r.routed_by_residency << ", per device";
                for (int d = 0; d < n_dev; ++d) os << " " << r.served[(size_t) d];
                os << "\n";
                so += os.str();
            }
            const int32_t c32 = 0;
            const uint64_t lo = so.size(), le = 0;
            (void) (write_all(fd, &c32, 4) && write_all(fd, &lo, 8) && write_all(fd, so.data(), lo) && write_all(fd, &le, 8));
            close(fd);
            continue;
        }
        ++active;
        std::thread([&r, &active, fd, strs]() {
            const int d = r.pick(Router::job_files(strs));
            int32_t code = 1;
            std::string so, se;
            if (!ask_child(r, d, strs, code, so, se)) {
                code = 1;
                se = "Error: the index_and_search server of device " + std::to_string(d) + " did not answer\n";
                so.clear();
            }
            {
                std::lock_guard<std::mutex> lk(r.mu);
                --r.inflight[(size_t) d];
            }
            const uint64_t lo = so.size(), le = se.size();
            (void) (write_all(fd, &code, 4) && write_all(fd, &lo, 8) && write_all(fd, so.data(), lo) && write_all(fd, &le, 8) &&
                    write_all(fd, se.data(), le));
            close(fd);
            --active;
        }).detach();
    }
    while (active.load() > 0) usleep(10000);                        // (requests under way finish first)
    for (int d = 0; d < n_dev; ++d) {
        const int fd = connect_to(r.child_sock[(size_t) d]);
        if (fd >= 0) {
            (void) send_request(fd, {"/", "index_and_search", "--server-stop"});
            close(fd);
        }
    }
    for (pid_t p : r.child_pid) {
        int st = 0;
        (void) waitpid(p, &st, 0);
    }
    close(ls);
    unlink(sock_path);
    return 0;
}

// forwards this invocation to a resident server; -1 = no server there (run locally)
static int forward(const char *sock_path, int argc, char **argv)
{
    const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un addr;
    memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX;
    if (fd < 0 || strlen(sock_path) >= sizeof addr.sun_path) return -1;
    strcpy(addr.sun_path, sock_path);
    if (connect(fd, (sockaddr *) &addr, sizeof addr) != 0) {
        close(fd);
        return -1;
    }
    char cwd[PATH_MAX];
    if (!getcwd(cwd, sizeof cwd)) strcpy(cwd, ".");
    std::vector<std::string> strs(1, cwd);
    for (int i = 0; i < argc; ++i) strs.push_back(argv[i]);
    const uint32_t n = (uint32_t) strs.size();
    bool ok = write_all(fd, &n, 4);
    for (const std::string &x : strs) {
        const uint32_t len = (uint32_t) x.size();
        ok = ok && write_all(fd, &len, 4) && write_all(fd, x.data(), len);
    }
    int32_t code = 1;
    uint64_t lo = 0, le = 0;
    std::string so, se;
    ok = ok && read_all(fd, &code, 4) && read_all(fd, &lo, 8);
    if (ok) so.resize(lo), ok = lo == 0 || read_all(fd, &so[0], lo);
    ok = ok && read_all(fd, &le, 8);
    if (ok) se.resize(le), ok = le == 0 || read_all(fd, &se[0], le);
    close(fd);
    if (!ok) {
        if (argc >= 2 && !strcmp(argv[1], "--server-stop")) return 0;
        std::cerr << "Error: the index_and_search server at " << sock_path << " closed the connection\n";
        return 1;
    }
    fwrite(so.data(), 1, so.size(), stdout);
    fwrite(se.data(), 1, se.size(), stderr);
    return code;
}

int main(int argc, char **argv)
{
    if (argc >= 3 && !strcmp(argv[1], "--serve")) {
        const char *nd = (argc == 5 && !strcmp(argv[3], "--devices")) ? argv[4] : (argc == 3 ? getenv("COMMET_SERVER_DEVICES") : nullptr);
        if (argc != 3 && !nd) {
            std::cerr << "Usage : ./index_and_search --serve <socket> [--devices <n>|all]\n";
            return 1;
        }
        const int n_dev = !nd ? 1 : !strcmp(nd, "all") ? count_render_nodes() : atoi(nd);
        if (nd && n_dev < 1) {
            std::cerr << "Error: --devices " << nd << ": no device\n";
            return 1;
        }
        return nd ? serve_multi(argv[2], n_dev) : serve(argv[2]);
    }
    if (const char *srv = getenv("COMMET_SERVER")) {
        const int rc = forward(srv, argc, argv);
        if (rc >= 0) return rc;
        if (argc == 2 && (!strcmp(argv[1], "--server-stats") || !strcmp(argv[1], "--server-stop"))) {
            std::cerr << "Error: no index_and_search server at " << srv << "\n";
            return 1;
        }
    }
    try {
        return run_tool(argc, argv);
    } catch (const ToolExit &e) {
        return e.code;
    }
}
