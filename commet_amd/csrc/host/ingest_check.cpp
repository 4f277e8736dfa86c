// ingest_check.cpp — CPU-only driver of host/ingest_pack.hpp for the tests (tests/test_ingest_pack.py and the
// sanitizer builds of tests/test_sanitizers.py): the threaded parser + packer that libcommet_hip.so runs in front of
// hipMemcpyAsync, here with a sink that "uploads" into host memory.
//
//   ingest_check [--arrays] [--stage-triples N] [--stage-reads N] [--threads T] out.bin file...
//
// out.bin: u64 n_reads, n_bases, triples, min_len, max_len, n_empty, n_files; u64 file_reads[n_files]; u64 empty[n_empty];
//          u64 goff[n_reads]; u32 planes[3 * triples]
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "ingest_pack.hpp"

using namespace commet_host;

namespace {

struct MemSink {
    std::vector<uint32_t> planes;
    std::vector<uint64_t> goff;
    uint64_t cap_triples = 1 << 16, cap_reads = 1 << 10;
    struct Buf {
        std::vector<uint32_t> p;
        std::vector<uint64_t> g;
    };
    std::vector<Buf> bufs;   // one per worker
    void prepare(uint64_t reads, uint64_t bases, int workers)
    {
        planes.assign(3 * ((bases >> 5) + reads + 1), 0);
        goff.assign(reads + 1, 0);
        bufs.resize(workers);
        for (Buf &b : bufs) b.p.resize(3 * cap_triples), b.g.resize(cap_reads);
    }
    bool acquire(int worker, PackStage &st)
    {
        st.planes = bufs[worker].p.data(), st.goff = bufs[worker].g.data();
        st.cap_triples = cap_triples, st.cap_reads = cap_reads;
        return true;
    }
    bool flush(int, const PackStage &st, uint64_t triple0, uint64_t n_triples, uint64_t read0, uint64_t n_reads)
    {
        if (3 * (triple0 + n_triples) > planes.size() || read0 + n_reads >= goff.size() + (n_reads ? 0 : 1)) return false;
        memcpy(planes.data() + 3 * triple0, st.planes, n_triples * 12);   // disjoint ranges: no lock
        memcpy(goff.data() + read0, st.goff, n_reads * 8);
        return true;
    }
};

}  // namespace

int main(int argc, char **argv)
{
    bool arrays = false;
    int T = 4;
    MemSink sink;
    int a = 1;
    for (; a < argc && argv[a][0] == '-'; ++a) {
        const std::string o = argv[a];
        if (o == "--arrays") arrays = true;
        else if (o == "--stage-triples" && a + 1 < argc) sink.cap_triples = strtoull(argv[++a], nullptr, 10);
        else if (o == "--stage-reads" && a + 1 < argc) sink.cap_reads = strtoull(argv[++a], nullptr, 10);
        else if (o == "--threads" && a + 1 < argc) T = atoi(argv[++a]);
        else {
            fprintf(stderr, "unknown option %s\n", argv[a]);
            return 2;
        }
    }
    if (argc - a < 2) {
        fprintf(stderr, "usage: ingest_check [--arrays] [--stage-triples N] [--stage-reads N] [--threads T] out.bin file...\n");
        return 2;
    }
    const char *out = argv[a++];
    std::vector<std::unique_ptr<ReadFileData>> files;
    std::vector<const char *> data;
    std::vector<size_t> sizes;
    std::vector<ReadFormat> fmts;
    for (; a < argc; ++a) {
        std::unique_ptr<ReadFileData> f(new ReadFileData);
        if (!f->open_file(argv[a]) || f->format() == ReadFormat::Unknown) {
            fprintf(stderr, "cannot read %s\n", argv[a]);
            return 1;
        }
        data.push_back(f->data()), sizes.push_back(f->size()), fmts.push_back(f->format());
        files.push_back(std::move(f));
    }
    std::vector<uint64_t> file_reads;
    uint64_t n_reads = 0, n_bases = 0;
    PackSummary sm;
    std::string err;
    bool ok;
    if (!arrays) {
        ok = ingest_files<MemSink>(data, sizes, fmts, T,
                                   [&](uint64_t r, uint64_t b, int w) -> MemSink * {
                                       sink.prepare(r, b, w);
                                       return &sink;
                                   },
                                   file_reads, n_reads, n_bases, sm, err);
    } else {
        // the (bases, offsets) entry point: records are first laid out the way commet_readset_append gets them
        std::vector<uint8_t> bases;
        std::vector<uint64_t> offs(1, 0);
        for (size_t f = 0; f < data.size(); ++f) {
            std::vector<IngestPiece> ps;
            split_file((int) f, fmts[f], data[f], sizes[f], 1, ps);
            uint64_t cnt = 0;
            for (IngestPiece &p : ps) {
                count_piece(p);
                // plain re-parse: sequence bytes of every record, concatenated
                if (p.fmt == ReadFormat::Fastq) {
                    for_each_fastq_record(p.d, p.n, p.n_reads, [&](const char *s, size_t len) {
                        bases.insert(bases.end(), s, s + len);
                        offs.push_back(bases.size());
                    });
                } else {
                    size_t i = 0;
                    while (i < p.n && p.d[i] != '>') {
                        const char *nl = (const char *) memchr(p.d + i, '\n', p.n - i);
                        i = nl ? (size_t) (nl - p.d) + 1 : p.n;
                    }
                    while (i < p.n) {
                        const char *nl = (const char *) memchr(p.d + i, '\n', p.n - i);
                        size_t j = nl ? (size_t) (nl - p.d) + 1 : p.n;
                        while (j < p.n && p.d[j] != '>') {
                            nl = (const char *) memchr(p.d + j, '\n', p.n - j);
                            const size_t e = nl ? (size_t) (nl - p.d) : p.n;
                            bases.insert(bases.end(), p.d + j, p.d + e);
                            j = nl ? e + 1 : p.n;
                        }
                        offs.push_back(bases.size());
                        i = j;
                    }
                }
                cnt += p.n_reads;
            }
            file_reads.push_back(cnt);
        }
        n_reads = offs.size() - 1, n_bases = bases.size();
        sink.prepare(n_reads, n_bases, T);
        ok = ingest_arrays(bases.data(), offs.data(), n_reads, 0, 0, T, sink, sm, err);
    }
    if (!ok) {
        fprintf(stderr, "ingest failed: %s\n", err.c_str());
        return 1;
    }
    FILE *fh = fopen(out, "wb");
    if (!fh) return 1;
    const uint64_t triples = (n_bases >> 5) + n_reads + 1;
    const uint64_t head[7] = {n_reads, n_bases, triples, n_reads ? sm.min_len : 0, sm.max_len, sm.empty_reads.size(), file_reads.size()};
    fwrite(head, 8, 7, fh);
    fwrite(file_reads.data(), 8, file_reads.size(), fh);
    if (!sm.empty_reads.empty()) fwrite(sm.empty_reads.data(), 8, sm.empty_reads.size(), fh);
    if (n_reads) fwrite(sink.goff.data(), 8, n_reads, fh);
    fwrite(sink.planes.data(), 4, 3 * triples, fh);
    fclose(fh);
    return 0;
}
