// capi/images.hpp — a resident set outside its process: packed images (commet_readset_save / _load) and the device-to-device hand-over between the processes of a node (commet_readset_export / _import, HIP IPC)
// (a part of the one translation unit capi.hip: included there, in order, after the kernels and state.hpp)
#pragma once

extern "C" {

/* ---- packed images of a read set (k-independent): parse once, load everywhere ---------------------------------- */
namespace {
struct PackHeader {
    char     magic[8];          // "CMTPK01"
    uint64_t n_reads, n_bases, triples, n_files, n_empty;
    uint32_t uniform_len, min_len, max_len, pad;
};
inline uint64_t align64(uint64_t x) { return (x + 63) & ~63ull; }
struct PackLayout {
    uint64_t files_at, empty_at, planes_at, goff_at, total;
    PackLayout(const PackHeader &h)
    {
        files_at = align64(sizeof(PackHeader));
        empty_at = files_at + h.n_files * sizeof(FileSpan);
        planes_at = align64(empty_at + h.n_empty * 8);
        goff_at = align64(planes_at + h.triples * 12);
        total = goff_at + (h.uniform_len ? 0 : (h.n_reads + 1) * 8);
    }
};
}  // namespace

int commet_readset_save(const commet_readset *rs, const char *path)
{
    if (!rs->finalized) return fail("read set not finalized");
    commet_ctx *c = rs->ctx;
    HIP_OK(hipSetDevice(c->device));
    PackHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "CMTPK01", 8);
    h.n_reads = rs->n_reads, h.n_bases = rs->n_bases, h.triples = (rs->n_bases >> 5) + rs->n_reads + 1;
    h.n_files = rs->files.size(), h.n_empty = rs->empty_reads.size();
    h.uniform_len = rs->uniform_len, h.min_len = rs->min_len, h.max_len = rs->max_len;
    const PackLayout lay(h);
    const std::string tmp = std::string(path) + ".tmp";
    (void) unlink(tmp.c_str());                                   // (what an interrupted save may have left)
    const int fd = open(tmp.c_str(), O_RDWR | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);   // never through a link somebody else planted
    if (fd < 0) return fail("cannot create %s: %s", tmp.c_str(), strerror(errno));
    if (ftruncate(fd, (off_t) lay.total) != 0) {
        close(fd);
        return fail("cannot size %s to %llu bytes: %s", tmp.c_str(), (unsigned long long) lay.total, strerror(errno));
    }
    uint8_t *m = (uint8_t *) mmap(nullptr, lay.total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return fail("cannot map %s: %s", tmp.c_str(), strerror(errno));
    memcpy(m, &h, sizeof h);
    if (h.n_files) memcpy(m + lay.files_at, rs->files.data(), h.n_files * sizeof(FileSpan));
    if (h.n_empty) memcpy(m + lay.empty_at, rs->empty_reads.data(), h.n_empty * 8);
    hipError_t e = hipStreamSynchronize(c->load_stream);
    if (e == hipSuccess) e = hipMemcpy(m + lay.planes_at, rs->d_planes, h.triples * 12, hipMemcpyDeviceToHost);
    if (e == hipSuccess && !h.uniform_len) e = hipMemcpy(m + lay.goff_at, rs->d_goff, (h.n_reads + 1) * 8, hipMemcpyDeviceToHost);
    munmap(m, lay.total);
    if (e != hipSuccess) {
        unlink(tmp.c_str());
        return fail("read set download failed: %s", hipGetErrorString(e));
    }
    if (rename(tmp.c_str(), path) != 0) return fail("cannot rename %s: %s", tmp.c_str(), strerror(errno));
    return 0;
}

commet_readset *commet_readset_load(commet_ctx *c, const char *path)
{
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        fail("cannot open %s: %s", path, strerror(errno));
        return nullptr;
    }
    struct stat st;
    PackHeader h;
    if (fstat(fd, &st) != 0 || (uint64_t) st.st_size < sizeof h || pread(fd, &h, sizeof h, 0) != (ssize_t) sizeof h ||
        memcmp(h.magic, "CMTPK01", 8) != 0) {
        close(fd);
        fail("%s is not a packed read set", path);
        return nullptr;
    }
    // the counts are bounded by the file's own size before any arithmetic is done with them
    const uint64_t fsz = (uint64_t) st.st_size;
    if (h.n_files > fsz / sizeof(FileSpan) || h.n_empty > fsz / 8 || h.triples > fsz / 12 || h.n_reads > h.triples || (h.n_bases >> 5) > h.triples) {
        close(fd);
        fail("%s: inconsistent packed read set", path);
        return nullptr;
    }
    const PackLayout lay(h);
    if (h.triples != (h.n_bases >> 5) + h.n_reads + 1 || lay.total != (uint64_t) st.st_size) {
        close(fd);
        fail("%s: inconsistent packed read set", path);
        return nullptr;
    }
    const uint8_t *m = (const uint8_t *) mmap(nullptr, lay.total, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        fail("cannot map %s: %s", path, strerror(errno));
        return nullptr;
    }
    commet_readset *rs = commet_readset_create(c, h.n_reads, h.n_bases);
    if (!rs) {
        munmap((void *) m, lay.total);
        return nullptr;
    }
    const FileSpan *fs = (const FileSpan *) (m + lay.files_at);
    rs->files.assign(fs, fs + h.n_files);
    const uint64_t *er = (const uint64_t *) (m + lay.empty_at);
    rs->empty_reads.assign(er, er + h.n_empty);
    rs->n_reads = h.n_reads;
    rs->n_bases = h.n_bases;
    const uint32_t mm[3] = {h.n_reads ? h.min_len : 0xFFFFFFFFu, h.max_len, 0u};
    hipError_t e = hipMemcpyAsync(rs->d_lenmm, mm, sizeof mm, hipMemcpyHostToDevice, c->load_stream);
    // the image is pageable memory: the copies below are staged by the runtime and return when the source has been read
    if (e == hipSuccess) e = hipMemcpyAsync(rs->d_planes, m + lay.planes_at, h.triples * 12, hipMemcpyHostToDevice, c->load_stream);
    if (e == hipSuccess && !h.uniform_len)
        e = hipMemcpyAsync(rs->d_goff, m + lay.goff_at, (h.n_reads + 1) * 8, hipMemcpyHostToDevice, c->load_stream);
    if (e == hipSuccess && h.n_reads) {
        ReadsView v = rs->view();
        v.uniform_len = h.uniform_len;
        COMMET_LAUNCH(kmer_counts_kernel, dim3((unsigned) ((h.n_reads + 255) / 256)), dim3(256), 0, c->load_stream, v, c->k, rs->d_kcnt,
                           rs->d_lenmm);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->load_stream);
    munmap((void *) m, lay.total);
    if (e != hipSuccess) {
        fail("read set upload failed: %s", hipGetErrorString(e));
        commet_readset_destroy(rs);
        return nullptr;
    }
    return rs;
}

/* ---- a resident set handed to another process of the node without a file ---------------------------------------- */
namespace {
struct ExportTail {                 // behind PackHeader + file spans + empty reads, 8-byte aligned
    hipIpcMemHandle_t planes, goff;
    int32_t device, has_goff;
};
}  // namespace

int commet_readset_export(const commet_readset *rs, void *blob, uint64_t cap, uint64_t *blob_bytes)
{
    if (!rs->finalized) return fail("read set not finalized");
    commet_ctx *c = rs->ctx;
    HIP_OK(hipSetDevice(c->device));
    PackHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "CMTIPC1", 8);
    h.n_reads = rs->n_reads, h.n_bases = rs->n_bases, h.triples = (rs->n_bases >> 5) + rs->n_reads + 1;
    h.n_files = rs->files.size(), h.n_empty = rs->empty_reads.size();
    h.uniform_len = rs->uniform_len, h.min_len = rs->min_len, h.max_len = rs->max_len;
    const uint64_t files_at = align64(sizeof h), empty_at = files_at + h.n_files * sizeof(FileSpan);
    const uint64_t tail_at = align64(empty_at + h.n_empty * 8), total = tail_at + sizeof(ExportTail);
    if (blob_bytes) *blob_bytes = total;
    if (!blob || cap < total) return blob ? fail("export buffer too small (%llu bytes needed)", (unsigned long long) total) : 0;   // (size query)
    uint8_t *m = (uint8_t *) blob;
    memset(m, 0, total);
    memcpy(m, &h, sizeof h);
    if (h.n_files) memcpy(m + files_at, rs->files.data(), h.n_files * sizeof(FileSpan));
    if (h.n_empty) memcpy(m + empty_at, rs->empty_reads.data(), h.n_empty * 8);
    ExportTail t;
    memset(&t, 0, sizeof t);
    t.device = c->device, t.has_goff = h.uniform_len ? 0 : 1;
    HIP_OK(hipStreamSynchronize(c->load_stream));        // the planes are complete
    {   // buffers from the stream-ordered pool have no IPC handle: moved into hipMalloc blocks once, here (state.hpp, dm_make_shareable)
        commet_readset *w = const_cast<commet_readset *>(rs);
        bool busy;
        {
            std::lock_guard<std::mutex> lk(c->ql_mu);           // (in_job is written under this mutex)
            busy = rs->in_job;
        }
        if (busy) return fail("read set is part of a running job: export it before or after");
        HIP_OK(dm_make_shareable((void **) &w->d_planes));
        if (t.has_goff) HIP_OK(dm_make_shareable((void **) &w->d_goff));
    }
    HIP_OK(hipIpcGetMemHandle(&t.planes, rs->d_planes));
    if (t.has_goff) HIP_OK(hipIpcGetMemHandle(&t.goff, rs->d_goff));
    memcpy(m + tail_at, &t, sizeof t);
    return 0;
}

commet_readset *commet_readset_import(commet_ctx *c, const void *blob, uint64_t blob_bytes)
{
    PackHeader h;
    if (!blob || blob_bytes < sizeof h) {
        fail("not an exported read set");
        return nullptr;
    }
    memcpy(&h, blob, sizeof h);
    const uint64_t files_at = align64(sizeof h);
    if (memcmp(h.magic, "CMTIPC1", 8) != 0 || h.n_files > blob_bytes / sizeof(FileSpan) || h.n_empty > blob_bytes / 8 ||
        h.triples != (h.n_bases >> 5) + h.n_reads + 1) {
        fail("not an exported read set");
        return nullptr;
    }
    const uint64_t empty_at = files_at + h.n_files * sizeof(FileSpan), tail_at = align64(empty_at + h.n_empty * 8);
    if (tail_at + sizeof(ExportTail) != blob_bytes) {
        fail("inconsistent exported read set");
        return nullptr;
    }
    const uint8_t *m = (const uint8_t *) blob;
    ExportTail t;
    memcpy(&t, m + tail_at, sizeof t);
    commet_readset *rs = commet_readset_create(c, h.n_reads, h.n_bases);
    if (!rs) return nullptr;
    const FileSpan *fs = (const FileSpan *) (m + files_at);
    rs->files.assign(fs, fs + h.n_files);
    const uint64_t *er = (const uint64_t *) (m + empty_at);
    rs->empty_reads.assign(er, er + h.n_empty);
    rs->n_reads = h.n_reads;
    rs->n_bases = h.n_bases;
    // the owner's buffers, mapped into this process (another device of the node: over xGMI), copied device to device
    void *src_planes = nullptr, *src_goff = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&src_planes, t.planes, hipIpcMemLazyEnablePeerAccess);
    if (e == hipSuccess && t.has_goff) e = hipIpcOpenMemHandle(&src_goff, t.goff, hipIpcMemLazyEnablePeerAccess);
    const uint32_t mm[3] = {h.n_reads ? h.min_len : 0xFFFFFFFFu, h.max_len, 0u};
    if (e == hipSuccess) e = hipMemcpyAsync(rs->d_lenmm, mm, sizeof mm, hipMemcpyHostToDevice, c->load_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(rs->d_planes, src_planes, h.triples * 12, hipMemcpyDeviceToDevice, c->load_stream);
    if (e == hipSuccess && t.has_goff) e = hipMemcpyAsync(rs->d_goff, src_goff, (h.n_reads + 1) * 8, hipMemcpyDeviceToDevice, c->load_stream);
    if (e == hipSuccess && h.n_reads) {
        ReadsView v = rs->view();
        v.uniform_len = h.uniform_len;
        COMMET_LAUNCH(kmer_counts_kernel, dim3((unsigned) ((h.n_reads + 255) / 256)), dim3(256), 0, c->load_stream, v, c->k, rs->d_kcnt,
                      rs->d_lenmm);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->load_stream);
    if (src_planes) (void) hipIpcCloseMemHandle(src_planes);
    if (src_goff) (void) hipIpcCloseMemHandle(src_goff);
    if (e != hipSuccess) {
        fail("read set import failed (device %d -> %d): %s", t.device, c->device, hipGetErrorString(e));
        (void) hipGetLastError();
        commet_readset_destroy(rs);
        return nullptr;
    }
    return rs;
}

}  // extern "C"
