// Host-side loader of x-vectors from Kaldi binary archives, the step in front of the embedding
// prologue.  The reference reads one record at a time through Python file objects under a lock
// (hyperion/io/ark_data_reader.py:554-622, hyperion/utils/kaldi_matrix.py:49-117) and stacks the
// arrays afterwards; here the records named by an index (archive, byte offset: the columns of an scp
// file, hyperion/utils/scp_list.py:160-201) land directly in the rows of one caller buffer, typically
// page-locked memory that is then copied to the device in one transfer.  Threads use pread (no shared
// file position, no locks); records that follow each other in an archive are fetched with one read.
//
// Record layout at its offset (kaldi_matrix.py:163-178, kaldi_io_funcs.py:11-15, 58-65, 101-107):
//   "\0B"  token "FV " | "DV " | "FM " | "DM "  then for matrices '\4' int32 rows,  '\4' int32 cols,
//   then rows x cols little-endian float32 (F) or float64 (D).
// Compressed matrices (tokens CM, CM2, CM3) and text archives are not x-vector formats and are refused.
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace {

constexpr int64_t ARK_MAX_HEADER = 15;            // \0B + "FM " + 2 x (\4 + int32)
constexpr int64_t ARK_SPAN_BYTES = 8ll << 20;     // records closer than this are read together

struct ArkShape {
    int32_t rows, cols, elem_bytes, is_matrix;
    int64_t header_bytes;
};

// Parses a record header from `p` (at least `avail` bytes); returns an error text or nullptr.
const char* parse_header(const unsigned char* p, int64_t avail, ArkShape* s) {
    if (avail < 10) return "truncated record";
    if (p[0] != 0 || p[1] != 'B') return "not a binary record (text archives are not supported)";
    if (p[2] == 'C') return "compressed matrix records are not supported";
    if ((p[2] != 'F' && p[2] != 'D') || (p[3] != 'V' && p[3] != 'M') || p[4] != ' ')
        return "unknown record token";
    s->elem_bytes = p[2] == 'F' ? 4 : 8;
    s->is_matrix = p[3] == 'M';
    int64_t at = 5;
    int32_t dims[2] = {1, 0};
    for (int i = s->is_matrix ? 0 : 1; i < 2; ++i) {
        if (at + 5 > avail) return "truncated record";
        if (p[at] != 4) return "bad integer marker in record header";
        memcpy(&dims[i], p + at + 1, 4);
        at += 5;
    }
    if (dims[0] < 0 || dims[1] < 0) return "negative dimension in record header";
    s->rows = dims[0];
    s->cols = dims[1];
    s->header_bytes = at;
    return nullptr;
}

struct Failure {
    std::atomic<int64_t> index{-1};
    std::string text;  // written only by the thread that won `index`
    void report(int64_t i, const std::string& t) {
        int64_t none = -1;
        if (index.compare_exchange_strong(none, i)) text = t;
    }
};

struct ReadJob {
    const char* const* archives;
    const int32_t* archive_of;
    const int64_t* offsets;
    const int64_t* rows;  // output row of record i (nullptr: row i)
    int32_t dim, out_f64;
    void* out;
    int64_t ld;
};

void convert_row(const unsigned char* src, int elem_bytes, int dim, int out_f64, void* dst) {
    if (elem_bytes == 4 && !out_f64) memcpy(dst, src, (size_t)dim * 4);
    else if (elem_bytes == 8 && out_f64) memcpy(dst, src, (size_t)dim * 8);
    else if (elem_bytes == 4) {
        double* d = static_cast<double*>(dst);
        for (int j = 0; j < dim; ++j) { float v; memcpy(&v, src + 4 * j, 4); d[j] = (double)v; }
    } else {
        float* d = static_cast<float*>(dst);
        for (int j = 0; j < dim; ++j) { double v; memcpy(&v, src + 8 * (size_t)j, 8); d[j] = (float)v; }
    }
}

void read_range(const ReadJob& job, int64_t i0, int64_t i1, Failure* fail) {
    int fd = -1, fd_archive = -1;  // one open archive at a time (callers pass records grouped by archive)
    std::vector<unsigned char> buf;
    const int64_t rec_max = ARK_MAX_HEADER + (int64_t)job.dim * 8;
    auto out_row = [&](int64_t i) {
        const int64_t row = job.rows ? job.rows[i] : i;
        return static_cast<unsigned char*>(job.out) + (size_t)row * (size_t)job.ld * (job.out_f64 ? 8 : 4);
    };
    int64_t i = i0;
    while (i < i1 && fail->index.load(std::memory_order_relaxed) < 0) {
        const int32_t a = job.archive_of[i];
        // a run of records in ascending position inside one archive, within one span
        int64_t j = i + 1;
        while (j < i1 && job.archive_of[j] == a && job.offsets[j] > job.offsets[j - 1] &&
               job.offsets[j] + rec_max - job.offsets[i] <= ARK_SPAN_BYTES)
            ++j;
        if (fd_archive != a) {
            if (fd >= 0) close(fd);
            fd = open(job.archives[a], O_RDONLY | O_CLOEXEC);
            fd_archive = a;
            if (fd < 0) {
                fail->report(i, std::string("cannot open ") + job.archives[a] + ": " + strerror(errno));
                break;
            }
        }
        const int64_t base = job.offsets[i];
        const int64_t want = job.offsets[j - 1] + rec_max - base;
        buf.resize((size_t)want);
        int64_t got = 0;
        while (got < want) {
            const ssize_t r = pread(fd, buf.data() + got, (size_t)(want - got), (off_t)(base + got));
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) break;  // error or end of file: whatever arrived is checked record by record
            got += r;
        }
        for (int64_t r = i; r < j; ++r) {
            const int64_t at = job.offsets[r] - base;
            ArkShape s;
            const char* err = at < 0 || at >= got ? "offset beyond the end of the archive"
                                                  : parse_header(buf.data() + at, got - at, &s);
            if (!err && s.rows != 1) err = "record is a matrix with more than one row";
            if (!err && s.cols != job.dim) err = "record has a different dimension";
            if (!err && at + s.header_bytes + (int64_t)s.cols * s.elem_bytes > got) err = "truncated record";
            if (err) {
                fail->report(r, std::string(err) + " (" + job.archives[a] + ":" +
                                    std::to_string((long long)job.offsets[r]) + ")");
                break;
            }
            convert_row(buf.data() + at + s.header_bytes, s.elem_bytes, job.dim, job.out_f64, out_row(r));
        }
        i = j;
    }
    if (fd >= 0) close(fd);
}

}  // namespace

extern "C" int hypmi_ark_probe(const char* archive, int64_t offset, int32_t* rows, int32_t* cols,
                               int32_t* elem_bytes, int32_t* is_matrix) {
    HYPMI_REQUIRE(archive && rows && cols && elem_bytes && is_matrix && offset >= 0, "ark_probe: bad argument");
    const int fd = open(archive, O_RDONLY | O_CLOEXEC);
    HYPMI_REQUIRE(fd >= 0, "ark_probe: cannot open %s: %s", archive, strerror(errno));
    unsigned char head[ARK_MAX_HEADER];
    const ssize_t got = pread(fd, head, sizeof(head), (off_t)offset);
    close(fd);
    ArkShape s;
    const char* err = got < 0 ? strerror(errno) : parse_header(head, got, &s);
    HYPMI_REQUIRE(!err, "ark_probe: %s (%s:%lld)", err, archive, (long long)offset);
    *rows = s.rows;
    *cols = s.cols;
    *elem_bytes = s.elem_bytes;
    *is_matrix = s.is_matrix;
    return HYPMI_OK;
}

extern "C" int hypmi_ark_read_vectors(const char* const* archives, int32_t n_archives,
                                      const int32_t* archive_of, const int64_t* offsets,
                                      const int64_t* rows, int64_t n, int32_t dim, int32_t out_f64,
                                      void* out, int64_t ld, int32_t n_threads) {
    HYPMI_REQUIRE(n >= 0 && dim > 0 && ld >= dim && n_archives >= 0, "ark_read_vectors: bad shape");
    if (n == 0) return HYPMI_OK;
    HYPMI_REQUIRE(archives && archive_of && offsets && out, "ark_read_vectors: null pointer");
    for (int64_t i = 0; i < n; ++i)
        HYPMI_REQUIRE(archive_of[i] >= 0 && archive_of[i] < n_archives && offsets[i] >= 0 &&
                          (!rows || (rows[i] >= 0 && rows[i] < n)),
                      "ark_read_vectors: record %lld has no valid archive, offset or row", (long long)i);
    int64_t threads = n_threads > 0 ? n_threads : 1;
    threads = std::min<int64_t>(threads, std::min<int64_t>(64, (n + 255) / 256));
    const ReadJob job = {archives, archive_of, offsets, rows, dim, out_f64, out, ld};
    Failure fail;
    if (threads <= 1) {
        read_range(job, 0, n, &fail);
    } else {
        std::vector<std::thread> pool;
        const int64_t per = (n + threads - 1) / threads;
        for (int64_t t = 0; t < threads; ++t) {
            const int64_t i0 = t * per, i1 = std::min(n, i0 + per);
            if (i0 < i1) pool.emplace_back(read_range, std::cref(job), i0, i1, &fail);
        }
        for (auto& th : pool) th.join();
    }
    if (fail.index.load() >= 0) {
        hypmi_set_error("ark_read_vectors: record %lld: %s", (long long)fail.index.load(), fail.text.c_str());
        return HYPMI_EIO;
    }
    return HYPMI_OK;
}
