// host_io.cpp — libemcid_host.so: the v* cache reads of an edit (include/emcid_host.h, emcid_read_npz_rows_f32).  Plain C++17, no
// GPU.  The reference reads one `np.savez(f, v_star=...)` file per request with np.load (emcid/emcid_main.py:885-899: 1 000
// zip opens + header evals per mass edit, ~100 ms of Python); here the same files are read by a few threads straight into the
// caller's (pinned) row buffer: open, read, zip local header, npy header, convert.  Anything that is not exactly such a file
// (compressed member, other member order, Fortran order, other dtype, wrong length) is REPORTED per row, and the caller takes
// numpy's path for it — the formats numpy can write are numpy's business.
#include "../../include/emcid_host.h"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {

inline uint32_t le16(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t le32(const unsigned char* p) { return le16(p) | (le16(p + 2) << 16); }

// value text after `'key':` inside an npy header dict, or nullptr
const char* header_value(const std::string& h, const char* key) {
    const std::string k = std::string("'") + key + "'";
    size_t p = h.find(k);
    if (p == std::string::npos) return nullptr;
    p = h.find(':', p + k.size());
    if (p == std::string::npos) return nullptr;
    ++p;
    while (p < h.size() && h[p] == ' ') ++p;
    return h.c_str() + p;
}

// 0 = row written; 1 = no such file; 2 = not servable here (caller: numpy).  `buf` is the thread's scratch.
int read_one(const char* path, const char* member, int64_t width, float* out, std::vector<unsigned char>& buf) {
    // O_NOATIME: the first read of a freshly written file moves its access time (relatime), i.e. dirties the inode and takes a
    // journal handle — and that waits while the file system commits a large transaction (a burst of file creations a few seconds
    // earlier: bench.py's 26 000 cache files; profiles/r05_outlier.txt: one call in twenty 150-210 ms, all of it in these reads).
    // The flag needs the file's owner (or CAP_FOWNER): EPERM -> the plain open.
    int fd = open(path, O_RDONLY | O_CLOEXEC | O_NOATIME);
    if (fd < 0 && errno == EPERM) fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return errno == ENOENT ? 1 : 2;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 64 || st.st_size > (1 << 26)) {
        close(fd);
        return 2;
    }
    buf.resize((size_t)st.st_size);
    size_t got = 0;
    while (got < buf.size()) {
        const ssize_t r = read(fd, buf.data() + got, buf.size() - got);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) break;
        got += (size_t)r;
    }
    close(fd);
    if (got != buf.size()) return 2;
    const unsigned char* b = buf.data();
    const size_t n = buf.size();
    // zip local file header: signature, method 0 (stored), member name = "<member>.npy"
    if (memcmp(b, "PK\x03\x04", 4) != 0 || le16(b + 8) != 0) return 2;
    const size_t n_name = le16(b + 26), n_extra = le16(b + 28), m_len = strlen(member);
    if (n_name != m_len + 4 || 30 + n_name + n_extra + 12 > n) return 2;
    if (memcmp(b + 30, member, m_len) != 0 || memcmp(b + 30 + m_len, ".npy", 4) != 0) return 2;
    size_t o = 30 + n_name + n_extra;
    if (memcmp(b + o, "\x93NUMPY", 6) != 0) return 2;
    const unsigned major = b[o + 6];
    size_t hlen;
    if (major == 1) {
        hlen = le16(b + o + 8);
        o += 10;
    } else if (major == 2 || major == 3) {
        hlen = le32(b + o + 8);
        o += 12;
    } else {
        return 2;
    }
    if (o + hlen > n) return 2;
    const std::string header((const char*)b + o, hlen);
    o += hlen;
    const char* descr = header_value(header, "descr");
    const char* order = header_value(header, "fortran_order");
    const char* shape = header_value(header, "shape");
    if (!descr || !order || !shape) return 2;
    int item;
    if (strncmp(descr, "'<f4'", 5) == 0) item = 4;
    else if (strncmp(descr, "'<f8'", 5) == 0) item = 8;
    else return 2;
    if (*shape != '(') return 2;
    int64_t count = 1, dims = 0, lead = 1;
    const char* p = shape + 1;
    while (*p && *p != ')') {
        while (*p == ' ' || *p == ',') ++p;
        if (*p == ')') break;
        if (*p < '0' || *p > '9') return 2;
        char* end = nullptr;
        const long long v = strtoll(p, &end, 10);
        if (end == p || v < 0 || v > (1 << 26)) return 2;
        if (dims == 0) lead = v;
        count *= v;
        ++dims;
        if (count > (1 << 26)) return 2;
        p = end;
    }
    if (*p != ')') return 2;
    // a vector of `width`, or the (1, width) layout of the reference's use_new_compute_z files (:951-968); a Fortran-ordered
    // vector is the same bytes
    const bool vec = dims == 1 && count == width;
    const bool row = dims == 2 && lead == 1 && count == width;
    if (!vec && !row) return 2;
    (void)order;      // a vector (or one row) has the same bytes in either order
    if (o + (size_t)count * item > n) return 2;
    if (item == 4) {
        memcpy(out, b + o, (size_t)width * 4);
    } else {
        for (int64_t i = 0; i < width; ++i) {
            double v;
            memcpy(&v, b + o + (size_t)i * 8, 8);
            out[i] = (float)v;      // numpy's astype(float32): round to nearest even, like this conversion
        }
    }
    return 0;
}

}  // namespace

extern "C" int64_t emcid_read_npz_rows_f32(const char* paths, const int64_t* off, int64_t n, const char* member, int64_t width,
                                           float* out, int64_t ld, uint8_t* status, int32_t n_threads) {
    if (!paths || !off || n < 0 || !member || width <= 0 || !out || ld < width || !status) return -1;
    if (n == 0) return 0;
    int nt = n_threads < 1 ? 1 : n_threads > 64 ? 64 : n_threads;
    if ((int64_t)nt > (n + 63) / 64) nt = (int)((n + 63) / 64);      // at least 64 files per thread
    // no exception leaves a worker (std::terminate would take the whole Python process down) or the calling thread while workers
    // are joinable: a failed allocation marks the rest of that thread's files "not served" (status 2: they take the numpy path)
    auto work = [&](int64_t lo, int64_t hi) noexcept {
        int64_t i = lo;
        try {
            std::vector<unsigned char> buf;
            std::string path;
            for (; i < hi; ++i) {
                path.assign(paths + off[i], (size_t)(off[i + 1] - off[i]));
                status[i] = (uint8_t)read_one(path.c_str(), member, width, out + i * ld, buf);
            }
        } catch (...) {
            for (; i < hi; ++i) status[i] = 2;
        }
    };
    if (nt == 1) {
        work(0, n);
    } else {
        std::vector<std::thread> pool;
        int started = 0;
        try {
            pool.reserve((size_t)nt - 1);
            for (int t = 1; t < nt; ++t, ++started) pool.emplace_back(work, n * t / nt, n * (t + 1) / nt);
        } catch (...) {
        }
        work(0, n / nt);
        for (auto& th : pool) th.join();
        for (int t = started + 1; t < nt; ++t) work(n * t / nt, n * (t + 1) / nt);      // threads that could not be started
    }
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) bad += status[i] != 0;
    return bad;
}
