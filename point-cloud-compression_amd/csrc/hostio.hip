// hostio.hip -- the three files of every cloud, written from / read into the packed stream buffer of one batch (pccx/codec.py
// packed_layout) by a small pool of host threads.  Pure host code (no HIP call): compress.py:139-152 writes <name>.p.bin, <name>.s.bin and
// <name>.c.bin INSIDE its timer and decompress.py:80-91,113 reads them back inside its own, one open/write/close per file from Python;
// for a batch of 1024 clouds that is 6144 small-file operations per step, 70 ms of interpreter time (gpurun_out/r3d/files.json) beside a
// 37 ms GPU step.  Here a batch's files are cut from the ONE host copy of the packed buffer (the bytes are never re-assembled per cloud)
// by threads that share an atomic cloud counter.  Formats are the reference's: .s.bin = the first s_nbytes[b] bytes
// of row b of s_bytes, .p.bin = the first p_nbytes[b] bytes of row b of p_bytes, .c.bin = 16 bytes [cx, cy, cz, longest] fp32.
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace {

// byte offsets of the five sections (codec.packed_layout): s_nbytes (B) i32 | p_nbytes (B) i32 | c (B,4) f32 | s_bytes (B,s_stride) |
// p_bytes (B,p_cap)
struct Layout {
    size_t sn, pn, c, sb, pb, end;
    Layout(int B, int s_stride, int p_cap)
    {
        sn = 0, pn = 4 * (size_t)B, c = 8 * (size_t)B, sb = c + 16 * (size_t)B;
        pb = sb + (size_t)B * s_stride, end = pb + (size_t)B * p_cap;
    }
};

// A pool that lives as long as the process (allocated once, never destroyed: its threads sleep on the condition variable at exit).
// run(n, threads, fn): fn(i) for every i < n, on `threads` threads counting the caller; returns when all are done.
class Pool {
public:
    static Pool &get()
    {
        static Pool *p = new Pool();
        return *p;
    }
    void run(int n, int threads, const std::function<void(int)> &fn)
    {
        if (threads > n) threads = n;
        if (threads <= 1) {
            for (int i = 0; i < n; ++i) fn(i);
            return;
        }
        std::unique_lock<std::mutex> call(call_mu_);          // one batch at a time
        grow(threads - 1);
        {
            std::lock_guard<std::mutex> g(mu_);
            fn_ = &fn, n_ = n, next_.store(0), want_ = threads - 1, running_ = 0, ++epoch_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> g(mu_);
        done_.wait(g, [&] { return want_ == 0 && running_ == 0; });
        fn_ = nullptr;
    }

private:
    void work()
    {
        for (int i; (i = next_.fetch_add(1)) < n_;) (*fn_)(i);
    }
    void grow(int workers)
    {
        while ((int)th_.size() < workers) {
            th_.emplace_back([this] {
                uint64_t seen = 0;
                for (;;) {
                    {
                        std::unique_lock<std::mutex> g(mu_);
                        cv_.wait(g, [&] { return epoch_ != seen && want_ > 0; });
                        seen = epoch_, --want_, ++running_;
                    }
                    work();
                    {
                        std::lock_guard<std::mutex> g(mu_);
                        --running_;
                    }
                    done_.notify_all();
                }
            });
            th_.back().detach();
        }
    }
    std::mutex call_mu_, mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> th_;
    const std::function<void(int)> *fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, want_ = 0, running_ = 0;
    uint64_t epoch_ = 0;
};

int default_threads()
{
    unsigned hw = std::thread::hardware_concurrency();
    return hw == 0 ? 4 : (hw > 8 ? 8 : (int)hw);
}

bool write_all(const char *path, const void *buf, size_t len, std::string &err)
{
    // an existing file first: O_CREAT makes the kernel take the DIRECTORY's lock exclusively for the lookup, which serialises the
    // threads of a batch that rewrites one directory (the steady state of a codec loop)
    int fd = open(path, O_WRONLY | O_TRUNC | O_CLOEXEC);
    if (fd < 0 && errno == ENOENT) fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd < 0) {
        err = std::string("open ") + path + ": " + strerror(errno);
        return false;
    }
    const char *p = (const char *)buf;
    while (len > 0) {
        ssize_t w = write(fd, p, len);
        if (w < 0) {
            if (errno == EINTR) continue;
            err = std::string("write ") + path + ": " + strerror(errno);
            close(fd);
            return false;
        }
        p += w, len -= (size_t)w;
    }
    if (close(fd) != 0) {
        err = std::string("close ") + path + ": " + strerror(errno);
        return false;
    }
    return true;
}

// the whole file into buf (at most cap bytes): its length, or -1 (error / larger than cap)
long read_all(const char *path, void *buf, size_t cap, std::string &err)
{
    int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
        err = std::string("open ") + path + ": " + strerror(errno);
        return -1;
    }
    size_t got = 0;
    char *p = (char *)buf;
    char extra;
    for (;;) {
        ssize_t r = got < cap ? read(fd, p + got, cap - got) : read(fd, &extra, 1);
        if (r < 0) {
            if (errno == EINTR) continue;
            err = std::string("read ") + path + ": " + strerror(errno);
            close(fd);
            return -1;
        }
        if (r == 0) break;
        if (got >= cap) {
            err = std::string(path) + ": longer than the " + std::to_string(cap) + " bytes the stream buffer holds per cloud";
            close(fd);
            return -1;
        }
        got += (size_t)r;
    }
    close(fd);
    return (long)got;
}

struct Names {
    const char *dir, *names;
    const int64_t *off;
    std::string path(int b, const char *ext) const
    {
        std::string s(dir);
        if (!s.empty() && s.back() != '/') s.push_back('/');
        s += names + off[b];
        s += ext;
        return s;
    }
};

int check_common(const void *packed, int B, int s_stride, int p_cap, const char *dir, const char *names, const int64_t *off, const char *who)
{
    PCCX_CHECK_ARG(packed && dir && names && off, "%s: null pointer", who);
    PCCX_CHECK_ARG(B >= 0 && s_stride >= 0 && p_cap >= 0, "%s: negative size", who);
    return PCCX_OK;
}

}  // namespace

extern "C" int pccx_write_streams_host(const void *packed_host, int B, int s_stride, int p_cap, const char *dir, const char *names,
                                       const int64_t *name_off, int threads)
{
    int rc = check_common(packed_host, B, s_stride, p_cap, dir, names, name_off, "pccx_write_streams_host");
    if (rc != PCCX_OK) return rc;
    const Layout L(B, s_stride, p_cap);
    const uint8_t *base = (const uint8_t *)packed_host;
    const int32_t *sn = (const int32_t *)(base + L.sn), *pn = (const int32_t *)(base + L.pn);
    for (int b = 0; b < B; ++b)      // a negative count is the coder's "output buffer too small" (rangecoder.hip): nothing is written
        PCCX_CHECK_ARG(sn[b] >= 0 && sn[b] <= s_stride && pn[b] >= 0 && pn[b] <= p_cap,
                       "pccx_write_streams_host: cloud %d has byte counts %d / %d outside its rows of %d / %d bytes (a negative count "
                       "means the coder's output buffer was too small)", b, sn[b], pn[b], s_stride, p_cap);
    const Names nm{dir, names, name_off};
    std::mutex emu;
    std::string first;
    Pool::get().run(B, threads > 0 ? threads : default_threads(), [&](int b) {
        std::string err;
        // the reference's order (compress.py:139-152): .p.bin, .s.bin, .c.bin
        bool ok = write_all(nm.path(b, ".p.bin").c_str(), base + L.pb + (size_t)b * p_cap, (size_t)pn[b], err) &&
                  write_all(nm.path(b, ".s.bin").c_str(), base + L.sb + (size_t)b * s_stride, (size_t)sn[b], err) &&
                  write_all(nm.path(b, ".c.bin").c_str(), base + L.c + 16 * (size_t)b, 16, err);
        if (!ok) {
            std::lock_guard<std::mutex> g(emu);
            if (first.empty()) first = err;
        }
    });
    if (!first.empty()) {
        pccx_set_error("pccx_write_streams_host: %s", first.c_str());
        return PCCX_ERR_ARG;
    }
    return PCCX_OK;
}

extern "C" int pccx_read_streams_host(void *packed_host, int B, int s_stride, int p_cap, const char *dir, const char *names,
                                      const int64_t *name_off, int threads)
{
    int rc = check_common(packed_host, B, s_stride, p_cap, dir, names, name_off, "pccx_read_streams_host");
    if (rc != PCCX_OK) return rc;
    const Layout L(B, s_stride, p_cap);
    uint8_t *base = (uint8_t *)packed_host;
    int32_t *sn = (int32_t *)(base + L.sn), *pn = (int32_t *)(base + L.pn);
    const Names nm{dir, names, name_off};
    std::mutex emu;
    std::string first;
    Pool::get().run(B, threads > 0 ? threads : default_threads(), [&](int b) {
        std::string err;
        // the reference's order (decompress.py:80-91,113): .s.bin, .p.bin, .c.bin
        long s = read_all(nm.path(b, ".s.bin").c_str(), base + L.sb + (size_t)b * s_stride, (size_t)s_stride, err);
        long p = s < 0 ? -1 : read_all(nm.path(b, ".p.bin").c_str(), base + L.pb + (size_t)b * p_cap, (size_t)p_cap, err);
        long c = p < 0 ? -1 : read_all(nm.path(b, ".c.bin").c_str(), base + L.c + 16 * (size_t)b, 16, err);
        if (c >= 0 && c != 16) err = nm.path(b, ".c.bin") + ": " + std::to_string(c) + " bytes, expected the 16 of [cx, cy, cz, longest]", c = -1;
        if (c < 0) {
            sn[b] = pn[b] = 0;
            std::lock_guard<std::mutex> g(emu);
            if (first.empty()) first = err;
            return;
        }
        sn[b] = (int32_t)s, pn[b] = (int32_t)p;
        // rows are handed to the decoders whole: clear what the files did not fill
        memset(base + L.sb + (size_t)b * s_stride + s, 0, (size_t)s_stride - (size_t)s);
        memset(base + L.pb + (size_t)b * p_cap + p, 0, (size_t)p_cap - (size_t)p);
    });
    if (!first.empty()) {
        pccx_set_error("pccx_read_streams_host: %s", first.c_str());
        return PCCX_ERR_ARG;
    }
    return PCCX_OK;
}

extern "C" int pccx_stream_sizes_host(int B, const char *dir, const char *names, const int64_t *name_off, int64_t *s_sizes, int64_t *p_sizes,
                                      int threads)
{
    PCCX_CHECK_ARG(dir && names && name_off && s_sizes && p_sizes, "pccx_stream_sizes_host: null pointer");
    PCCX_CHECK_ARG(B >= 0, "pccx_stream_sizes_host: negative size");
    const Names nm{dir, names, name_off};
    Pool::get().run(B, threads > 0 ? threads : default_threads(), [&](int b) {
        struct stat st;
        s_sizes[b] = stat(nm.path(b, ".s.bin").c_str(), &st) == 0 ? (int64_t)st.st_size : -1;
        p_sizes[b] = stat(nm.path(b, ".p.bin").c_str(), &st) == 0 ? (int64_t)st.st_size : -1;
    });
    return PCCX_OK;
}

extern "C" size_t pccx_streams_packed_bytes(int B, int s_stride, int p_cap)
{
    if (B < 0 || s_stride < 0 || p_cap < 0) return 0;
    return Layout(B, s_stride, p_cap).end;
}
