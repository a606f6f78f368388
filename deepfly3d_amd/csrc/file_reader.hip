// Host side of the input front-end: the files of one device batch -> one staging buffer, laid out as
// df3d_jpeg_decode_luma expects them (every file at a 16-byte aligned offset), read by a few native threads.  The reference
// leaves this to df2d's DataLoader worker processes; here a batch of 896 camera frames (65 MB) has to arrive in well under
// the 42 ms its hourglass takes, which one Python thread per file cannot do (the interpreter lock serialises the ~1 000
// open / read / close calls).  No device work in this file.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cstring>
#include <thread>
#include <vector>

#include "common.h"

namespace {

struct ReadJob {
    const char* const* paths;
    int n;
    std::vector<long long> size;
    std::atomic<int> next{0};
    std::atomic<int> failed{-1};  // index of the first file that failed
    int err = 0;
};

template <class F>
void run_threads(int threads, F&& body) {
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(body);
    body();
    for (auto& th : pool) th.join();
}

void fail(ReadJob& job, int i, int err) {
    int none = -1;
    if (job.failed.compare_exchange_strong(none, i)) job.err = err;
}

}  // namespace

extern "C" int df3d_read_files(const char* const* paths, int n, unsigned char* dst, size_t dst_bytes, unsigned* starts, unsigned* sizes,
                               size_t* total_bytes, int threads) {
    DF3D_CHECK_ARG(n >= 0 && total_bytes, "bad arguments");
    *total_bytes = 0;
    if (n == 0) return DF3D_OK;
    DF3D_CHECK_ARG(paths && starts && sizes, "null pointer");
    threads = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
    if (threads > n) threads = n;
    ReadJob job;
    job.paths = paths;
    job.n = n;
    job.size.assign(n, 0);
    // 1. sizes (stat, not open: two batches of ~900 files in flight must not need ~1 800 descriptors)
    run_threads(threads, [&]() {
        for (int i; (i = job.next.fetch_add(1)) < n;) {
            struct stat st;
            if (stat(paths[i], &st) != 0) {
                fail(job, i, errno);
                continue;
            }
            if (!S_ISREG(st.st_mode)) {
                fail(job, i, EINVAL);
                continue;
            }
            job.size[i] = (long long)st.st_size;
        }
    });
    if (job.failed.load() >= 0) {
        df3d::set_error("df3d_read_files: %s: %s", paths[job.failed.load()], strerror(job.err));
        return DF3D_EIO;
    }
    // 2. layout: 16-byte aligned offsets in path order
    unsigned long long off = 0;
    for (int i = 0; i < n; ++i) {
        if (off + (unsigned long long)job.size[i] + 64 >= (1ull << 32)) {
            df3d::set_error("df3d_read_files: more than 4 GiB of file data in one batch");
            return DF3D_EINVAL;
        }
        starts[i] = (unsigned)off;
        sizes[i] = (unsigned)job.size[i];
        off += ((unsigned long long)job.size[i] + 15ull) / 16ull * 16ull;
    }
    *total_bytes = (size_t)off;
    if (!dst || off + 16 > dst_bytes) return DF3D_ENOSPC;  // the caller learns the size and comes back with a larger buffer
    // 3. read (one descriptor per thread at a time); a file that changed size since step 1 is an error
    job.next.store(0);
    run_threads(threads, [&]() {
        for (int i; (i = job.next.fetch_add(1)) < n;) {
            const int fd = open(paths[i], O_RDONLY | O_CLOEXEC);
            if (fd < 0) {
                fail(job, i, errno);
                continue;
            }
            unsigned char* p = dst + starts[i];
            long long left = job.size[i];
            while (left > 0) {
                const ssize_t got = read(fd, p, (size_t)left);
                if (got < 0 && errno == EINTR) continue;
                if (got <= 0) {  // error, or the file shrank under us
                    fail(job, i, got < 0 ? errno : EIO);
                    break;
                }
                p += got;
                left -= got;
            }
            close(fd);
            const unsigned long long pad = ((unsigned long long)job.size[i] + 15ull) / 16ull * 16ull - (unsigned long long)job.size[i];
            if (left == 0 && pad) memset(p, 0, (size_t)pad);
        }
    });
    memset(dst + off, 0, 16);
    if (job.failed.load() >= 0) {
        df3d::set_error("df3d_read_files: %s: %s", paths[job.failed.load()], strerror(job.err));
        return DF3D_EIO;
    }
    return DF3D_OK;
}
