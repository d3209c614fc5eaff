// file_write.cpp -- how fast one regular file takes 2 GB of output on this box, by method: what bounds `bin/rkmh stream > calls.tsv`.
//   g++ -O2 -pthread tools/ubench/file_write.cpp -o /tmp/file_write && /tmp/file_write /tmp/fw.bin
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "/tmp/fw.bin";
    const size_t total = (size_t)2 << 30, chunk = (size_t)2 << 20;
    char* buf = nullptr;
    if (posix_memalign((void**)&buf, 4096, chunk)) return 1;
    memset(buf, 'x', chunk);
    auto run = [&](const char* tag, int flags, int threads, bool prealloc) {
        unlink(path);
        int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | flags, 0644);
        if (fd < 0) { printf("%-44s open failed\n", tag); return; }
        if (prealloc && posix_fallocate(fd, 0, (off_t)total) != 0) { printf("%-44s fallocate failed\n", tag); close(fd); return; }
        const double t = now();
        std::vector<std::thread> th;
        bool bad = false;
        for (int i = 0; i < threads; ++i)
            th.emplace_back([&, i] {
                for (size_t off = (size_t)i * chunk; off < total; off += (size_t)threads * chunk)
                    if (pwrite(fd, buf, chunk, (off_t)off) != (ssize_t)chunk) { bad = true; return; }
            });
        for (auto& x : th) x.join();
        const double dt = now() - t;
        close(fd);
        printf("%-44s %s %.2f GB/s\n", tag, bad ? "FAILED" : "ok", total / dt / 1e9);
    };
    run("buffered pwrite, 1 thread", 0, 1, false);
    run("buffered pwrite, 4 threads", 0, 4, false);
    run("buffered pwrite, 8 threads", 0, 8, false);
    run("buffered pwrite, 1 thread, preallocated", 0, 1, true);
    run("buffered pwrite, 4 threads, preallocated", 0, 4, true);
    run("O_DIRECT pwrite, 1 thread", O_DIRECT, 1, false);
    run("O_DIRECT pwrite, 4 threads", O_DIRECT, 4, false);
    run("O_DIRECT pwrite, 4 threads, preallocated", O_DIRECT, 4, true);
    unlink(path);
    return 0;
}
