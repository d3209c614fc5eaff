// tools/ubench/out_write.cpp -- how fast a 1.4 GB result file can be written: N threads of pwrite (mode 0) or memcpy into a shared mapping (mode 1)
// g++ -O2 -pthread -o /tmp/out_write tools/ubench/out_write.cpp && /tmp/out_write /tmp/x.bin 3 0
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <chrono>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
using namespace std; using clk = chrono::steady_clock;
int main(int argc, char** argv) {
    const char* path = argv[1]; int nt = atoi(argv[2]); int mode = atoi(argv[3]); size_t total = (size_t)1400 << 20, blk = (size_t)32 << 20;
    vector<char> src(blk, 'x');
    unlink(path);
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    auto t0 = clk::now();
    char* map = nullptr;
    if (mode == 1) { ftruncate(fd, total); map = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); }
    if (mode == 2) { posix_fallocate(fd, 0, total); }
    vector<thread> th;
    size_t nb = total / blk;
    for (int t = 0; t < nt; ++t) th.emplace_back([&, t] {
        for (size_t b = t; b < nb; b += nt) {
            if (mode == 1) memcpy(map + b * blk, src.data(), blk);
            else { size_t d = 0; while (d < blk) { ssize_t n = pwrite(fd, src.data() + d, blk - d, b * blk + d); if (n <= 0) abort(); d += n; } }
        }
    });
    for (auto& x : th) x.join();
    if (map) munmap(map, total);
    double s = chrono::duration<double>(clk::now() - t0).count();
    close(fd);
    double s2 = chrono::duration<double>(clk::now() - t0).count();
    printf("%s threads %d mode %d: %.3f s (%.1f GB/s), with close %.3f\n", path, nt, mode, s, total / s / 1e9, s2);
    auto t1 = clk::now(); unlink(path); printf("  unlink %.3f s\n", chrono::duration<double>(clk::now() - t1).count());
}
