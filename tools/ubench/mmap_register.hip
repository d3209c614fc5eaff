// Can the DMA engine read a FASTQ file straight out of the page cache?  The file is mapped read-only, the mapping page-locked with
// hipHostRegister (read-only flag first, default flags second), and uploaded in 16 MB pieces; beside it the shipped route: pread
// into a page-locked buffer, then the same uploads.  Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/mmap_register tools/ubench/mmap_register.hip
// Usage: /tmp/mmap_register <file>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    if (argc < 2) return 1;
    const int fd = open(argv[1], O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) { perror("open"); return 1; }
    const size_t N = (size_t)st.st_size, P = (size_t)16 << 20;
    hipFree(0);
    void* d = nullptr;
    if (hipMalloc(&d, P * 4) != hipSuccess) return 1;
    hipStream_t s; hipStreamCreate(&s);
    // warm the page cache
    { char* b = (char*)malloc(P); for (size_t o = 0; o < N; o += P) if (pread(fd, b, P, (off_t)o) < 0) return 1; free(b); }
    // route A: pread into a page-locked buffer (two of them, alternating), upload
    for (int rep = 0; rep < 2; ++rep) {
        void* h[2]; hipHostMalloc(&h[0], P); hipHostMalloc(&h[1], P);
        double t0 = now();
        int k = 0;
        for (size_t o = 0; o < N; o += P, k ^= 1) {
            const size_t n = N - o < P ? N - o : P;
            if (pread(fd, h[k], n, (off_t)o) != (ssize_t)n) return 1;
            hipMemcpyAsync((char*)d + (k ? P : 0), h[k], n, hipMemcpyHostToDevice, s);
            if (k) hipStreamSynchronize(s);
        }
        hipStreamSynchronize(s);
        double t1 = now();
        printf("pread + upload (one thread):            %.3f s  %.1f GB/s\n", t1 - t0, N / (t1 - t0) / 1e9);
        hipHostFree(h[0]); hipHostFree(h[1]);
    }
    // route B: the mapping itself, registered
    for (unsigned flags : {0x08u /* hipHostRegisterReadOnly */, 0u}) {
        void* m = mmap(nullptr, N, PROT_READ, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) { perror("mmap"); return 1; }
        double t0 = now();
        hipError_t e = hipHostRegister(m, N, flags);
        double t1 = now();
        printf("hipHostRegister(mapping, flags=%#x): %s, %.3f s\n", flags, hipGetErrorString(e), t1 - t0);
        if (e == hipSuccess) {
            for (int rep = 0; rep < 2; ++rep) {
                double t2 = now();
                int k = 0;
                for (size_t o = 0; o < N; o += P, k = (k + 1) & 3) {
                    const size_t n = N - o < P ? N - o : P;
                    hipMemcpyAsync((char*)d + (size_t)k * P, (char*)m + o, n, hipMemcpyHostToDevice, s);
                }
                hipStreamSynchronize(s);
                double t3 = now();
                printf("upload from the registered mapping:      %.3f s  %.1f GB/s\n", t3 - t2, N / (t3 - t2) / 1e9);
            }
            double t4 = now(); hipHostUnregister(m); printf("hipHostUnregister: %.3f s\n", now() - t4);
        } else (void)hipGetLastError();
        munmap(m, N);
    }
    return 0;
}
