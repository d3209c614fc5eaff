// tools/ubench/malloc_vs_kernels.hip -- what hipMalloc / hipFree / hipHostMalloc cost while kernels of another thread's stream are running
// hipcc --offload-arch=gfx950 -O2 -o /tmp/mvk tools/ubench/malloc_vs_kernels.hip -lpthread && /tmp/mvk
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }
__global__ void spin(long cycles, int* out) { // one wave per workgroup busy for `cycles`
    const long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (out && threadIdx.x == 999) *out = 1;
}
int main() {
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    int* d = nullptr;
    hipMalloc(&d, 4);
    for (int mode = 0; mode < 3; ++mode) { // 0: idle GPU; 1: one long single-workgroup kernel at a time; 2: 512 workgroups of 64 KB LDS-free spinning waves
        std::atomic<bool> stop{false};
        std::thread bg;
        if (mode) bg = std::thread([&] {
            while (!stop.load()) {
                hipLaunchKernelGGL(spin, dim3(mode == 1 ? 1 : 512), dim3(64), 0, st, 100000000L / 1, d); // ~50 ms at 2 GHz-ish clock64 (100 MHz counter: adjust below)
                hipStreamSynchronize(st);
            }
        });
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        std::vector<void*> ps;
        double worst = 0, total = 0;
        for (int i = 0; i < 8; ++i) {
            void* p = nullptr;
            const auto a = clk::now();
            hipMalloc(&p, (size_t)1 << 30);
            const double t = ms(a);
            worst = t > worst ? t : worst; total += t;
            ps.push_back(p);
        }
        const auto h0 = clk::now();
        void* hp = nullptr;
        hipHostMalloc(&hp, (size_t)64 << 20, hipHostMallocDefault);
        const double th = ms(h0);
        const auto f0 = clk::now();
        for (void* p : ps) hipFree(p);
        const double tf = ms(f0);
        hipHostFree(hp);
        stop = true;
        if (bg.joinable()) bg.join();
        printf("mode %d: 8 x hipMalloc(1 GB): total %.1f ms, worst %.1f ms; hipHostMalloc(64 MB) %.1f ms; 8 x hipFree %.1f ms\n", mode, total, worst, th, tf);
    }
    // how long one spin kernel really takes
    const auto a = clk::now();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 100000000L, d);
    hipStreamSynchronize(st);
    printf("one spin kernel: %.1f ms\n", ms(a));
    return 0;
}
