// Microbenchmark: VALU ceiling of canonical-murmur hashing on gfx950 (no memory traffic).
// Each thread hashes ITER synthetic 16-byte windows (fwd+rc = 2 murmur3_x64_128 each) from registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../rkmh_amd/csrc/rk_device.hpp"
using namespace rk;

template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t* out, int iters, uint32_t seed) {
    uint64_t a = threadIdx.x * 0x9E3779B97F4A7C15ull + blockIdx.x, b = a ^ 0x1234567887654321ull;
    uint64_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint64_t h1 = seed, h2 = seed, g1 = seed, g2 = seed;
        mm_block(h1, h2, a, b);
        uint64_t f = mm_finish(h1, h2, 16, 0);
        if (MODE == 0) {
            mm_block(g1, g2, ~b, ~a);
            uint64_t r = mm_finish(g1, g2, 16, 0);
            acc += f < r ? f : r;
        } else acc += f;
        a += 0x632BE59BD9B4E019ull; b ^= acc;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    uint64_t* d; hipMalloc(&d, 256 * 8192 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 1; waves <= 8; waves *= 2) {
        int blocks = 256 * waves; // 4 waves per block => `waves` waves per SIMD
        int iters = 2000;
        hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 10, 42u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 42u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double nh = (double)blocks * 256 * iters;
        printf("waves/SIMD=%d canonical hashes/s = %.3e (%.2f cycles@2.4GHz per wave-hash-pair per SIMD)\n", waves, nh / (ms * 1e-3),
               (ms * 1e-3) * 2.4e9 / ((double)iters * waves));
    }
    return 0;
}
