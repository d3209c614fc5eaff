// ubench_l2_same_line.hip -- does a lane's SECOND load from the cache line it has just requested cost another L2 request?
// (What a filter sector of 24 or 32 bytes for six or eight windows would do: global_load_dwordx4 + global_load_dwordx2/x4 at +16.)
// Lanes touch distinct random lines of an L2-resident 2 MB table; variants: one 16-byte load per line; two 16-byte loads from the
// same 32-byte half of the line, issued back to back; two loads from two DIFFERENT lines (the price of two requests).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/ubench_l2_same_line.hip -o /tmp/ubench_sl && /tmp/ubench_sl
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(64, 8) void k(const uint4* __restrict__ tab, uint32_t mask, int iters, uint32_t* out) {
    uint32_t x = (blockIdx.x * 64u + threadIdx.x) * 2654435761u + 0x9E3779B9u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        const uint32_t i = ((x >> 7) & mask) & ~1u; // even entry: the start of a 32-byte half line
        uint4 a = tab[i], b = a;
        if (MODE == 1) b = tab[i + 1];                                   // same 32 bytes, next 16
        if (MODE == 2) b = tab[((x * 2246822519u) >> 7) & mask];         // another line
        acc += a.x ^ a.w ^ b.y;
    }
    if (acc == 0x12345u) out[threadIdx.x] = acc;
}

template <int MODE>
void run(const char* what, const uint4* d_tab, size_t entries, uint32_t* d_out) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int waves = p.multiProcessorCount * 32, iters = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, d_tab, (uint32_t)(entries - 1), iters, d_out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, d_tab, (uint32_t)(entries - 1), iters, d_out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double lanes = (double)waves * 64 * iters * 5;
    printf("%-58s %.3g lane-iterations/s\n", what, lanes / (ms * 1e-3));
}

int main() {
    uint32_t* d_out; hipMalloc(&d_out, 4096);
    const size_t entries = 2 * 65536; // 2 MB
    uint4* d; hipMalloc(&d, entries * 16); hipMemset(d, 1, entries * 16);
    run<0>("one 16-byte load per iteration", d, entries, d_out);
    run<1>("two loads, same 32 bytes of one line", d, entries, d_out);
    run<2>("two loads, two lines", d, entries, d_out);
    return 0;
}
