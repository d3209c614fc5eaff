// ubench_l2_requests.hip -- how many DISTINCT-LINE 16-byte requests per second this chip serves out of tables that live in the L2s
// (the access pattern of k_classify_kmer's filter sectors and map buckets: every lane of a wave its own cache line), by table size
// and by loads in flight per lane.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/ubench_l2_requests.hip -o /tmp/ubench_l2 && /tmp/ubench_l2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int INFLIGHT>
__global__ __launch_bounds__(64, 8) void k(const uint4* __restrict__ tab, uint32_t mask, int iters, uint32_t* out) {
    uint32_t x[INFLIGHT];
#pragma unroll
    for (int j = 0; j < INFLIGHT; ++j) x[j] = (blockIdx.x * 64u + threadIdx.x) * 2654435761u + 0x9E3779B9u * (j + 1);
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint4 v[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) { x[j] = x[j] * 1664525u + 1013904223u; v[j] = tab[(x[j] >> 7) & mask]; }
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) acc += v[j].x ^ v[j].w;
    }
    if (acc == 0x12345u) out[threadIdx.x] = acc;
}

template <int INFLIGHT>
void run(const uint4* d_tab, size_t entries, uint32_t* d_out) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int waves = p.multiProcessorCount * 32, iters = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<INFLIGHT>, dim3(waves), dim3(64), 0, 0, d_tab, (uint32_t)(entries - 1), iters, d_out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<INFLIGHT>, dim3(waves), dim3(64), 0, 0, d_tab, (uint32_t)(entries - 1), iters, d_out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double req = (double)waves * 64 * iters * INFLIGHT * 5;
    printf("table %7.2f MB  %d loads in flight per lane: %.3g requests/s (%.2f per clock and L2 channel at 128 channels x %.2f GHz)\n",
           entries * 16 / 1048576.0, INFLIGHT, req / (ms * 1e-3), req / (ms * 1e-3) / 128 / (p.clockRate * 1e3), p.clockRate / 1e6);
}

int main() {
    uint32_t* d_out; hipMalloc(&d_out, 4096);
    for (size_t mb : {1, 2, 4, 8, 32, 256}) {
        const size_t entries = mb * 65536; // 16-byte entries; sizes are powers of two
        uint4* d; hipMalloc(&d, entries * 16); hipMemset(d, 1, entries * 16);
        run<1>(d, entries, d_out); run<2>(d, entries, d_out); run<4>(d, entries, d_out);
        hipFree(d);
    }
    return 0;
}
