#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) P16 { u4 v; };
__global__ void k(uint32_t* out, int off) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) buf[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const P16* p = reinterpret_cast<const P16*>(buf + off + threadIdx.x);
    u4 v = p->v;
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
// same via inline asm: force one ds_read_b128 on an arbitrary byte address
__global__ void k2(uint32_t* out, int off) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) buf[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    uint32_t addr = (uint32_t)(uintptr_t)(buf) + off + threadIdx.x;
    u4 v;
    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
__global__ void k3(uint32_t* out, uint32_t a, uint32_t b) { // alignbyte with shift operand > 3
    uint32_t sh = threadIdx.x;
    out[threadIdx.x] = __builtin_amdgcn_alignbyte(a, b, sh);
}
int main() {
    uint32_t* d; hipMalloc(&d, 64 * 16);
    uint32_t h[256];
    for (int which = 0; which < 2; ++which) {
        int bad = 0;
        for (int off = 0; off < 8; ++off) {
            hipMemset(d, 0, 64 * 16);
            if (which == 0) hipLaunchKernelGGL(k, 1, 64, 0, 0, d, off); else hipLaunchKernelGGL(k2, 1, 64, 0, 0, d, off);
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) { printf("kernel %d off %d: %s\n", which, off, hipGetErrorString(e)); return 1; }
            hipMemcpy(h, d, 64 * 16, hipMemcpyDeviceToHost);
            for (int t = 0; t < 64; ++t) for (int j = 0; j < 16; ++j) {
                uint8_t want = (uint8_t)((off + t + j) * 7 + 3);
                uint8_t got = (uint8_t)(h[t * 4 + j / 4] >> (8 * (j % 4)));
                if (want != got) ++bad;
            }
        }
        printf("kernel %d: %d bad bytes\n", which, bad);
    }
    hipLaunchKernelGGL(k3, 1, 64, 0, 0, d, 0x44332211u, 0xddccbbaau);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 64 * 4, hipMemcpyDeviceToHost);
    for (int t = 0; t < 9; ++t) printf("alignbyte sh=%d -> %08x\n", t, h[t]);
    return 0;
}
