// How long does page-locking an already-touched 256 MB host buffer take (hipHostRegister), against hipHostMalloc of the same size
// and against the staging memcpy it would save?  Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/host_register tools/ubench/host_register.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = (size_t)256 << 20;
    hipFree(0);
    void* d = nullptr; hipMalloc(&d, N);
    for (int rep = 0; rep < 3; ++rep) {
        char* p = (char*)aligned_alloc(4096, N);
        double t0 = now(); memset(p, 1, N); double t1 = now();
        hipError_t e = hipHostRegister(p, N, hipHostRegisterDefault); double t2 = now();
        hipMemcpy(d, p, N, hipMemcpyHostToDevice); double t3 = now();
        hipMemcpy(d, p, N, hipMemcpyHostToDevice); double t4 = now();
        hipHostUnregister(p); double t5 = now();
        hipMemcpy(d, p, N, hipMemcpyHostToDevice); double t6 = now();
        free(p);
        void* q = nullptr; double t7 = now(); hipHostMalloc(&q, N, hipHostMallocDefault); double t8 = now(); memset(q, 1, N); double t9 = now(); hipHostFree(q);
        printf("rep %d: first touch %.1f ms, hipHostRegister %.1f ms (%s), H2D registered %.1f / %.1f ms, unregister %.1f ms, H2D pageable %.1f ms, hipHostMalloc %.1f ms + touch %.1f ms\n",
               rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, hipGetErrorString(e), (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3, (t8 - t7) * 1e3, (t9 - t8) * 1e3);
    }
    return 0;
}
