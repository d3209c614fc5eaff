// ubench_valu_rates.hip -- issue cost (cycles per wave64 instruction on one SIMD) of the VALU / SALU / LDS instructions k_classify_kmer is
// made of, measured with every SIMD of the chip full (8 waves per SIMD, independent chains).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/ubench_valu_rates.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(64, 8) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t v[16], w[16];
    uint64_t q[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = seed * (i + 3) + threadIdx.x; w[i] = v[i] ^ 0x5bd1e995u; q[i] = ((uint64_t)v[i] << 32) | w[i]; }
    const uint32_t c = seed | 0x9E3779B1u, one = 1u;
    __shared__ uint32_t lds[1024];
    lds[threadIdx.x] = 0;
    for (int it = 0; it < iters; ++it) {
#define A(i) \
        if (OP == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[i]) : "v"(w[i])); \
        if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c)); \
        if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c)); \
        if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(v[i]), "v"(c) : "vcc"); \
        if (OP == 4) asm volatile("v_lshl_add_u64 %0, %0, 4, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 15])); \
        if (OP == 5) asm volatile("v_lshrrev_b64 %0, 28, %0" : "+v"(q[i])); \
        if (OP == 6) asm volatile("v_bfe_u32 %0, %0, %1, 5" : "+v"(v[i]) : "v"(w[i])); \
        if (OP == 7) asm volatile("v_alignbit_b32 %0, %0, %1, 6" : "+v"(v[i]) : "v"(w[i])); \
        if (OP == 8) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w[i]), "v"(c)); \
        if (OP == 9) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(v[i]) : "v"(w[i])); \
        if (OP == 10) asm volatile("v_lshlrev_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(v[i]) : "v"(one)); \
        if (OP == 11) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(v[i]) : "v"(w[i]), "v"(c)); \
        if (OP == 12) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(c)); \
        if (OP == 13) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(w[i])); \
        if (OP == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(w[i]) : ); \
        if (OP == 15) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(v[i]), "v"(w[i]) : "vcc"); \
        if (OP == 16) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(v[i]) : "s20"); \
        if (OP == 17) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i])); \
        if (OP == 18) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(v[i]) : "v"(w[i])); \
        if (OP == 19) asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc"); \
        if (OP == 20) asm volatile("s_bcnt1_i32_b64 s20, vcc" : : : "s20", "scc"); \
        if (OP == 21) asm volatile("v_bfrev_b32 %0, %0" : "+v"(v[i])); \
        if (OP == 22) asm volatile("v_min_u32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i])); \
        if (OP == 23) asm volatile("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(v[i]));
        REP16(A)
#undef A
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + (uint32_t)q[i] + (uint32_t)(q[i] >> 32);
    if (s == 0x12345u) out[threadIdx.x] = s + lds[threadIdx.x];
}

template <int OP>
double run(const char* name, uint32_t* d_out) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int waves = p.multiProcessorCount * 32, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(waves), dim3(64), 0, 0, d_out, iters, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(waves), dim3(64), 0, 0, d_out, iters, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)waves / (p.multiProcessorCount * 4) * iters * 16 * 5; // instructions issued by one SIMD
    const double cyc = ms * 1e-3 * (p.clockRate * 1e3) / per_simd;
    printf("%-28s %.2f cycles per wave64 instruction per SIMD (%.3f ms)\n", name, cyc, ms / 5);
    return cyc;
}

int main() {
    uint32_t* d; hipMalloc(&d, 4096);
    run<0>("v_add_u32", d); run<1>("v_mul_lo_u32", d); run<2>("v_mul_hi_u32", d); run<3>("v_mad_u64_u32", d);
    run<4>("v_lshl_add_u64", d); run<5>("v_lshrrev_b64", d); run<6>("v_bfe_u32", d); run<7>("v_alignbit_b32", d);
    run<8>("v_perm_b32", d); run<9>("v_mbcnt_lo_u32_b32", d); run<10>("v_lshlrev_b32_sdwa", d); run<11>("v_bitop3_b32", d);
    run<12>("v_mul_u32_u24", d); run<13>("v_mad_u32_u24", d); run<14>("v_cndmask_b32", d); run<15>("v_cmp_eq_u32", d);
    run<16>("v_readlane_b32", d); run<17>("v_add_u32_dpp row_shr", d); run<18>("v_lshl_add_u32", d);
    run<19>("s_add_u32", d); run<20>("s_bcnt1_i32_b64", d); run<21>("v_bfrev_b32", d); run<22>("v_min_u32", d); run<23>("v_xor_b32_sdwa", d);
    return 0;
}
