// rk_count.hip -- pass 1 of the -M path (src/rkmh.cpp:904-910: every hash of every read increments HASHTCounter[h % slots])
// WITHOUT one global atomic per window.
//
// Why: 1.35e8 device-scope atomics per 1 M reads take 5.2 ms whatever the table size (2.6e10 atomics/s is the device's rate,
// DESIGN.md section 6).  Here the fused kernel's count form only WRITES each window's slot (rk_classify.hip, MODE 1 with a slot
// array), and the counting itself happens in LDS:
//   k_slot_hist     per span of the slot array: how many slots fall into each bin (a bin = R consecutive sub-ranges of 2^15 slots)
//   k_bin_prefix    per bin: exclusive prefix over the spans (+ k_bin_starts: prefix over the bins)   -> no cursor atomics,
//                   the layout of the binned copy is a pure function of the data
//   k_slot_scatter  per span: chunks of 16384 slots are counting-sorted by bin in LDS and appended to the bins in runs
//   k_count_bins    per (bin, sub-range): the bin's slots are read (L2), those of the sub-range counted in 128 KB of LDS
//                   counters, and the counters ADDED to the table with plain 16-byte read-modify-writes (a sub-range has one
//                   owner per launch; launches into one table are chained by an event in rk_api.hip)
// All streaming: ~0.6 GB written + read per stage at C2 instead of 1.35e8 atomics.
#include "rk_kernels.hpp"

#ifndef RK_CB_NT
#define RK_CB_NT 1
#endif

namespace rk {

typedef uint32_t u32x4c __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t cb_bin(uint32_t slot, const CountPlan& pl) {
    const uint32_t sub = slot >> CB_SUB_LG;
    return pl.R == 1 ? sub : __umulhi(sub, pl.magicR); // sub * R < 2^32: the magic division is exact
}

// exclusive prefix of one value per thread over a 1024-thread block; *total = the sum.  ws: 17 dwords of LDS
__device__ __forceinline__ uint32_t cb_block_scan(uint32_t v, uint32_t* ws, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) ws[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 16; ++i) { const uint32_t t = ws[i]; ws[i] = run; run += t; }
        ws[16] = run;
    }
    __syncthreads();
    *total = ws[16];
    return ws[w] + inc - v;
}

__global__ __launch_bounds__(CB_THREADS) void k_slot_hist(const uint32_t* __restrict__ flat, CountPlan pl, uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[CB_MAX_BINS];
    lh[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * pl.span;
    const uint64_t hi = lo + pl.span < pl.n ? lo + pl.span : pl.n; // n is a multiple of 4 and the array is sentinel-filled
    for (uint64_t i = lo + (uint64_t)threadIdx.x * 4; i < hi; i += (uint64_t)CB_THREADS * 4) {
        const u32x4c v = __builtin_nontemporal_load(reinterpret_cast<const u32x4c*>(flat + i));
        if (v.x != CB_NONE) atomicAdd(&lh[cb_bin(v.x, pl)], 1u);
        if (v.y != CB_NONE) atomicAdd(&lh[cb_bin(v.y, pl)], 1u);
        if (v.z != CB_NONE) atomicAdd(&lh[cb_bin(v.z, pl)], 1u);
        if (v.w != CB_NONE) atomicAdd(&lh[cb_bin(v.w, pl)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < pl.nb) hist[(size_t)blockIdx.x * pl.nb + threadIdx.x] = lh[threadIdx.x];
}

// off[g][b] = entries of bin b in the spans before g; total[b] = entries of bin b.  32 bins x 32 groups of spans per block
__global__ __launch_bounds__(1024) void k_bin_prefix(const uint32_t* __restrict__ hist, CountPlan pl, uint32_t* __restrict__ off,
                                                     uint32_t* __restrict__ total) {
    __shared__ uint32_t part[32][33];
    const uint32_t bi = threadIdx.x & 31, gg = threadIdx.x >> 5, b = blockIdx.x * 32 + bi;
    const uint32_t rpg = (pl.G + 31) / 32, g0 = gg * rpg, g1 = g0 + rpg < pl.G ? g0 + rpg : pl.G;
    uint32_t sum = 0;
    if (b < pl.nb) for (uint32_t g = g0; g < g1; ++g) sum += hist[(size_t)g * pl.nb + b];
    part[gg][bi] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (uint32_t j = 0; j < gg; ++j) run += part[j][bi];
    if (b < pl.nb) {
        for (uint32_t g = g0; g < g1; ++g) { off[(size_t)g * pl.nb + b] = run; run += hist[(size_t)g * pl.nb + b]; }
        if (gg == 31) total[b] = run;
    }
}

// bin_start[0 .. nb] = exclusive prefix of total[]
__global__ __launch_bounds__(CB_THREADS) void k_bin_starts(const uint32_t* __restrict__ total, CountPlan pl, uint32_t* __restrict__ bin_start) {
    __shared__ uint32_t ws[17];
    uint32_t sum;
    const uint32_t v = threadIdx.x < pl.nb ? total[threadIdx.x] : 0u;
    const uint32_t ex = cb_block_scan(v, ws, &sum);
    if (threadIdx.x < pl.nb) bin_start[threadIdx.x] = ex;
    if (threadIdx.x == 0) bin_start[pl.nb] = sum;
}

__global__ __launch_bounds__(CB_THREADS) void k_slot_scatter(const uint32_t* __restrict__ flat, CountPlan pl, const uint32_t* __restrict__ off,
                                                             const uint32_t* __restrict__ bin_start, uint32_t* __restrict__ bin_data) {
    extern __shared__ uint32_t sm[];
    uint32_t* buf = sm;                      // [CB_CHUNK]
    uint32_t* lh = buf + CB_CHUNK;           // [1024] entries of the chunk per bin
    uint32_t* lstart = lh + CB_MAX_BINS;     // [1024] their first index in buf
    uint32_t* cursor = lstart + CB_MAX_BINS; // [1024] next free entry of each bin for THIS span
    uint32_t* ws = cursor + CB_MAX_BINS;     // [17]
    const uint32_t tid = threadIdx.x;
    cursor[tid] = tid < pl.nb ? bin_start[tid] + off[(size_t)blockIdx.x * pl.nb + tid] : 0u;
    const uint64_t lo = (uint64_t)blockIdx.x * pl.span;
    const uint64_t hi = lo + pl.span < pl.n ? lo + pl.span : pl.n;
    for (uint64_t c0 = lo; c0 < hi; c0 += CB_CHUNK) {
        lh[tid] = 0;
        __syncthreads();
        constexpr int QPT = CB_CHUNK / (CB_THREADS * 4); // 16-byte loads per thread and chunk
        uint32_t v[4 * QPT], rk[4 * QPT];
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            const uint64_t i = c0 + ((uint64_t)j * CB_THREADS + tid) * 4;
            u32x4c q = {CB_NONE, CB_NONE, CB_NONE, CB_NONE};
            if (i < hi) q = __builtin_nontemporal_load(reinterpret_cast<const u32x4c*>(flat + i));
            v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
        }
#pragma unroll
        for (int e = 0; e < 4 * QPT; ++e) { rk[e] = 0; if (v[e] != CB_NONE) rk[e] = atomicAdd(&lh[cb_bin(v[e], pl)], 1u); }
        __syncthreads();
        uint32_t total;
        const uint32_t cnt = lh[tid];
        const uint32_t ex = cb_block_scan(cnt, ws, &total);
        lstart[tid] = ex;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4 * QPT; ++e) if (v[e] != CB_NONE) buf[lstart[cb_bin(v[e], pl)] + rk[e]] = v[e];
        __syncthreads();
        for (uint32_t i = tid; i < total; i += CB_THREADS) {
            const uint32_t x = buf[i];
            const uint32_t b = cb_bin(x, pl);
            bin_data[cursor[b] + (i - lstart[b])] = x;
        }
        __syncthreads();
        cursor[tid] += cnt;
    }
}

template <bool ALIGNED>
__global__ __launch_bounds__(CB_THREADS) void k_count_bins(const uint32_t* __restrict__ bin_data, const uint32_t* __restrict__ bin_start,
                                                           CountPlan pl, int32_t* __restrict__ counter) {
    extern __shared__ uint32_t cnt[]; // [2^15]
    // the R sub-ranges of a bin run on one XCD (blockIdx % 8, round-robin dispatch), close in time: the bin is read from HBM once
    const uint32_t grp = blockIdx.x / (8u * pl.R), within = blockIdx.x % (8u * pl.R);
    const uint32_t r = within >> 3, b = grp * 8u + (within & 7u);
    const uint32_t sub = b * pl.R + r;
    if (b >= pl.nb || sub >= pl.nsub) return;
    const uint32_t tid = threadIdx.x;
    uint4* c4 = reinterpret_cast<uint4*>(cnt);
    for (uint32_t j = tid; j < (1u << CB_SUB_LG) / 4; j += CB_THREADS) c4[j] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const uint32_t lo = bin_start[b], hi = bin_start[b + 1];
    for (uint32_t i = (lo & ~3u) + tid * 4; i < hi; i += CB_THREADS * 4) {
        const u32x4c v = *reinterpret_cast<const u32x4c*>(bin_data + i); // entries before lo belong to another bin: no match
        if ((v.x >> CB_SUB_LG) == sub) atomicAdd(&cnt[v.x & ((1u << CB_SUB_LG) - 1)], 1u);
        if ((v.y >> CB_SUB_LG) == sub && i + 1 < hi) atomicAdd(&cnt[v.y & ((1u << CB_SUB_LG) - 1)], 1u);
        if ((v.z >> CB_SUB_LG) == sub && i + 2 < hi) atomicAdd(&cnt[v.z & ((1u << CB_SUB_LG) - 1)], 1u);
        if ((v.w >> CB_SUB_LG) == sub && i + 3 < hi) atomicAdd(&cnt[v.w & ((1u << CB_SUB_LG) - 1)], 1u);
    }
    __syncthreads();
    const uint64_t base = (uint64_t)sub << CB_SUB_LG;
    for (uint32_t j = tid * 4; j < (1u << CB_SUB_LG); j += CB_THREADS * 4) {
        const uint4 c = c4[j >> 2];
        if (!(c.x | c.y | c.z | c.w)) continue;
        const uint64_t s = base + j;
        if (ALIGNED && s + 4 <= pl.slots) {
            typedef int i32x4c __attribute__((ext_vector_type(4)));
            i32x4c* p = reinterpret_cast<i32x4c*>(counter + s);
#if RK_CB_NT // the table streams past the L2 the bin's re-reads live in (0.77 -> 0.71 ms at 200 M slots)
            i32x4c t = __builtin_nontemporal_load(p);
#else
            i32x4c t = *p;
#endif
            t.x += (int)c.x; t.y += (int)c.y; t.z += (int)c.z; t.w += (int)c.w;
#if RK_CB_NT
            __builtin_nontemporal_store(t, p);
#else
            *p = t;
#endif
        } else {
            const uint32_t cc[4] = {c.x, c.y, c.z, c.w};
            for (int e = 0; e < 4; ++e) if (cc[e] && s + e < pl.slots) counter[s + e] += (int)cc[e];
        }
    }
}

bool count_plan(uint64_t slots, uint64_t n_entries, CountPlan* out) {
    if (slots == 0 || slots >= 0xFFFFFFFFull || n_entries == 0 || n_entries >= (1ull << 31)) return false;
    CountPlan pl{};
    pl.slots = slots;
    pl.nsub = (uint32_t)((slots + (1u << CB_SUB_LG) - 1) >> CB_SUB_LG);
    pl.R = (pl.nsub + CB_MAX_BINS - 1) / CB_MAX_BINS;
    pl.nb = (pl.nsub + pl.R - 1) / pl.R;
    pl.magicR = pl.R > 1 ? 0xFFFFFFFFu / pl.R + 1u : 0u;
    pl.n = (n_entries + 3) & ~3ull;
    uint64_t chunks = (pl.n + CB_CHUNK - 1) / CB_CHUNK;
    pl.G = (uint32_t)(chunks < 512 ? chunks : 512);
    pl.span = ((chunks + pl.G - 1) / pl.G) * CB_CHUNK;
    pl.G = (uint32_t)((pl.n + pl.span - 1) / pl.span);
    *out = pl;
    return true;
}

size_t count_plan_scratch_bytes(const CountPlan& pl) {
    // flat + binned copy (+ 16 bytes each: the last quad), hist + off [G][nb], total [nb], bin_start [nb + 1]
    return (pl.n * 4 + 64) * 2 + ((size_t)pl.G * pl.nb * 2 + 2 * (size_t)pl.nb + 8) * 4 + 256;
}

CountScratch count_plan_carve(const CountPlan& pl, void* ws) {
    CountScratch s;
    uint8_t* p = (uint8_t*)ws;
    s.flat = (uint32_t*)p; p += pl.n * 4 + 64;
    s.binned = (uint32_t*)p; p += pl.n * 4 + 64;
    s.hist = (uint32_t*)p; p += (size_t)pl.G * pl.nb * 4;
    s.off = (uint32_t*)p; p += (size_t)pl.G * pl.nb * 4;
    s.total = (uint32_t*)p; p += (size_t)pl.nb * 4;
    s.bin_start = (uint32_t*)p;
    return s;
}

hipError_t launch_count_prepare(const CountPlan& pl, const CountScratch& s, hipStream_t st) {
    return hipMemsetAsync(s.flat, 0xFF, pl.n * 4 + 64, st); // every entry = "no window here" until the fused kernel writes a slot
}

hipError_t launch_count_bins(const CountPlan& pl, const CountScratch& s, int32_t* counter, hipStream_t st) {
    hipLaunchKernelGGL(k_slot_hist, dim3(pl.G), dim3(CB_THREADS), 0, st, s.flat, pl, s.hist);
    hipLaunchKernelGGL(k_bin_prefix, dim3((pl.nb + 31) / 32), dim3(1024), 0, st, s.hist, pl, s.off, s.total);
    hipLaunchKernelGGL(k_bin_starts, dim3(1), dim3(CB_THREADS), 0, st, s.total, pl, s.bin_start);
    const size_t lds_a = ((size_t)CB_CHUNK + 3 * CB_MAX_BINS + 32) * 4;
    const size_t lds_b = (size_t)4 << CB_SUB_LG;
    const bool aligned = ((uintptr_t)counter & 15) == 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_slot_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(aligned ? reinterpret_cast<const void*>(k_count_bins<true>) : reinterpret_cast<const void*>(k_count_bins<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_slot_scatter, dim3(pl.G), dim3(CB_THREADS), lds_a, st, s.flat, pl, s.off, s.bin_start, s.binned);
    const uint32_t grid = ((pl.nb + 7) / 8) * 8 * pl.R;
    if (aligned)
        hipLaunchKernelGGL(k_count_bins<true>, dim3(grid), dim3(CB_THREADS), lds_b, st, s.binned, s.bin_start, pl, counter);
    else
        hipLaunchKernelGGL(k_count_bins<false>, dim3(grid), dim3(CB_THREADS), lds_b, st, s.binned, s.bin_start, pl, counter);
    return hipGetLastError();
}

} // namespace rk
