// rk_call.hip -- kernels of the `call` sub-command (main_call, /root/reference/src/rkmh.cpp:1455-1904).
//
//   k_depth_insert   read_hash_to_depth[h] += 1 for every read hash      rkmh.cpp:1613-1622  (exact u64 -> count map:
//                    open-addressing table in HBM, atomicCAS on the key, atomicAdd on the count; hash 0 -- the
//                    invalid-k-mer sentinel -- is a key like any other, kept in a separate word)
//   k_depth_lookup   depth of every reference window                      rkmh.cpp:1785
//   k_scan_*         prefix sums of the depths (the reference's sliding window mean, rkmh.cpp:1786-1791, is a
//                    difference of two prefix sums; the window is NOT reset between references, Appendix C.9)
//   k_call_enumerate at every position whose depth is below half the window mean: the 3k SNP k-mers and the k
//                    one-base-deletion k-mers are hashed (canonical murmur3) and looked up; candidates that
//                    pass the depth tests of rkmh.cpp:1814 / :1853 are appended as records   rkmh.cpp:1801-1865
// One workgroup (one wave) per reference position; lanes = the 4k candidate k-mers.
#include "rk_kernels.hpp"

namespace rk {

__device__ __forceinline__ uint64_t dt_slot(uint64_t h, uint64_t mask) { return ((h * 0x9E3779B97F4A7C15ull) >> 17) & mask; }

__global__ __launch_bounds__(256) void k_depth_insert(const uint64_t* __restrict__ h, uint64_t n, unsigned long long* keys,
                                                      uint32_t* counts, uint64_t mask, uint32_t* zero_count) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t v = h[i];
        if (v == 0) { atomicAdd(zero_count, 1u); continue; }
        uint64_t s = dt_slot(v, mask);
        for (;;) {
            const unsigned long long old = atomicCAS(&keys[s], 0ull, (unsigned long long)v);
            if (old == 0ull || old == v) { atomicAdd(&counts[s], 1u); break; }
            s = (s + 1) & mask;
        }
    }
}
__device__ __forceinline__ int depth_of(const DepthTable& t, uint64_t h) {
    if (h == 0) return (int)*t.zero_count;
    uint64_t s = dt_slot(h, t.mask);
    for (;;) {
        const uint64_t key = t.keys[s];
        if (key == h) return (int)t.counts[s];
        if (key == 0) return 0;
        s = (s + 1) & t.mask;
    }
}
__global__ __launch_bounds__(256) void k_depth_lookup(const uint64_t* __restrict__ h, uint64_t n, DepthTable t, int32_t* depth) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) depth[i] = depth_of(t, h[i]);
}

hipError_t launch_depth_insert(const uint64_t* h, uint64_t n, const DepthTable& t, hipStream_t st) {
    if (n == 0) return hipSuccess;
    uint32_t grid = (uint32_t)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_depth_insert, dim3(grid), dim3(256), 0, st, h, n, reinterpret_cast<unsigned long long*>(t.keys), t.counts,
                       t.mask, t.zero_count);
    return hipGetLastError();
}
hipError_t launch_depth_lookup(const uint64_t* h, uint64_t n, const DepthTable& t, int32_t* depth, hipStream_t st) {
    if (n == 0) return hipSuccess;
    uint32_t grid = (uint32_t)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_depth_lookup, dim3(grid), dim3(256), 0, st, h, n, t, depth);
    return hipGetLastError();
}

// ---- exclusive prefix sums of int32 -> int64, 1024 elements per block, up to three levels --------
__global__ __launch_bounds__(256) void k_scan_block(const int32_t* __restrict__ in, uint64_t n, int64_t* out, int64_t* totals) {
    __shared__ int64_t part[256];
    const uint64_t base = (uint64_t)blockIdx.x * 1024;
    int64_t v[4], s = 0;
    for (int j = 0; j < 4; ++j) { uint64_t i = base + threadIdx.x * 4 + j; v[j] = i < n ? in[i] : 0; s += v[j]; }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        int64_t t = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    int64_t run = part[threadIdx.x] - s;
    for (int j = 0; j < 4; ++j) { uint64_t i = base + threadIdx.x * 4 + j; if (i < n) out[i] = run; run += v[j]; }
    if (threadIdx.x == 255) totals[blockIdx.x] = part[255];
}
__global__ __launch_bounds__(256) void k_scan_block64(const int64_t* __restrict__ in, uint64_t n, int64_t* out, int64_t* totals) {
    __shared__ int64_t part[256];
    const uint64_t base = (uint64_t)blockIdx.x * 1024;
    int64_t v[4], s = 0;
    for (int j = 0; j < 4; ++j) { uint64_t i = base + threadIdx.x * 4 + j; v[j] = i < n ? in[i] : 0; s += v[j]; }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        int64_t t = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    int64_t run = part[threadIdx.x] - s;
    for (int j = 0; j < 4; ++j) { uint64_t i = base + threadIdx.x * 4 + j; if (i < n) out[i] = run; run += v[j]; }
    if (threadIdx.x == 255) totals[blockIdx.x] = part[255];
}
__global__ __launch_bounds__(256) void k_scan_add(int64_t* out, uint64_t n, const int64_t* __restrict__ block_off) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] += block_off[i >> 10];
}
// out[i] = sum of in[0..i) ; scratch needs (n/1024 + 1) * 2 + 2050 int64
hipError_t launch_exclusive_scan(const int32_t* in, uint64_t n, int64_t* out, int64_t* scratch, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const uint64_t nb1 = (n + 1023) / 1024;
    int64_t* t1 = scratch;               // [nb1] block totals
    int64_t* t1s = scratch + nb1;        // [nb1] their exclusive scan
    int64_t* t2 = t1s + nb1;             // [nb2]
    hipLaunchKernelGGL(k_scan_block, dim3((uint32_t)nb1), dim3(256), 0, st, in, n, out, t1);
    if (nb1 > 1) {
        const uint64_t nb2 = (nb1 + 1023) / 1024;
        if (nb2 > 1024) return hipErrorInvalidValue; // > 2^30 positions
        int64_t* t2s = t2 + nb2;
        int64_t* t3 = t2s + nb2;
        hipLaunchKernelGGL(k_scan_block64, dim3((uint32_t)nb2), dim3(256), 0, st, (const int64_t*)t1, nb1, t1s, t2);
        if (nb2 > 1) {
            hipLaunchKernelGGL(k_scan_block64, dim3(1), dim3(256), 0, st, (const int64_t*)t2, nb2, t2s, t3);
            hipLaunchKernelGGL(k_scan_add, dim3((uint32_t)((nb1 + 255) / 256)), dim3(256), 0, st, t1s, nb1, (const int64_t*)t2s);
        }
        hipLaunchKernelGGL(k_scan_add, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, out, n, (const int64_t*)t1s);
    }
    return hipGetLastError();
}

// SNP alternatives in the order of rkmh.cpp:1634-1637: A->C,T,G  C->T,G,A  T->C,G,A  G->A,C,T (none for other bases)
__device__ __forceinline__ uint8_t snp_alt(uint8_t orig, int j) {
    switch (orig) {
        case 'A': return (uint8_t)("CTG"[j]);
        case 'C': return (uint8_t)("TGA"[j]);
        case 'T': return (uint8_t)("CGA"[j]);
        case 'G': return (uint8_t)("ACT"[j]);
        default: return 0;
    }
}

// One wave per reference window g (global index over all references, in order).
__global__ __launch_bounds__(64) void k_call_enumerate(const uint8_t* __restrict__ ref_upper, const uint64_t* __restrict__ ref_off,
                                                       const uint64_t* __restrict__ win_off, int nref, uint64_t nwin_total,
                                                       const int32_t* __restrict__ depth, const int64_t* __restrict__ prefix,
                                                       int k, int window_len, DepthTable t, DevPolicy pol, CallRecord* out,
                                                       uint32_t* out_count, uint32_t out_cap) {
    const int lane = threadIdx.x;
    for (uint64_t g = blockIdx.x; g < nwin_total; g += gridDim.x) {
        const int d = depth[g];
        const uint64_t cnt = g + 1 < (uint64_t)window_len ? g + 1 : (uint64_t)window_len;   // window never resets between references
        const int64_t sum = prefix[g] + d - prefix[g + 1 - cnt];
        const int avg_d = (int)(sum / (int64_t)cnt);                                         // (int)(double mean), rkmh.cpp:1791
        if (!((double)d < 0.5 * (double)avg_d)) continue;                                    // rkmh.cpp:1801
        int lo = 0, hi = nref - 1;                                                           // reference holding window g
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (win_off[mid] <= g) lo = mid; else hi = mid - 1; }
        const int ri = lo;
        const uint64_t j = g - win_off[ri];
        const uint8_t* seq = ref_upper + ref_off[ri];
        // candidates: c in [0,3k): SNP at alt_pos = c/3 with alternative c%3; c in [3k,4k): deletion of d_alt[alt_pos], alt_pos = c-3k+1
        for (int c = lane; c < 4 * k; c += 64) {
            uint8_t orig, alt;
            int alt_pos, kind;
            uint64_t h;
            if (c < 3 * k) {
                kind = 0; alt_pos = c / 3;
                orig = seq[j + alt_pos];
                alt = snp_alt(orig, c % 3);
                if (alt == 0) continue;                                                       // rotate_snps has no answer: no alternatives
                h = canonical_bytes([&](int q) -> uint8_t { return q == alt_pos ? alt : seq[j + q]; }, k, pol);
            } else {
                if (j == 0) continue;                                                         // d_alt is empty at the first position
                kind = 1; alt_pos = c - 3 * k + 1;
                orig = seq[j - 1 + alt_pos];
                alt = '-';
                h = canonical_bytes([&](int q) -> uint8_t { return seq[j - 1 + q + (q >= alt_pos ? 1 : 0)]; }, k, pol);
            }
            const int alt_depth = depth_of(t, h);
            bool call;
            if (kind == 0) call = ((double)alt_depth >= 0.1 * (double)avg_d) && (alt_depth > d);      // rkmh.cpp:1814
            else call = (double)alt_depth > 0.9 * (double)avg_d;                                    // rkmh.cpp:1853
            if (!call) continue;
            const uint32_t o = atomicAdd(out_count, 1u);
            if (o < out_cap) {
                CallRecord r;
                r.ref = ri; r.pos = (int32_t)(j + (uint64_t)alt_pos + 1); r.alt_depth = alt_depth; r.avg_d = avg_d; r.depth = d;
                r.orig = orig; r.alt = alt; r.kind = (uint8_t)kind; r.pad = 0;
                out[o] = r;
            }
        }
    }
}

hipError_t launch_call_enumerate(const uint8_t* ref_upper, const uint64_t* ref_off, const uint64_t* win_off, int nref,
                                 uint64_t nwin_total, const int32_t* depth, const int64_t* prefix, int k, int window_len,
                                 const DepthTable& t, const DevPolicy& pol, CallRecord* out, uint32_t* out_count, uint32_t out_cap,
                                 hipStream_t st) {
    if (nwin_total == 0) return hipSuccess;
    uint32_t grid = (uint32_t)(nwin_total < 65536 ? nwin_total : 65536);
    hipLaunchKernelGGL(k_call_enumerate, dim3(grid), dim3(64), 0, st, ref_upper, ref_off, win_off, nref, nwin_total, depth, prefix, k,
                       window_len, t, pol, out, out_count, out_cap);
    return hipGetLastError();
}

} // namespace rk

// ---- the C ABI of `call` (include/rkmh_amd.h) ----
#include "rk_api_internal.hpp"

// ---- call ------------------------------------------------------------------------------------------
extern "C" int rk_call(rk_ctx* c, const uint8_t* ref_bases, const uint64_t* ref_offsets, int nref,
                       const uint8_t* read_bases, const uint64_t* read_offsets, int64_t nreads, int k, int window_len,
                       rk_call_record** out, int64_t* nout) {
    static_assert(sizeof(rk_call_record) == sizeof(CallRecord), "record layouts must match");
    if (!c || !ref_offsets || !read_offsets || nref < 1 || nreads < 0 || !out || !nout) return fail(RK_ERR_ARG, "bad arguments");
    if (k < 1 || k > RK_MAX_K) return fail(RK_ERR_LIMIT, "k=%d outside [1,%d]", k, RK_MAX_K);
    if (window_len < 1) return fail(RK_ERR_ARG, "window length must be positive");
    RKCHK(set_dev(c));
    *out = nullptr; *nout = 0;
    GeneralCfg cfg; cfg.ks.n = 1; cfg.ks.k[0] = k;
    // windows of the reads / of the references
    uint64_t wr = 0;
    for (int64_t i = 0; i < nreads; ++i) wr += (uint64_t)num_windows((int)(read_offsets[i + 1] - read_offsets[i]), k, c->pol.drop_last_window);
    std::vector<uint64_t> win_off((size_t)nref + 1, 0);
    for (int i = 0; i < nref; ++i)
        win_off[(size_t)i + 1] = win_off[(size_t)i] + (uint64_t)num_windows((int)(ref_offsets[i + 1] - ref_offsets[i]), k, c->pol.drop_last_window);
    const uint64_t wtot = win_off[(size_t)nref];
    if (wtot >= (1ull << 30)) return fail(RK_ERR_LIMIT, "more than 2^30 reference positions");
    // exact depth map
    uint64_t cap = 1024;
    while (cap < 2 * wr) cap <<= 1;
    DevBuf d_keys, d_counts, d_depth, d_prefix, d_scratch, d_ref, d_refoff, d_winoff, d_rec, d_cnt;
    int rc = RK_OK;
    auto cleanup = [&]() { for (DevBuf* b : {&d_keys, &d_counts, &d_depth, &d_prefix, &d_scratch, &d_ref, &d_refoff, &d_winoff, &d_rec, &d_cnt}) b->release(); };
#define CALLCHK(expr) do { rc = (expr); if (rc != RK_OK) { cleanup(); return rc; } } while (0)
#define CALLHIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return fail(RK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); } } while (0)
    CALLCHK(d_keys.reserve(cap * 8));
    CALLCHK(d_counts.reserve(cap * 4 + 16));
    CALLHIP(hipMemsetAsync(d_keys.p, 0, cap * 8, c->st));
    CALLHIP(hipMemsetAsync(d_counts.p, 0, cap * 4 + 16, c->st));
    DepthTable t;
    t.keys = d_keys.as<uint64_t>(); t.counts = d_counts.as<uint32_t>(); t.zero_count = d_counts.as<uint32_t>() + cap; t.mask = cap - 1;
    {   // pass over the reads: read_hash_to_depth[h] += 1 (rkmh.cpp:1613-1622)
        GeneralCfg c1 = cfg; c1.depth_insert = &t;
        GeneralOut none;
        CALLCHK(general_run(c, read_bases, nullptr, read_offsets, nreads, c1, none));
    }
    // references: upper-cased copy stays on the device for the candidate enumeration
    const uint64_t rbytes = ref_offsets[nref];
    CALLCHK(d_ref.reserve(rbytes + 64));
    CALLCHK(d_refoff.reserve(((size_t)nref + 1) * 8));
    CALLCHK(d_winoff.reserve(((size_t)nref + 1) * 8));
    CALLCHK(d_depth.reserve((wtot + 1) * 4));
    CALLCHK(d_prefix.reserve((wtot + 2) * 8));
    CALLCHK(d_scratch.reserve(((wtot / 1024 + 2) * 2 + 4200) * 8));
    if (rbytes) CALLHIP(hipMemcpyAsync(d_ref.p, ref_bases, rbytes, hipMemcpyHostToDevice, c->st));
    CALLHIP(launch_to_upper(d_ref.as<uint8_t>(), rbytes, c->st));
    CALLHIP(hipMemcpyAsync(d_refoff.p, ref_offsets, ((size_t)nref + 1) * 8, hipMemcpyHostToDevice, c->st));
    CALLHIP(hipMemcpyAsync(d_winoff.p, win_off.data(), ((size_t)nref + 1) * 8, hipMemcpyHostToDevice, c->st));
    {   // depth of every reference window, in reference order (rkmh.cpp:1785)
        GeneralCfg c2 = cfg; c2.depth_lookup = &t; c2.depth_out = d_depth.as<int32_t>();
        GeneralOut none;
        CALLCHK(general_run(c, ref_bases, nullptr, ref_offsets, nref, c2, none));
    }
    CALLHIP(hipMemsetAsync(d_prefix.p, 0, (wtot + 2) * 8, c->st));
    CALLHIP(launch_exclusive_scan(d_depth.as<int32_t>(), wtot, d_prefix.as<int64_t>(), d_scratch.as<int64_t>(), c->st));
    uint32_t rcap = 1u << 16;
    CALLCHK(d_cnt.reserve(16));
    std::vector<rk_call_record> recs;
    for (;;) {
        CALLCHK(d_rec.reserve((size_t)rcap * sizeof(CallRecord)));
        CALLHIP(hipMemsetAsync(d_cnt.p, 0, 16, c->st));
        CALLHIP(launch_call_enumerate(d_ref.as<uint8_t>(), d_refoff.as<uint64_t>(), d_winoff.as<uint64_t>(), nref, wtot,
                                      d_depth.as<int32_t>(), d_prefix.as<int64_t>(), k, window_len, t, c->pol, d_rec.as<CallRecord>(),
                                      d_cnt.as<uint32_t>(), rcap, c->st));
        uint32_t n = 0;
        CALLHIP(hipMemcpyAsync(&n, d_cnt.p, 4, hipMemcpyDeviceToHost, c->st));
        CALLHIP(hipStreamSynchronize(c->st));
        if (n > rcap) { rcap = n + 1024; continue; } // rare: more calls than expected, run again with room for all
        recs.resize(n);
        if (n) CALLHIP(hipMemcpy(recs.data(), d_rec.p, (size_t)n * sizeof(CallRecord), hipMemcpyDeviceToHost));
        break;
    }
    cleanup();
#undef CALLCHK
#undef CALLHIP
    rk_call_record* r = (rk_call_record*)malloc(sizeof(rk_call_record) * (recs.empty() ? 1 : recs.size()));
    if (!r) return fail(RK_ERR_NOMEM, "malloc");
    if (!recs.empty()) memcpy(r, recs.data(), recs.size() * sizeof(rk_call_record));
    *out = r; *nout = (int64_t)recs.size();
    return RK_OK;
}


