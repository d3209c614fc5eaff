// rk_classify.hip -- the fused per-read kernel of the classify/stream hot path (gfx950, wave64).
//
// Replaces the body of main_stream's read loop, /root/reference/src/rkmh.cpp:856-888 (and the two
// passes of the -M variant, :904-934):
//     to_upper -> calc_hashes -> [mask_by_frequency] -> minhashes -> R x hash_intersection_size -> argmax/diff
// for every read whose non-zero hashes all fit the sketch (n <= S: minhashes() keeps everything, so its
// sort cannot change any intersection size and is skipped).  Other reads are flagged max_id = -2 and
// rerouted by the host through k_hash_tiles + k_sort_intersect.
//
// Work decomposition (one 256-thread workgroup = one TILE of T consecutive reads):
//   phase 0  the tile's bases are ONE contiguous byte range of the batch: stage it with coalesced dword
//            loads into LDS as an upper-cased forward image, a reverse-complement image and a validity
//            bitmap; per-read byte/window prefix tables.
//   phase 1  windows of all T reads are flattened over the 256 threads (no lane idles on a short read):
//            canonical murmur3 of the window (two unaligned LDS window reads), one 16-byte bucket load
//            from the L2-resident reference index, hits appended to a per-read LDS list.
//   phase 2  16 lanes per read: LDS bitmap detects possibly repeated hits (then an exact multiset pass),
//            per-reference 16-bit LDS counters accumulate the postings, DPP row reductions give
//            max / first-max-index / best-earlier-score -> one int4 per read.
// Integer work only (no MFMA); the kernel is VALU bound by the 2 x MurmurHash3_x64_128 per window.
#include "rk_kernels.hpp"

#include <cstdlib>

namespace rk {

constexpr int TILE_THREADS = 256;
constexpr int GROUPS = TILE_THREADS / 16; // phase-2 lane groups
constexpr int BM_WORDS = 64;              // 2048-bit duplicate-detection bitmap per group

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
// reductions over a 16-lane DPP row; every lane of the row ends with the result
__device__ __forceinline__ uint32_t row_max_u32(uint32_t v) {
    uint32_t t;
    t = (uint32_t)dpp_i32<0xB1>((int)v); v = t > v ? t : v;   // quad_perm [1,0,3,2]
    t = (uint32_t)dpp_i32<0x4E>((int)v); v = t > v ? t : v;   // quad_perm [2,3,0,1]
    t = (uint32_t)dpp_i32<0x141>((int)v); v = t > v ? t : v;  // row_half_mirror
    t = (uint32_t)dpp_i32<0x140>((int)v); v = t > v ? t : v;  // row_mirror
    return v;
}
__device__ __forceinline__ int row_max_i32(int v) {
    int t;
    t = dpp_i32<0xB1>(v); v = t > v ? t : v;
    t = dpp_i32<0x4E>(v); v = t > v ? t : v;
    t = dpp_i32<0x141>(v); v = t > v ? t : v;
    t = dpp_i32<0x140>(v); v = t > v ? t : v;
    return v;
}
__device__ __forceinline__ uint32_t row_or_u32(uint32_t v) {
    v |= (uint32_t)dpp_i32<0xB1>((int)v);
    v |= (uint32_t)dpp_i32<0x4E>((int)v);
    v |= (uint32_t)dpp_i32<0x141>((int)v);
    v |= (uint32_t)dpp_i32<0x140>((int)v);
    return v;
}

struct TileGeom {
    int32_t T;            // reads per tile (<= 64)
    int32_t cap_bytes;    // staged bytes per tile
    int32_t cap_windows;  // hit-list capacity per tile (all k)
    int32_t cwords;       // 16-bit counter words per phase-2 group = (nref + 1) / 2
};

__host__ __device__ inline size_t tile_lds_bytes(const TileGeom& g) {
    return ((size_t)stage_lds_dwords(g.cap_bytes) + 4 * (size_t)(g.T + 1) + (size_t)g.cap_windows +
            (size_t)GROUPS * (size_t)(g.cwords + BM_WORDS) + 4) * 4;
}

// for every posting (ref, mult) of an index slot
template <typename F>
__device__ __forceinline__ void for_postings(const RefIndex& ix, uint32_t v, F f) {
    if (!(v >> 31)) f(v & 0xFFFFFu, (v >> 20) & 0x7FFu);
    else {
        const uint32_t off = v & 0x7fffffffu;
        const uint32_t cnt = ix.post[off];
        for (uint32_t c = 0; c < cnt; ++c) f(ix.post[off + 1 + 2 * c], ix.post[off + 2 + 2 * c]);
    }
}

__device__ __forceinline__ uint32_t bm_bit(uint32_t slot) { return (slot * 0x9E3779B1u) >> 21; } // 11 bits

template <int KT, int MODE>
__global__ __launch_bounds__(TILE_THREADS) void k_classify_tile(const uint8_t* __restrict__ bases, const uint32_t* __restrict__ offs,
                                                                uint32_t nreads, KsArr ks, int S, RefIndex ix, int32_t* counter,
                                                                uint64_t slots, int min_occ, int32_t* out4, DevPolicy pol,
                                                                TileGeom geo) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int T = geo.T;
    uint32_t* stage = smem;
    uint32_t* rstart = stage + stage_lds_dwords(geo.cap_bytes); // [T+1] byte offset of read t inside the tile
    uint32_t* wstart = rstart + (T + 1);                        // [T+1] first flattened window of read t
    uint32_t* nhit = wstart + (T + 1);                          // [T+1] hits per read
    uint32_t* nzero = nhit + (T + 1);                           // [T+1] zero hashes per read; [T] = tile-has-invalid flag
    uint32_t* hits = nzero + (T + 1);                           // [cap_windows] slots (phase 1) -> vals (phase 2)
    uint32_t* grp = hits + geo.cap_windows;                     // GROUPS x (cwords + BM_WORDS)
    const int tid = threadIdx.x;

    if (MODE == 0) { // counters and bitmaps stay zero between reads (phase 2 undoes what it sets)
        for (int i = tid; i < GROUPS * (geo.cwords + BM_WORDS); i += TILE_THREADS) grp[i] = 0;
    }
    const uint32_t ntiles = (nreads + (uint32_t)T - 1) / (uint32_t)T;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t r0 = tile * (uint32_t)T;
        const int Tn = (int)((nreads - r0) < (uint32_t)T ? (nreads - r0) : (uint32_t)T);
        __syncthreads(); // previous tile fully consumed
        // ---- phase 0: prefix tables (wave 0) ----------------------------------------------------
        const uint32_t tstart = offs[r0];
        if (tid < 64) {
            uint32_t o0 = 0, o1 = 0;
            if (tid < Tn) { o0 = offs[r0 + tid]; o1 = offs[r0 + tid + 1]; }
            const int len = (int)(o1 - o0);
            uint32_t nw = 0;
            if (tid < Tn) {
                if (KT) nw = (uint32_t)num_windows(len, KT, pol.drop_last_window);
                else for (int j = 0; j < ks.n; ++j) nw += (uint32_t)num_windows(len, ks.k[j], pol.drop_last_window);
            }
            uint32_t inc = nw;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                uint32_t v = (uint32_t)__shfl_up((int)inc, o);
                if (tid >= o) inc += v;
            }
            if (tid < Tn) { rstart[tid] = o0 - tstart; wstart[tid + 1] = inc; nhit[tid] = 0; nzero[tid] = 0; }
            if (tid == Tn - 1) rstart[Tn] = o1 - tstart;
            if (tid == 0) { wstart[0] = 0; nzero[T] = 0; }
        }
        __syncthreads();
        const uint32_t B = rstart[Tn];
        const uint32_t W = wstart[Tn];
        if (B > (uint32_t)geo.cap_bytes || W > (uint32_t)geo.cap_windows) { // a read longer than the hint: reroute the tile
            if (MODE == 0 && tid < Tn) reinterpret_cast<int4*>(out4)[r0 + tid] = make_int4(-2, 0, 0, 0);
            continue;
        }
        Staged s = stage_piece(bases, tstart, B, stage, geo.cap_bytes, tid, TILE_THREADS, [] { __syncthreads(); });
        // does any real base of the tile fail the ACGT test?  (lets phase 1 skip the per-window bit test)
        {
            uint32_t any = 0;
            const uint32_t lo_bit = s.fbase, hi_bit = s.fbase + B; // [lo,hi)
            for (uint32_t wd = tid; wd * 32 < hi_bit; wd += TILE_THREADS) {
                uint32_t m = s.inv[wd];
                const uint32_t b0 = wd * 32;
                if (b0 < lo_bit) m &= ~0u << (lo_bit - b0);
                if (b0 + 32 > hi_bit) m &= ~0u >> (b0 + 32 - hi_bit);
                any |= m;
            }
            if (any) nzero[T] = 1;
        }
        __syncthreads();
        const bool has_invalid = nzero[T] != 0;

        // ---- phase 1: flattened windows ----------------------------------------------------------
        {
            int t = 0;
            for (uint32_t w = tid; w < W; w += TILE_THREADS) {
                while (w >= wstart[t + 1]) ++t;
                uint32_t i = w - wstart[t];
                const uint32_t rs = rstart[t];
                int k = KT;
                if (!KT) {
                    const int len = (int)(rstart[t + 1] - rs);
                    for (int j = 0; j < ks.n; ++j) {
                        k = ks.k[j];
                        const uint32_t nwk = (uint32_t)num_windows(len, k, pol.drop_last_window);
                        if (i < nwk) break;
                        i -= nwk;
                    }
                }
                const uint32_t p = rs + i; // window start inside the tile
                uint64_t h;
                if (has_invalid && !window_valid<KT>(s, p, k)) h = 0;
                else {
                    const uint64_t f = murmur_window<KT>(s.fwd, s.fbase + p, k, pol.seed, pol.fold);
                    const uint64_t r = murmur_window<KT>(s.rc, B - (uint32_t)k - p, k, pol.seed, pol.fold);
                    h = f < r ? f : r;
                }
                if (MODE == 1) {
                    if (pol.counter_counts_zero || h != 0) atomicAdd(&counter[h % slots], 1);
                    continue;
                }
                if (counter) { // mask_by_frequency, rkmh.cpp:916
                    const int c = counter[h % slots];
                    if (pol.mask_strict_less ? (c < min_occ) : (c <= min_occ)) h = 0;
                }
                if (h == 0) { atomicAdd(&nzero[t], 1u); continue; }
                const uint32_t slot = index_find(ix, h);
                if (slot != IDX_NOT_FOUND) {
                    const uint32_t j = atomicAdd(&nhit[t], 1u);
                    hits[wstart[t] + j] = slot;
                }
            }
        }
        if (MODE == 1) continue;
        __syncthreads();

        // ---- phase 2: 16 lanes per read -----------------------------------------------------------
        {
            const int g = tid >> 4, sl = tid & 15;
            uint32_t* c16 = grp + g * (geo.cwords + BM_WORDS);
            uint32_t* bm = c16 + geo.cwords;
            for (int t = g; t < Tn; t += GROUPS) {
                const uint32_t H = nhit[t];
                uint32_t* hl = hits + wstart[t];
                const int nmins = (int)(wstart[t + 1] - wstart[t]) - (int)nzero[t];
                if (nmins > S) { // bottom-S selection matters: general path
                    if (sl == 0) reinterpret_cast<int4*>(out4)[r0 + t] = make_int4(-2, 0, 0, 0);
                    continue;
                }
                // A0: possibly repeated hits?  (same slot twice => multiset semantics need occurrence ranks)
                uint32_t dup = 0;
                for (uint32_t j = sl; j < H; j += 16) {
                    const uint32_t bit = bm_bit(hl[j]);
                    const uint32_t old = atomicOr(&bm[bit >> 5], 1u << (bit & 31));
                    dup |= (old >> (bit & 31)) & 1u;
                }
                dup = row_or_u32(dup);
                for (uint32_t j = sl; j < H; j += 16) bm[bm_bit(hl[j]) >> 5] = 0; // undo
                // A: accumulate postings into the per-reference counters
                if (!dup) {
                    for (uint32_t j = sl; j < H; j += 16) {
                        const uint32_t v = ix.vals[hl[j]];
                        hl[j] = v; // later passes only need the postings
                        for_postings(ix, v, [&](uint32_t ref, uint32_t) { atomicAdd(&c16[ref >> 1], 1u << ((ref & 1) * 16)); });
                    }
                } else {
                    // exact: occurrence rank among equal slots, contribution iff rank < multiplicity in the reference
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                    for (uint32_t j = sl; j < H; j += 16) {
                        const uint32_t slot = hl[j];
                        uint32_t rank = 0;
                        for (uint32_t u = 0; u < j; ++u) rank += (hl[u] == slot) ? 1u : 0u;
                        const uint32_t v = ix.vals[slot];
                        for_postings(ix, v, [&](uint32_t ref, uint32_t mult) {
                            if (rank < mult) atomicAdd(&c16[ref >> 1], 1u << ((ref & 1) * 16));
                        });
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                    for (uint32_t j = sl; j < H; j += 16) hl[j] = ix.vals[hl[j]];
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                // B: best candidate = max count, then smallest reference index (first max wins, rkmh.cpp:878)
                uint32_t best = 0; // (count << 16) | (0xFFFF - ref); 0 = none
                for (uint32_t j = sl; j < H; j += 16) {
                    for_postings(ix, hl[j], [&](uint32_t ref, uint32_t) {
                        const uint32_t c = (c16[ref >> 1] >> ((ref & 1) * 16)) & 0xFFFFu;
                        const uint32_t key = (c << 16) | (0xFFFFu - ref);
                        best = (c != 0 && key > best) ? key : best;
                    });
                }
                best = row_max_u32(best);
                const int max_id = best ? (int)(0xFFFFu - (best & 0xFFFFu)) : 0;
                const int max_shared = (int)(best >> 16);
                // C: best score among EARLIER references (untouched ones count 0; none => -1)
                int prev = max_id > 0 ? 0 : -1;
                for (uint32_t j = sl; j < H; j += 16) {
                    for_postings(ix, hl[j], [&](uint32_t ref, uint32_t) {
                        const int c = (int)((c16[ref >> 1] >> ((ref & 1) * 16)) & 0xFFFFu);
                        if ((int)ref < max_id && c > prev) prev = c;
                    });
                }
                prev = row_max_i32(prev);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                // D: undo the counters
                for (uint32_t j = sl; j < H; j += 16)
                    for_postings(ix, hl[j], [&](uint32_t ref, uint32_t) { c16[ref >> 1] = 0; });
                if (sl == 0) reinterpret_cast<int4*>(out4)[r0 + t] = make_int4(max_id, max_shared, max_shared - prev, nmins);
            }
        }
    }
}

static TileGeom make_geom(int maxlen, const KsArr& ks, int nref, const DevPolicy& pol) {
    TileGeom g;
    if (maxlen < 1) maxlen = 1;
    int T = 4800 / maxlen;
    if (const char* e = getenv("RKMH_TILE_T")) T = atoi(e);
    if (T > 32) T = 32;
    if (T < 4) T = 4;
    g.T = T;
    g.cap_bytes = T * maxlen;
    int wmax = 0; // most windows one read can have (all k)
    for (int j = 0; j < ks.n; ++j) wmax += num_windows(maxlen, ks.k[j], pol.drop_last_window);
    g.cap_windows = T * (wmax > 0 ? wmax : 1);
    g.cwords = (nref + 1) / 2;
    return g;
}

// 16 phase-2 groups x nref 16-bit counters must fit beside the tile in LDS
bool classify_tile_supported(int nref) { return nref <= 2048; }

hipError_t launch_classify_tile(const uint8_t* bases, const uint32_t* offs, uint32_t nreads, const KsArr& ks, int S,
                                const RefIndex& ix, int32_t* counter, uint64_t slots, int min_occ, int mode,
                                int32_t* out4, const DevPolicy& pol, int maxlen, hipStream_t st) {
    if (nreads == 0) return hipSuccess;
    const TileGeom geo = make_geom(maxlen, ks, mode == 0 ? ix.nref : 0, pol);
    const size_t lds = tile_lds_bytes(geo);
    const uint32_t ntiles = (nreads + (uint32_t)geo.T - 1) / (uint32_t)geo.T;
    uint32_t grid = ntiles;
    if (const char* g = getenv("RKMH_TILE_GRID")) { uint32_t v = (uint32_t)atoi(g); if (v && v < grid) grid = v; }
    const bool k16 = (ks.n == 1 && ks.k[0] == 16);
#define RK_LAUNCH(KT, MODE)                                                                                          \
    do {                                                                                                             \
        if (lds > 64 * 1024) {                                                                                       \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_classify_tile<KT, MODE>),             \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
            if (e != hipSuccess) return e;                                                                           \
        }                                                                                                            \
        hipLaunchKernelGGL((k_classify_tile<KT, MODE>), dim3(grid), dim3(TILE_THREADS), lds, st, bases, offs, nreads, \
                           ks, S, ix, counter, slots, min_occ, out4, pol, geo);                                      \
    } while (0)
    if (mode == 0) { if (k16) RK_LAUNCH(16, 0); else RK_LAUNCH(0, 0); }
    else           { if (k16) RK_LAUNCH(16, 1); else RK_LAUNCH(0, 1); }
#undef RK_LAUNCH
    return hipGetLastError();
}

} // namespace rk
