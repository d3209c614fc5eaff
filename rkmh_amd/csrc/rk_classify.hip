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
//            bitmap; per-read tables; a bitmap of the byte positions that start no hashable window
//            (read tails, windows holding a non-ACGT base).
//   phase 1  the tile's byte positions are flattened over the 256 threads; the hot loop is branch-free:
//            two unaligned LDS window reads, both murmur3 chains in one basic block, one 16-byte bucket
//            load from the L2-resident reference index whose latency hides behind the NEXT position's
//            hashing.  Fingerprint matches (about 1 window in 8) are pushed to an LDS queue with one
//            wave-aggregated atomic.
//   phase 1b the queue is drained with every lane busy: full-key verification, postings accumulated into
//            per-read 16-bit LDS counters; since counts only grow, an LDS atomicMax of (count, -ref) per
//            increment leaves (max_shared, first max_id) behind without any scan.  An exact per-read
//            LDS hash set flags reads that hit the same sketch hash twice (multiset semantics).
//   phase 2  16 lanes per read: best earlier score for `diff` (DPP row reduction over the counters),
//            the exact multiset recount for flagged reads, one int4 per read, counters re-zeroed.
// Integer work only (no MFMA).
#include "rk_kernels.hpp"

#include <cstdlib>

namespace rk {

constexpr int TILE_THREADS = 256;
constexpr int GROUPS = TILE_THREADS / 16; // phase-2 lane groups
constexpr int DSET = 128;                 // per-read exact hit multiset (slot + occurrence count), open addressing

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

struct TileGeom {
    int32_t T;            // reads per tile (<= 64)
    int32_t cap_bytes;    // staged bytes per tile
    int32_t qcap;         // candidate queue entries
    int32_t cwords;       // 16-bit counter words per read = (nref + 1) / 2
    int32_t dbg;          // ablation switches for profiling (RKMH_DBG; 0 in production)
};

__host__ __device__ inline int tile_map_words(int cap_bytes) { return cap_bytes / 32 + 2; }
__host__ __device__ inline size_t tile_lds_bytes(const TileGeom& g) {
    return ((size_t)stage_lds_dwords(g.cap_bytes) + 6 * (size_t)(g.T + 1) + 2 * (size_t)tile_map_words(g.cap_bytes) +
            4 * (size_t)g.qcap + (size_t)g.T * (size_t)(g.cwords + DSET) + 8) * 4;
}

// for every posting (ref, mult) of an index value
template <typename F>
__device__ __forceinline__ void for_postings(const RefIndex& ix, uint32_t v, F f) {
    if (!(v >> 31)) f(v & 0xFFFFFu, (v >> 20) & 0x7FFu);
    else {
        const uint32_t off = v & 0x7fffffffu;
        const uint32_t cnt = ix.post[off];
        for (uint32_t c = 0; c < cnt; ++c) f(ix.post[off + 1 + 2 * c], ix.post[off + 2 + 2 * c]);
    }
}

// full lookup: fingerprint scan, then key and value of a matching slot are fetched together
__device__ __forceinline__ bool index_lookup(const RefIndex& ix, uint64_t h, uint32_t& slot, uint32_t& val) {
    const uint32_t fp = index_fp(h);
    uint32_t b = index_bucket(h, ix.bshift);
    for (;;) {
        const uint4 f = ix.fpb[b];
        uint32_t m = (f.x == fp ? 1u : 0u) | (f.y == fp ? 2u : 0u) | (f.z == fp ? 4u : 0u) | (f.w == fp ? 8u : 0u);
        while (m) {
            const uint32_t q = (uint32_t)__ffs((int)m) - 1u;
            m &= m - 1u;
            const uint32_t s = 4 * b + q;
            const uint64_t key = ix.keys[s];
            const uint32_t v = ix.vals[s];
            if (key == h) { slot = s; val = v; return true; }
        }
        if (f.w == 0) return false;
        b = (b + 1) & ix.bmask;
    }
}

template <int KT, int MODE, int FOLD>
__global__ __launch_bounds__(TILE_THREADS) void k_classify_tile(const uint8_t* __restrict__ bases, const uint32_t* __restrict__ offs,
                                                                uint32_t nreads, KsArr ks, int S, RefIndex ix, int32_t* counter,
                                                                uint64_t slots, int min_occ, int32_t* out4, DevPolicy pol,
                                                                TileGeom geo) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int T = geo.T;
    const int QCAP = geo.qcap;
    uint32_t* stage = smem;
    uint32_t* qh32 = stage + ((stage_lds_dwords(geo.cap_bytes) + 1) & ~1); // [2*QCAP] candidate hashes (8-byte aligned)
    uint64_t* qh = reinterpret_cast<uint64_t*>(qh32);
    uint32_t* qp = qh32 + 2 * QCAP;                              // [QCAP] tile byte position of the candidate window
    uint32_t* qs = qp + QCAP;                                    // [QCAP] slot whose fingerprint matched (or NONE)
    uint32_t* rstart = qs + QCAP;                                // [T+1] byte offset of read t inside the tile
    uint32_t* nwin = rstart + (T + 1);                           // [T+1] windows of read t (all k)
    uint32_t* nzero = nwin + (T + 1);                            // [T+1] zero hashes per read
    uint32_t* best = nzero + (T + 1);                            // [T+1] max over increments of (count << 16 | 0xFFFF - ref)
    uint32_t* flags = best + (T + 1);                            // [T+1] read needs the exact multiset recount
    uint32_t* misc = flags + (T + 1);                            // [0..3] per-wave queue lengths, [4] tile has invalid bases, [5] queue overflowed
    uint32_t* bad = misc + 8;                                     // bit p set <=> no hashable window starts at tile byte p
    uint32_t* tmap = bad + tile_map_words(geo.cap_bytes);        // read index holding tile byte 32*c
    uint32_t* c16 = tmap + tile_map_words(geo.cap_bytes);        // [T][cwords] packed 16-bit per-reference counters
    uint32_t* dset = c16 + T * geo.cwords;                       // [T][DSET] slots already hit by the read (+1; 0 = empty)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const int wave = tid >> 6;
    const uint32_t QW = (uint32_t)QCAP / 4u; // every wave owns a quarter of the queue: pushes need no atomics

    if (MODE == 0) // counters are re-zeroed by phase 2 after use
        for (int i = tid; i < T * geo.cwords; i += TILE_THREADS) c16[i] = 0;
    const uint32_t ntiles = (nreads + (uint32_t)T - 1) / (uint32_t)T;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t r0 = tile * (uint32_t)T;
        const int Tn = (int)((nreads - r0) < (uint32_t)T ? (nreads - r0) : (uint32_t)T);
        __syncthreads(); // previous tile fully consumed
        // ---- phase 0 -------------------------------------------------------------------------------
        const uint32_t tstart = offs[r0];
        if (tid <= Tn) {
            const uint32_t o0 = offs[r0 + tid];
            rstart[tid] = o0 - tstart;
            if (tid < Tn) {
                const int len = (int)(offs[r0 + tid + 1] - o0);
                uint32_t nw = 0;
                if (KT) nw = (uint32_t)num_windows(len, KT, pol.drop_last_window);
                else for (int j = 0; j < ks.n; ++j) nw += (uint32_t)num_windows(len, ks.k[j], pol.drop_last_window);
                nwin[tid] = nw; nzero[tid] = 0; best[tid] = 0; flags[tid] = 0;
            }
        }
        if (tid < 8) misc[tid] = 0;
        if (MODE == 0)
            for (int i = tid; i < Tn * DSET; i += TILE_THREADS) dset[i] = 0;
        __syncthreads();
        const uint32_t B = rstart[Tn];
        if (B > (uint32_t)geo.cap_bytes) { // a read longer than the hint: the host reroutes the tile
            if (MODE == 0 && tid < Tn) reinterpret_cast<int4*>(out4)[r0 + tid] = make_int4(-2, 0, 0, 0);
            continue;
        }
        Staged s = stage_piece(bases, tstart, B, stage, geo.cap_bytes, tid, TILE_THREADS, [] { __syncthreads(); });
        { // does any real base of the tile fail the ACGT test?
            uint32_t any = 0;
            const uint32_t lo_bit = s.fbase, hi_bit = s.fbase + B;
            for (uint32_t wd = tid; wd * 32 < hi_bit; wd += TILE_THREADS) {
                uint32_t m = s.inv[wd];
                const uint32_t b0 = wd * 32;
                if (b0 < lo_bit) m &= ~0u << (lo_bit - b0);
                if (b0 + 32 > hi_bit) m &= ~0u >> (b0 + 32 - hi_bit);
                any |= m;
            }
            if (any) misc[4] = 1;
        }
        for (uint32_t c = tid; c * 32 < B; c += TILE_THREADS) { // chunk map: last read starting at or before byte 32c
            const uint32_t pos = c * 32;
            int lo = 0, hi = Tn - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (rstart[mid] <= pos) lo = mid; else hi = mid - 1; }
            tmap[c] = (uint32_t)lo;
        }
        __syncthreads();
        const bool has_invalid = misc[4] != 0;
        auto read_of = [&](uint32_t p) -> int {
            int t = (int)tmap[p >> 5];
            while (p >= rstart[t + 1]) ++t;
            return t;
        };
        // one verified candidate: dedup set, postings -> counters, running best.  rec >= 0: queue entry to rewrite
        auto take_candidate = [&](uint64_t h, uint32_t p, uint32_t hint, int rec) {
            uint32_t slot = 0, v = 0;
            bool found = false;
            if (hint != IDX_NOT_FOUND) { // the one slot whose fingerprint matched: key and value in one round trip
                const uint64_t key = ix.keys[hint];
                v = ix.vals[hint];
                slot = hint;
                found = key == h;
            }
            if (!found) found = index_lookup(ix, h, slot, v);
            (void)rec;
            if (!found) return;
            const int t = read_of(p);
            // occurrence rank of this sketch hash within the read (multiset merge, rkmh.cpp:869): exact LDS multiset,
            // entry = (slot + 1) | occurrences-1 << 27
            uint32_t rank = 0;
            {
                uint32_t* ds = dset + t * DSET;
                const uint32_t key = slot + 1u;
                uint32_t idx = (slot * 0x9E3779B1u) >> (32 - 7);
                int probe = 0;
                for (; probe < DSET; ++probe) {
                    const uint32_t old = atomicCAS(&ds[idx], 0u, key);
                    if (old == 0u) break;
                    if ((old & 0x07FFFFFFu) == key) {
                        const uint32_t o2 = atomicAdd(&ds[idx], 1u << 27);
                        rank = (o2 >> 27) + 1u;
                        if (rank >= 30u) flags[t] = 1; // count field about to overflow: general path
                        break;
                    }
                    idx = (idx + 1) & (DSET - 1);
                }
                if (probe == DSET) { flags[t] = 1; return; } // more distinct hits than the set holds: general path
            }
            uint32_t* ct = c16 + t * geo.cwords;
            for_postings(ix, v, [&](uint32_t ref, uint32_t mult) {
                if (rank < mult) {
                    const uint32_t sh = (ref & 1u) * 16u;
                    const uint32_t old = atomicAdd(&ct[ref >> 1], 1u << sh);
                    const uint32_t c = ((old >> sh) & 0xFFFFu) + 1u;
                    atomicMax(&best[t], (c << 16) | (0xFFFFu - ref));
                }
            });
        };
        const uint32_t bad_words = (B + 31) / 32 + 1;
        const uint32_t nIt = (B + TILE_THREADS - 1) / TILE_THREADS;

        // ---- phase 1: one pass over the tile's byte positions per k-mer size ----------------------
        uint32_t qcount = 0; // this wave's queue length (wave-uniform)
        for (int kk = 0; kk < (KT ? 1 : ks.n); ++kk) {
            const int k = KT ? KT : ks.k[kk];
            if (kk) __syncthreads(); // previous k done with `bad`
            for (uint32_t i = tid; i < bad_words; i += TILE_THREADS) bad[i] = 0;
            __syncthreads();
            if (tid < Tn) { // the last (len - windows) positions of every read start no window
                const uint32_t rs = rstart[tid], re = rstart[tid + 1];
                uint32_t pos = rs + (uint32_t)num_windows((int)(re - rs), k, pol.drop_last_window);
                while (pos < re) {
                    const uint32_t lo = pos & 31, n = (32 - lo) < (re - pos) ? (32 - lo) : (re - pos);
                    atomicOr(&bad[pos >> 5], (n == 32 ? ~0u : ((1u << n) - 1u)) << lo);
                    pos += n;
                }
            }
            __syncthreads();
            if (MODE == 0 && has_invalid) { // windows holding a non-ACGT base hash to 0: count them, then skip them
                for (uint32_t p = tid; p < B; p += TILE_THREADS) {
                    if (!((bad[p >> 5] >> (p & 31)) & 1u) && !window_valid<KT>(s, p, k)) {
                        atomicOr(&bad[p >> 5], 1u << (p & 31));
                        atomicAdd(&nzero[read_of(p)], 1u);
                    }
                }
                __syncthreads();
            }
            uint4 fb = make_uint4(0, 0, 0, 0); // bucket fetched for the previous position (lookup in flight)
            uint64_t hp = 0;
            uint32_t pp = 0;
            for (uint32_t it = 0; it <= nIt; ++it) {
                const uint32_t p = it * TILE_THREADS + tid;
                uint64_t h = 0;
                if (it < nIt) {
                    const bool ok = p < B && !((bad[p >> 5] >> (p & 31)) & 1u);
                    const uint32_t pc = ok ? p : 0u; // keep the LDS addresses in range for idle lanes
                    if (MODE == 1 && has_invalid && !window_valid<KT>(s, pc, k)) h = 0;
                    else if (geo.dbg & 4) {
                        h = ((uint64_t)(s.fwd[(s.fbase + pc) >> 2] * 0x9E3779B1u) << 32) | (s.rc[(B - (uint32_t)k - pc) >> 2] * 0x85EBCA6Bu);
                    } else {
                        const uint64_t f = murmur_window<KT, FOLD>(s.fwd, s.fbase + pc, k, pol.seed, pol.fold);
                        const uint64_t r = murmur_window<KT, FOLD>(s.rc, B - (uint32_t)k - pc, k, pol.seed, pol.fold);
                        h = f < r ? f : r;
                    }
                    if (MODE == 1) {
                        if (ok && (pol.counter_counts_zero || h != 0)) atomicAdd(&counter[h % slots], 1);
                        continue;
                    }
                    h = ok ? h : 0;
                    if (counter && ok) { // mask_by_frequency, rkmh.cpp:916
                        const int c = counter[h % slots];
                        if (pol.mask_strict_less ? (c < min_occ) : (c <= min_occ)) h = 0;
                    }
                    if (ok && h == 0) atomicAdd(&nzero[read_of(p)], 1u);
                }
                // consume the lookup issued one iteration ago: fingerprint matches / full buckets are queued
                if (!(geo.dbg & 1)) {
                    const uint32_t fp = index_fp(hp);
                    const uint32_t mm = (fb.x == fp ? 1u : 0u) | (fb.y == fp ? 2u : 0u) | (fb.z == fp ? 4u : 0u) | (fb.w == fp ? 8u : 0u);
                    const bool cand = hp != 0 && (mm != 0 || fb.w != 0);
                    const uint64_t m = __ballot(cand);
                    if (cand) {
                        const uint32_t hint = (mm != 0 && (mm & (mm - 1u)) == 0)
                                                  ? 4u * index_bucket(hp, ix.bshift) + ((uint32_t)__ffs((int)mm) - 1u) : IDX_NOT_FOUND;
                        const uint32_t e = qcount + (uint32_t)__popcll(m & lt_mask);
                        if (e < QW) { const uint32_t q = (uint32_t)wave * QW + e; qh[q] = hp; qp[q] = pp; qs[q] = hint; }
                        else take_candidate(hp, pp, hint, -1); // queue full: handle in place
                    }
                    qcount += (uint32_t)__popcll(m);
                }
                if (geo.dbg & 8) { if (h == 0x1234567ull) nzero[0] = 1; } else
                if (h != 0) fb = ix.fpb[index_bucket(h, ix.bshift)]; // issue; lands while the next position is hashed
                hp = h;
                pp = p;
            }
        }
        if (MODE == 1) continue;
        if (lane == 0) misc[wave] = qcount < QW ? qcount : QW;
        __syncthreads();
        // ---- phase 1b: drain the candidate queue, every lane busy ----------------------------------
        const uint32_t qc0 = misc[0], qc1 = misc[1], qc2 = misc[2], qc3 = misc[3];
        const uint32_t qn = qc0 + qc1 + qc2 + qc3;
        auto qindex = [&](uint32_t e) -> uint32_t { // e-th queued candidate -> position in the segmented queue
            if (e < qc0) return e;
            e -= qc0;
            if (e < qc1) return QW + e;
            e -= qc1;
            if (e < qc2) return 2 * QW + e;
            return 3 * QW + (e - qc2);
        };
        for (uint32_t e = tid; e < qn && !(geo.dbg & 32); e += TILE_THREADS) {
            const uint32_t q = qindex(e);
            take_candidate(qh[q], qp[q], qs[q], (int)q);
        }
        __syncthreads();

        // ---- phase 2: 16 lanes per read -----------------------------------------------------------
        {
            const int g = tid >> 4, sl = tid & 15;
            for (int t = g; t < Tn; t += GROUPS) {
                uint32_t* ct = c16 + t * geo.cwords;
                const int nmins = (int)nwin[t] - (int)nzero[t];
                bool reroute = nmins > S; // bottom-S selection matters: general path
                uint32_t bk = best[t];
                if ((geo.dbg & 2) && !reroute) {
                    for (int w = sl; w < geo.cwords; w += 16) ct[w] = 0;
                    if (sl == 0) reinterpret_cast<int4*>(out4)[r0 + t] = make_int4(0, (int)(bk >> 16), 0, nmins);
                    continue;
                }
                if (flags[t]) reroute = true; // hit multiset overflowed: exact answer comes from the general path
                if (reroute) {
                    for (int w = sl; w < geo.cwords; w += 16) ct[w] = 0;
                    if (sl == 0) reinterpret_cast<int4*>(out4)[r0 + t] = make_int4(-2, 0, 0, 0);
                    continue;
                }
                // first max wins (rkmh.cpp:878); diff = max - best EARLIER score (untouched refs score 0; none => -1)
                const int max_id = bk ? (int)(0xFFFFu - (bk & 0xFFFFu)) : 0;
                const int max_shared = (int)(bk >> 16);
                int prev = max_id > 0 ? 0 : -1;
                for (int w = sl; 2 * w < max_id; w += 16) {
                    const uint32_t x = ct[w];
                    const int c0 = (int)(x & 0xFFFFu), c1 = (int)(x >> 16);
                    if (c0 > prev) prev = c0;                       // ref 2w < max_id
                    if (2 * w + 1 < max_id && c1 > prev) prev = c1;
                }
                prev = row_max_i32(prev);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                for (int w = sl; w < geo.cwords; w += 16) ct[w] = 0;
                if (sl == 0) reinterpret_cast<int4*>(out4)[r0 + t] = make_int4(max_id, max_shared, max_shared - prev, nmins);
            }
        }
    }
}

static TileGeom make_geom(int maxlen, int nref) {
    TileGeom g;
    if (maxlen < 1) maxlen = 1;
    int T = 4800 / maxlen;
    if (const char* e = getenv("RKMH_TILE_T")) T = atoi(e);
    if (T > 32) T = 32;
    if (T < 4) T = 4;
    g.T = T;
    g.cap_bytes = T * maxlen;
    g.qcap = ((g.cap_bytes / 4) + 255) & ~255; // about a quarter of the windows may be candidates before in-place handling
    if (const char* e = getenv("RKMH_TILE_QCAP")) g.qcap = atoi(e);
    g.cwords = (nref + 1) / 2;
    g.dbg = 0;
    if (const char* e = getenv("RKMH_DBG")) g.dbg = atoi(e);
    return g;
}

// T x nref 16-bit counters must fit beside the tile in LDS; reference ids must fit 16 bits
bool classify_tile_supported(int nref) { return nref <= 2048; }

hipError_t launch_classify_tile(const uint8_t* bases, const uint32_t* offs, uint32_t nreads, const KsArr& ks, int S,
                                const RefIndex& ix, int32_t* counter, uint64_t slots, int min_occ, int mode,
                                int32_t* out4, const DevPolicy& pol, int maxlen, hipStream_t st) {
    if (nreads == 0) return hipSuccess;
    TileGeom geo = make_geom(maxlen, mode == 0 ? ix.nref : 0);
    if (mode != 0) geo.qcap = 0;
    while (tile_lds_bytes(geo) > 64 * 1024 && geo.T > 4) { // keep at least two workgroups per CU
        geo.T /= 2;
        geo.cap_bytes = geo.T * maxlen;
        if (mode == 0) geo.qcap = ((geo.cap_bytes / 4) + 255) & ~255;
    }
    const size_t lds = tile_lds_bytes(geo);
    const uint32_t ntiles = (nreads + (uint32_t)geo.T - 1) / (uint32_t)geo.T;
    uint32_t grid = ntiles;
    if (const char* g = getenv("RKMH_TILE_GRID")) { uint32_t v = (uint32_t)atoi(g); if (v && v < grid) grid = v; }
    const bool k16 = (ks.n == 1 && ks.k[0] == 16);
#define RK_LAUNCH(KT, MODE, FOLD)                                                                                    \
    do {                                                                                                             \
        if (lds > 64 * 1024) {                                                                                       \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_classify_tile<KT, MODE, FOLD>),       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
            if (e != hipSuccess) return e;                                                                           \
        }                                                                                                            \
        hipLaunchKernelGGL((k_classify_tile<KT, MODE, FOLD>), dim3(grid), dim3(TILE_THREADS), lds, st, bases, offs,   \
                           nreads, ks, S, ix, counter, slots, min_occ, out4, pol, geo);                              \
    } while (0)
#define RK_LAUNCH_F(KT, MODE)                                                                                        \
    do {                                                                                                             \
        if (pol.fold == 0) RK_LAUNCH(KT, MODE, 0);                                                                   \
        else if (pol.fold == 1) RK_LAUNCH(KT, MODE, 1);                                                              \
        else RK_LAUNCH(KT, MODE, 2);                                                                                 \
    } while (0)
    if (mode == 0) { if (k16) RK_LAUNCH_F(16, 0); else RK_LAUNCH_F(0, 0); }
    else           { if (k16) RK_LAUNCH_F(16, 1); else RK_LAUNCH_F(0, 1); }
#undef RK_LAUNCH_F
#undef RK_LAUNCH
    return hipGetLastError();
}

} // namespace rk
