// tools/ubench/rk_classify_ablation.hip -- LAB COPY of rkmh_amd/csrc/rk_classify.hip as of round 5, kept for its timing experiments
// (-DRK_ABLATE=1 builds read RKMH_DBG: parts of the kernel switched off or faked, WRONG results; RKMH_TILE_* geometry overrides).
// NOT built by the Makefile and not part of the product; the numbers are in profiles/r03_ablation.txt, r03_k20_ablation.txt.
// rk_classify.hip -- the fused per-read kernel of the classify/stream hot path (gfx950, wave64).
//
// Replaces the body of main_stream's read loop, /root/reference/src/rkmh.cpp:856-888 (and the two
// passes of the -M variant, :904-934):
//     to_upper -> calc_hashes -> [mask_by_frequency] -> minhashes -> R x hash_intersection_size -> argmax/diff
// for every read whose non-zero hashes all fit the sketch (n <= S: minhashes() keeps everything, so its
// sort cannot change any intersection size and is skipped).  Other reads are flagged max_id = -2 and
// rerouted by the host through k_hash_tiles + k_sort_intersect.
//
// Work decomposition: ONE WAVE = one tile of T consecutive reads (T = 4 for 150 bp reads); a workgroup is a
// single wave, so nothing ever waits at a workgroup barrier and the 6 waves a SIMD holds drift apart into
// different phases -- latency-bound phases of one wave hide under the hashing of the others (measured: workgroups
// that all start together and stay in step are 17 % slower).  The kernel is bound by VALU issue (PMC: 92 % of the
// issue slots, next to an LDS pipeline that is 62 % busy), so every design decision below is about wave-level
// instruction count; occupancy only has to stay at 6 waves/SIMD (registers <= 80, LDS <= 6.5 KB per tile).
//   phase 0  the tile's bases are ONE contiguous byte range of the batch.  They (and the tile's offsets)
//            were prefetched into registers while the previous tile was hashed; they are written to LDS as
//            an upper-cased forward image and a reverse-complement image (v_perm_b32 letter tables).  A ballot
//            says whether any base fails the ACGT test; only such tiles (and tiles of ragged reads) build the
//            validity bitmap and the bitmap of byte positions that start no hashable window.
//   phase 1  the tile's windows are flattened over the 64 lanes (division-free running mapping window -> read,
//            position); the hot loop body is one basic block: two UNALIGNED 16-byte LDS reads (the two strands'
//            windows, no alignment funnel), both murmur3 chains, one 4-byte load (saddr form) of the window's word
//            of the first-level filter -- an L2-resident bit array in which every sketch hash sets two bits --
//            whose latency hides behind the NEXT window's hashing.  Windows that pass (every hit, about 1 window
//            in 8, plus a fraction of a percent) go to a wave-private LDS queue of 16-byte entries (ballot +
//            mbcnt, no atomics).  (RKMH_PREFILTER=0, and the masked -M form while the bucket table fits L2, load and
//            test the 16-byte fingerprint bucket of the reference index instead.)
//   drain    when the queue holds more than a wave's worth (and at the end) it is emptied with every lane
//            busy: lookup in the bucket table (fingerprints, then the 64-bit key: exactness comes from here), exact occurrence rank of the sketch hash within the read (LDS
//            multiset: the merge of rkmh.cpp:869 counts min(multiplicities)), postings added to per-read
//            packed LDS counters (8 bits when no read has more than 255 windows, else 16); since counts only grow, an LDS atomicMax of (count, -ref) per increment
//            leaves (max_shared, first max_id) behind without any scan.
//   phase 2  16 lanes per read (4 reads = one wave): best earlier score for `diff` (DPP row reduction over
//            the counters), one int4 per read, counters re-zeroed.
// Integer work only (no MFMA).
//
// Every window is hashed in this kernel (the hash-space form).  It serves -M (both passes), several k, and k outside 8..16; plain
// classification with a single k from 8 to 16 runs the k-mer-space kernel of rk_kmer.hip, which hashes nothing.  Bound twice over on
// MI355X: 92 % of the VALU issue slots AND the L2 request rate (one scattered 4-byte filter probe per window).
#include "rk_kernels.hpp"

#include <cstdlib>
#include <cstring>

namespace rk {

// Ablation switches used to attribute kernel time to its parts: build with
// -DRK_ABLATE=1 and set RKMH_DBG (1 no queueing, 2 no phase 2, 4 cheap hash, 8 no bucket loads, 32 no drain,
// 64 no hit multiset, 128 no counter updates, 256 no index verification in the drain).  Always available:
// RKMH_DBG=1024 turns the split-strand last step off (A/B), RKMH_TILE_* override the tile geometry (read once per process).
#ifndef RK_ABLATE
#define RK_ABLATE 0
#endif
#define RK_DBG(bit) (RK_ABLATE && (geo.dbg & (bit)))

constexpr int WAVE = 64;
// Occupancy target: the kernel is sensitive to it (5 -> 6 waves/SIMD = +11 % measured), so registers and LDS are
// both budgeted for it: VGPRs <= 512 / waves, LDS per single-wave workgroup <= 160 KB / (4 * waves).
#ifndef RK_WAVES_PER_SIMD
#define RK_WAVES_PER_SIMD 6
#endif
constexpr int PF_MAX = 6; // prefetched base dwords per lane: tile bytes <= PF*64*4 - 8 (template parameter PF = 2, 3 or 6)

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
// max over a 16-lane DPP row; every lane of the row ends with the result
__device__ __forceinline__ int row_max_i32(int v) {
    int t;
    t = dpp_i32<0xB1>(v); v = t > v ? t : v;   // quad_perm [1,0,3,2]
    t = dpp_i32<0x4E>(v); v = t > v ? t : v;   // quad_perm [2,3,0,1]
    t = dpp_i32<0x141>(v); v = t > v ? t : v;  // row_half_mirror
    t = dpp_i32<0x140>(v); v = t > v ? t : v;  // row_mirror
    return v;
}

// max over an aligned group of 8 lanes (half a DPP row)
__device__ __forceinline__ int half_row_max_i32(int v) {
    int t;
    t = dpp_i32<0xB1>(v); v = t > v ? t : v;   // quad_perm [1,0,3,2]
    t = dpp_i32<0x4E>(v); v = t > v ? t : v;   // quad_perm [2,3,0,1]
    t = dpp_i32<0x141>(v); v = t > v ? t : v;  // row_half_mirror
    return v;
}

// Orders this wave's LDS traffic for cross-lane hand-offs.  A workgroup is one wave: LDS instructions of a
// wave execute in order, so no hardware wait is needed -- only the compiler must not move accesses across.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct TileGeom {
    int32_t T;          // reads per tile (<= 16)
    int32_t cap_bytes;  // staged bytes per tile
    int32_t qcap;       // candidate queue entries (>= 128)
    int32_t cwords;     // counter words per read = ceil(nref / counters per word)
    int32_t clg;        // log2(counters per 32-bit word): 1 = 16-bit counters, 2 = 8-bit (reads with <= 255 windows)
    int32_t csparse;    // 1: the per-read counters are a small open-addressing map ref -> count (many references) and
                        //    cwords is its size (power of two); entry = (ref + 1) << 11 | count
    int32_t dset;       // slots of the per-read hit multiset (power of two)
    int32_t dbg;        // ablation switches (only read when built with -DRK_ABLATE=1)
    int32_t tpb;        // consecutive tiles per workgroup
    int32_t xcd;        // 1: workgroups that share an XCD (blockIdx % 8, round-robin dispatch) take neighbouring tiles
    uint64_t slots_m;   // floor((2^64 - 1) / slots) of the -M counter table: hash % slots by Barrett reduction (mod_slots)
    uint32_t slot_stride; // count pass that emits slots instead of counting (out4 = the flat array): entries per k-mer size
    int32_t magic_nw;   // windows (first k) of a read of the hinted length, and ...
    uint32_t magic;     // ... ceil(2^32 / magic_nw): the compact window -> read division of tiles made of such reads
    int32_t nmin_cap;   // row field 3 = min(non-zero hashes, nmin_cap) (-M with a bounded min_num: rk_set_min_num_bound; else INT_MAX)
    CompactSlots cs;    // count pass into a compact depth map (cs.tab != nullptr): `counter` then holds one entry per tracked slot
};

__host__ __device__ inline int tile_map_words(int cap_bytes) { return cap_bytes / 32 + 2; }
__host__ __device__ inline int tile_stage_dwords(const TileGeom& g) { return stage_lds_dwords(g.cap_bytes); }
__host__ __device__ inline size_t tile_lds_bytes(const TileGeom& g) {
    const size_t q_dw = 4 * (size_t)g.qcap; // queue region: 16-byte entries {hash, read}
    return ((size_t)((tile_stage_dwords(g) + 3) & ~3) + q_dw + 5 * (size_t)(g.T + 1) + 8 +
            2 * (size_t)tile_map_words(g.cap_bytes) +
            (size_t)g.T * (size_t)(g.cwords + g.dset)) * 4;
}

// for every posting (ref, mult) of an index value
template <typename F>
__device__ __forceinline__ void for_postings(const RefIndex& ix, uint32_t v, F f) {
    if (!(v >> 31)) {
        if (((v >> 29) & 3u) == 0u) f(v & 0xFFFFFu, (v >> 20) & 0x1FFu);
        else { f(v & 0x7FFu, 1u); f((v >> 11) & 0x7FFu, 1u); }
    } else {
        const uint32_t off = v & 0x7fffffffu;
        const uint32_t cnt = ix.post[off];
        for (uint32_t c = 0; c < cnt; ++c) f(ix.post[off + 1 + 2 * c], ix.post[off + 2 + 2 * c]);
    }
}

// full lookup: fingerprint scan (+ the bucket's first key id), then key and value of a matching key are fetched together;
// `slot` returns the key id (a dense number unique to the sketch hash)
// dropped: the key's entry carries the -M mask's verdict (RefIndex::kv is then the masked copy, rk_set_depth_filter with a bounded min_num)
__device__ __forceinline__ bool index_lookup(const RefIndex& ix, uint64_t h, uint32_t& slot, uint32_t& val, bool& dropped) {
    const uint32_t fp = index_fp(h);
    uint32_t b = index_bucket(h, ix.bmask);
    for (;;) {
        const uint4 f = ix.fpb[b];
        const uint32_t id0 = ix.base[b]; // fetched in the same round as the fingerprints
        uint32_t m = index_match_mask(f, fp);
        while (m) {
            const uint32_t q = (uint32_t)__ffs((int)m) - 1u;
            m &= m - 1u;
            const uint32_t s = id0 + q;
            const uint4 e = ix.kv[s]; // key and value in one 16-byte fetch from the dense (L2-resident) key array
            if (e.x == (uint32_t)h && e.y == (uint32_t)(h >> 32)) { slot = s; val = e.z; dropped = e.w != 0u; return true; }
        }
        if (!(f.x & IDX_OVF)) return false;
        b = (b + 1) & ix.bmask;
    }
}

// The lookup of a window is issued one step ahead: the bucket (or filter word) of window i is requested at the end of step i and
// examined after window i + 1 has been hashed.  Plain loads: hipcc places the wait in front of the first use, i.e. behind the next
// window's hashing (rounds 1-4 issued these loads through inline asm with hand-placed s_waitcnt and policed the ISA with a lint;
// k_classify_kmer showed that the compiler keeps the same overlap on its own).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bucket_load_async(const uint4* base, uint32_t byte_off, u32x4& f) {
    f = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(base) + byte_off);
}
__device__ __forceinline__ void bucket_wait(u32x4&) {}
// the same for one word of the first-level filter (RefIndex::pre)
__device__ __forceinline__ void word_load_async(const uint32_t* base, uint32_t byte_off, uint32_t& f) {
    f = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(base) + byte_off);
}
__device__ __forceinline__ void word_wait(uint32_t&) {}

// MODE 0: classify; MODE 1: count pass of -M (rkmh.cpp:904-910); MODE 2: classify with the -M mask (rkmh.cpp:916)
// MODE_ 3 / 4: MODE 0 / 2 with the first-level filter of large panels (RefIndex::pre) in front of the bucket table
template <int KT, int MODE_, int FOLD, int PF>
__global__ __launch_bounds__(WAVE, RK_WAVES_PER_SIMD) void k_classify_tile(const uint8_t* __restrict__ bases, const uint32_t* __restrict__ offs,
                                                           uint32_t nreads, KsArr ks, int S, RefIndex ix, int32_t* counter,
                                                           uint64_t slots, int min_occ, int32_t* out4, DevPolicy pol, TileGeom geo) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr bool PRE = MODE_ >= 3;
    constexpr int MODE = MODE_ == 3 ? 0 : (MODE_ == 4 ? 2 : MODE_);
    const int T = geo.T;
    const uint32_t QCAP = (uint32_t)geo.qcap;
    const uint32_t DS = (uint32_t)geo.dset;
    uint32_t* stage = smem;
    // candidate queue: one 16-byte entry per window {hash lo, hash hi, read within the tile, -} (one ds_write_b128)
    uint4* qe = reinterpret_cast<uint4*>(stage + ((tile_stage_dwords(geo) + 3) & ~3));
    uint32_t* rstart = reinterpret_cast<uint32_t*>(qe + QCAP);   // [T+1] byte offset of read t inside the tile
    uint32_t* nwin = rstart + (T + 1);                           // [T+1] windows of read t (all k)
    uint32_t* nzero = nwin + (T + 1);                            // [T+1] zero hashes per read
    uint32_t* best = nzero + (T + 1);                            // [T+1] max over increments of (count << 16 | 0xFFFF - ref)
    uint32_t* flags = best + (T + 1);                            // [T+1] read must go through the general path
    uint32_t* misc = flags + (T + 1);                            // [0] tile has invalid bases
    uint32_t* bad = misc + 8;                                    // bit p set <=> no hashable window starts at tile byte p
    uint32_t* tmap = bad + tile_map_words(geo.cap_bytes);        // read index holding tile byte 32*c
    uint32_t* c16 = tmap + tile_map_words(geo.cap_bytes);        // [T][cwords] packed 16-bit per-reference counters
    uint32_t* dset = c16 + T * geo.cwords;                       // [T][DS] multiset of the slots the read has hit
    // [64][2] hits with several postings (drain).  Aliases the first 32 queue entries: a drain step has its 64
    // entries in registers before it writes here, and entries left for later sit at index >= 64.
    uint32_t* mq = reinterpret_cast<uint32_t*>(qe);
    const int lane = threadIdx.x;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t clg = (uint32_t)geo.clg, cper_m1 = (1u << clg) - 1u, cbits = 32u >> clg, cmask = (1u << cbits) - 1u;

    if (MODE != 1) { // counters are re-zeroed by phase 2 after use
        for (int i = lane; i < T * geo.cwords; i += WAVE) c16[i] = 0;
    }
    const uint32_t ntiles = (nreads + (uint32_t)T - 1) / (uint32_t)T;

    // Software prefetch across tiles: while tile i is hashed, the offsets and the raw base dwords of this wave's
    // next tile travel into registers, so phase 0 of the next tile touches no global memory.
    uint32_t pf[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) pf[q] = 0u;
    uint32_t cur_a = 0, cur_b = 0, cur_o = 0; // tile byte range [a,b) in the batch, this lane's read offset
    // Count pass into a COMPACT depth map (MODE 1, geo.cs.tab): only slots that some index key maps to are tracked (a few hundred
    // thousand out of 2 * 10^8).  One bit of an L2-resident filter -- its word requested one step ahead, like the classifier's
    // first-level filter -- turns away the windows whose slot is not among them; the rest (true occurrences of sketch hashes and
    // the few collisions: a dozen per read instead of 135) are queued in LDS and, 64 at a time with every lane busy, find their
    // entry in a small table and count with an atomic.  The queue outlives the tiles: it holds slots, nothing of the tile.
    uint32_t cslot = CS_EMPTY;   // this step's slot (CS_EMPTY: the lane hashed no countable window)
    uint32_t cprev = CS_EMPTY;   // last step's slot, whose filter word is in flight ...
    uint32_t cword = 0;          // ... here
    uint32_t ccount = 0;         // queued slots (wave-uniform)
    auto compact_drain = [&](uint32_t from, uint32_t n_) { // queue entries [from, from + n_), n_ <= 64
        uint32_t* cq = reinterpret_cast<uint32_t*>(qe);
        if ((uint32_t)lane < n_) {
            const uint32_t s32 = cq[from + (uint32_t)lane];
            uint32_t idx = (s32 * 0x85EBCA6Bu) >> geo.cs.tab_shift;
            for (;;) {
                const uint2 e = geo.cs.tab[idx];
                if (e.x == s32) { atomicAdd(&counter[e.y], 1); break; }
                if (e.x == CS_EMPTY) break; // a false positive of the filter
                idx = (idx + 1u) & geo.cs.tab_mask;
            }
        }
    };
    auto tile_reads = [&](uint32_t tl) -> int {
        const uint32_t r = tl * (uint32_t)T;
        return (int)((nreads - r) < (uint32_t)T ? (nreads - r) : (uint32_t)T);
    };
    auto load_offsets = [&](uint32_t tl, uint32_t& a_, uint32_t& b_, uint32_t& o_) {
        const uint32_t r = tl * (uint32_t)T;
        const int n = tile_reads(tl);
        a_ = offs[r];
        b_ = offs[r + (uint32_t)n];
        o_ = offs[r + (uint32_t)(lane <= n ? lane : n)];
    };
    // Addresses are clamped into the tile (out-of-range lanes are zeroed when the image is built).
    auto load_bases = [&](uint32_t a_, uint32_t b_) {
        const uint32_t* g32 = reinterpret_cast<const uint32_t*>(bases) + (a_ >> 2);
        uint32_t ndw = ((a_ & 3u) + (b_ - a_) + 3u) >> 2;
        if (b_ - a_ > (uint32_t)geo.cap_bytes) ndw = 0; // oversized tile: rerouted, nothing to stage
#pragma unroll
        for (int q = 0; q < PF; ++q) { // register q of a lane = fwd-image dword jf = 64q + lane = global dword jf - 1
            if ((uint32_t)q * WAVE <= ndw) { // wave-uniform
                const uint32_t jf = (uint32_t)q * WAVE + (uint32_t)lane;
                const uint32_t j = jf >= 1 ? (jf <= ndw ? jf - 1 : (ndw ? ndw - 1 : 0u)) : 0u;
                pf[q] = g32[j];
            }
        }
    };
    auto wait_bases = [&]() {}; // (plain loads: the compiler waits where the registers are first read)
    // XCD-aware tile ownership: the dispatcher deals workgroups round-robin over the 8 XCDs, each with its own L2.
    // Workgroup b is given the tiles of "virtual" workgroup (b % 8) * ceil(grid / 8) + b / 8, so the workgroups of one
    // XCD walk one contiguous eighth of the batch and the cache lines that straddle two tiles are fetched into one L2.
    uint32_t vb = blockIdx.x;
    if (geo.xcd) { const uint32_t per = (gridDim.x + 7u) >> 3; vb = (blockIdx.x & 7u) * per + (blockIdx.x >> 3); }
    const uint32_t tile0 = vb * (uint32_t)geo.tpb;
    const uint32_t tile_end = (tile0 + (uint32_t)geo.tpb) < ntiles ? (tile0 + (uint32_t)geo.tpb) : ntiles;
    if (tile0 < ntiles) { load_offsets(tile0, cur_a, cur_b, cur_o); load_bases(cur_a, cur_b); }

    for (uint32_t tile = tile0; tile < tile_end; ++tile) {
        const uint32_t r0 = tile * (uint32_t)T;
        // wave-uniform values are moved to SGPRs explicitly: hipcc cannot prove that values loaded from one address by
        // every lane are uniform, and keeps loop counters derived from them in VGPRs with exec-mask loop control
        const uint32_t tstart = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur_a);
        const uint32_t B_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)(cur_b - cur_a));
        const uint32_t ntile = tile + 1u < tile_end ? tile + 1u : ntiles; // this workgroup's next tile (none: ntiles)
        uint32_t nxt_a = 0, nxt_b = 0, nxt_o = 0;
        if (ntile < ntiles) load_offsets(ntile, nxt_a, nxt_b, nxt_o); // lands during phase 0
        wave_sync(); // previous tile fully consumed
        // A tile holding a read longer than the caller's hint is rerouted by the host (rows of -2; the count pass skips it).
        // Here it becomes an EMPTY tile (no reads, no bytes) that runs through the body like any other.
        const bool oversized = B_all > (uint32_t)geo.cap_bytes;
        if (MODE != 1 && oversized && lane < tile_reads(tile)) reinterpret_cast<int4*>(out4)[r0 + lane] = make_int4(-2, 0, 0, 0);
        const int Tn = oversized ? 0 : tile_reads(tile);
        const uint32_t B = oversized ? 0u : B_all;
        // ---- phase 0, interval 1: everything that needs only registers ------------------------------
        Staged s;
        {
            const int fwd_dw = (FWD_PAD + 3 + geo.cap_bytes + TAIL_PAD + 3) / 4;
            const int rc_dw = (geo.cap_bytes + TAIL_PAD + 3) / 4;
            s.fwd = stage; s.rc = stage + fwd_dw; s.inv = s.rc + rc_dw;
            s.fbase = FWD_PAD + (tstart & 3u); s.nbases = B;
        }
        const uint32_t o_next = (uint32_t)__shfl_down((int)cur_o, 1); // offset of read lane+1
        if (lane <= Tn) {
            rstart[lane] = cur_o - tstart;
            if (lane < Tn) {
                const int len = (int)(o_next - cur_o);
                uint32_t nw = 0;
                if (KT) nw = (uint32_t)num_windows(len, KT, pol.drop_last_window);
                else for (int j = 0; j < ks.n; ++j) nw += (uint32_t)num_windows(len, ks.k[j], pol.drop_last_window);
                nwin[lane] = nw; nzero[lane] = 0; best[lane] = 0;
                // a read with more windows than a packed counter can count (only possible when the caller's length hint was
                // too small for this read: 8-bit counters are chosen for hints of <= 255 windows) takes the general path
                flags[lane] = nw > (geo.csparse ? 0x7FFu : cmask) ? 1u : 0u;
            }
        }
        const uint32_t ulen = (uint32_t)__builtin_amdgcn_readfirstlane((int)(o_next - cur_o)); // length of read 0
        const bool uniform = __ballot(lane < Tn && (o_next - cur_o) != ulen) == 0ull;     // every read of the tile as long
        const uint32_t bad_words = (B + 31) / 32 + 1;
        if (MODE != 1)
            for (uint32_t i = lane; i < (uint32_t)Tn * DS; i += WAVE) dset[i] = 0;
        const uint32_t ndw = ((tstart & 3u) + B + 3u) >> 2; // global dwords covering the tile
        // byte masks of the tile's first and last global dword (bytes of neighbouring tiles are not ours to judge)
        const uint32_t m_first = ~0u << (8u * (tstart & 3u));
        const uint32_t endb = ((tstart & 3u) + B) & 3u;
        const uint32_t m_last = endb ? (~0u >> (8u * (4u - endb))) : ~0u;
        uint32_t anyinv = 0;
        wait_bases();
#pragma unroll
        for (int q = 0; q < PF; ++q) { // upper-cased forward image; does any base of the tile fail the ACGT test?
            if ((uint32_t)q * WAVE <= ndw) { // wave-uniform
                const uint32_t jf = (uint32_t)q * WAVE + (uint32_t)lane;
                const bool in = jf >= 1 && jf <= ndw;
                const uint32_t x = in ? upper4(pf[q]) : 0u;
                if (jf <= ndw) s.fwd[jf] = x;
                uint32_t m = in ? ~0u : 0u;
                if (jf == 1) m &= m_first;
                if (jf == ndw) m &= m_last;
                anyinv |= acgt_mismatch4(x) & m;
            }
        }
        cur_a = nxt_a; cur_b = nxt_b; cur_o = nxt_o;
        const bool has_invalid = __ballot(anyinv != 0u) != 0ull;
        // "plain" tiles (all reads equally long, no invalid base, at least two windows per read: the common case) need neither the
        // start bitmap nor the position -> read map
        uint32_t nw_min;
        if (KT) nw_min = (uint32_t)num_windows((int)ulen, KT, pol.drop_last_window);
        else {
            nw_min = ~0u;
            for (int j = 0; j < ks.n; ++j) { const uint32_t v = (uint32_t)num_windows((int)ulen, ks.k[j], pol.drop_last_window); nw_min = v < nw_min ? v : nw_min; }
        }
        const bool plain = uniform && !has_invalid && nw_min >= 2u; // >= 2: the compact mapping divides by the window count
        wave_sync();
        // ---- phase 0, interval 2: reverse-complement image; for the rare tile with a non-ACGT base the validity bitmap ----
        {
            const uint32_t nrc = (B + 3) >> 2;
            for (uint32_t q = lane; q < nrc; q += WAVE) { // rc dword q = reversed complement of fwd bytes [fbase+B-4-4q, +4)
                const uint32_t pp_ = s.fbase + B - 4u - 4u * q; // >= fbase - 3: the pad dword in front absorbs it
                s.rc[q] = revcomp4(lds_load4_unaligned(s.fwd, pp_));
            }
        }
        if (has_invalid) { // 4 validity bits per dword, 8 lanes = one bitmap word (bits indexed by fwd byte position)
            for (uint32_t j0 = 0; j0 <= ndw; j0 += WAVE) {
                const uint32_t jf = j0 + (uint32_t)lane;
                const uint32_t x = jf <= ndw ? s.fwd[jf] : 0u;
                uint32_t n = invalid4(x) << (4 * (lane & 7));
                n |= (uint32_t)__shfl_xor((int)n, 1);
                n |= (uint32_t)__shfl_xor((int)n, 2);
                n |= (uint32_t)__shfl_xor((int)n, 4);
                if ((lane & 7) == 0 && (jf >> 3) <= (ndw >> 3)) s.inv[jf >> 3] = n;
            }
        }
        auto mark_tails = [&](int k) { // the last (len - windows) positions of every read start no window
            if (lane < Tn) {
                const uint32_t rs = rstart[lane], re = rstart[lane + 1];
                uint32_t pos = rs + (uint32_t)num_windows((int)(re - rs), k, pol.drop_last_window);
                while (pos < re) {
                    const uint32_t lo = pos & 31, n = (32 - lo) < (re - pos) ? (32 - lo) : (re - pos);
                    atomicOr(&bad[pos >> 5], (n == 32 ? ~0u : ((1u << n) - 1u)) << lo);
                    pos += n;
                }
            }
        };
        wave_sync();
        if (!plain) {
            for (uint32_t i = lane; i < bad_words; i += WAVE) bad[i] = 0;
            for (uint32_t c = lane; c * 32 < B; c += WAVE) { // chunk map: last read starting at or before byte 32c
                const uint32_t pos = c * 32;
                int lo = 0, hi = Tn - 1;
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (rstart[mid] <= pos) lo = mid; else hi = mid - 1; }
                tmap[c] = (uint32_t)lo;
            }
            wave_sync();
            mark_tails(KT ? KT : ks.k[0]);
            wave_sync();
        }
        // the next tile's bases travel while this tile is hashed
        if (ntile < ntiles) load_bases(cur_a, cur_b);
        auto read_of = [&](uint32_t p) -> int {
            int t = (int)tmap[p >> 5];
            while (p >= rstart[t + 1]) ++t;
            return t;
        };

        // +1 for reference `ref` of read t; the monotone counters make (max_shared, first max_id) a running atomicMax
        // count_posting: the +1 alone; returns the candidate for the running maximum (0: nothing to report)
        auto count_posting = [&](int t, uint32_t ref) -> uint32_t {
            if RK_DBG(128) return 0u;
            // packed counters: 16 bits each, or 8 bits when no read of the batch has more than 255 windows (a count never
            // exceeds the number of windows, so no field can carry into its neighbour)
            uint32_t cnt;
            if (geo.csparse) { // wave-uniform: large reference panels keep (ref, count) pairs of the references a read really hits
                uint32_t* row = c16 + t * geo.cwords;
                const uint32_t M1 = (uint32_t)geo.cwords - 1u, key = ref + 1u;
                uint32_t idx = ((ref * 0x9E3779B1u) >> 16) & M1, probe = 0;
                cnt = 0;
                for (; probe <= M1; ++probe) {
                    const uint32_t old = atomicCAS(&row[idx], 0u, (key << 11) | 1u);
                    if (old == 0u) { cnt = 1u; break; }
                    if ((old >> 11) == key) { cnt = (atomicAdd(&row[idx], 1u) & 0x7FFu) + 1u; break; }
                    idx = (idx + 1u) & M1;
                }
                if (probe > M1) { flags[t] = 1; return 0u; } // the read hits more references than the map holds: general path
            } else {
                const uint32_t sh = (ref & cper_m1) * cbits;
                const uint32_t old = atomicAdd(&c16[t * geo.cwords + (int)(ref >> clg)], 1u << sh);
                cnt = ((old >> sh) & cmask) + 1u;
            }
            return (cnt << 16) | (0xFFFFu - ref);
        };
        auto add_posting = [&](int t, uint32_t ref) {
            const uint32_t v = count_posting(t, ref);
            if (v) atomicMax(&best[t], v);
        };
        // one candidate window: verify the full key, find the occurrence rank of this sketch hash within the read
        // (exact LDS multiset: entry = (slot + 1) | (occurrences - 1) << 27) and add the postings the multiset merge of
        // rkmh.cpp:869 would count.  A hit with several postings is returned (read << 8 | rank, postings offset) for
        // the 16-lanes-per-hit pass instead of looping here with 63 lanes idle.
        auto apply_hit = [&](uint32_t slot, uint32_t v, int t, uint32_t& m_tr, uint32_t& m_off) -> bool {
            uint32_t rank = 0;
            if (!RK_DBG(64)) {
                uint32_t* ds = dset + (uint32_t)t * DS;
                const uint32_t key = slot + 1u;
                uint32_t idx = ((slot * 0x9E3779B1u) >> 16) & (DS - 1);
                uint32_t probe = 0;
                for (; probe < DS; ++probe) {
                    const uint32_t old = atomicCAS(&ds[idx], 0u, key);
                    if (old == 0u) break;
                    if ((old & 0x07FFFFFFu) == key) {
                        const uint32_t o2 = atomicAdd(&ds[idx], 1u << 27);
                        rank = (o2 >> 27) + 1u;
                        if (rank >= 30u) flags[t] = 1; // count field about to overflow: general path
                        break;
                    }
                    idx = (idx + 1) & (DS - 1);
                }
                if (probe == DS) { flags[t] = 1; return false; } // more distinct hits than the set holds: general path
            }
            if (!(v >> 31)) { // one posting (with multiplicity) or two single postings, stored inline
                if (((v >> 29) & 3u) == 0u) {
                    if (rank < ((v >> 20) & 0x1FFu)) add_posting(t, v & 0xFFFFFu);
                } else if (rank == 0) {
                    add_posting(t, v & 0x7FFu);
                    add_posting(t, (v >> 11) & 0x7FFu);
                }
                return false;
            }
            m_tr = ((uint32_t)t << 8) | rank;
            m_off = v & 0x7fffffffu;
            return true;
        };
        auto take_candidate = [&](uint64_t h, int t, uint32_t& m_tr, uint32_t& m_off) -> bool {
            uint32_t slot = 0, v = 0;
            bool dropped = false;
            if RK_DBG(256) { slot = (uint32_t)h & 0xFFFFu; v = (uint32_t)(h >> 40) % 180u | (1u << 20); } else
            if (!index_lookup(ix, h, slot, v, dropped)) return false; // re-reads the bucket (L1/L2 hit), then key + value together
            // -M with a bounded min_num: the mask is applied per KEY (the verdict rides in the key's entry) -- a masked hash is 0 (rkmh.cpp:916)
            if (dropped) { atomicAdd(&nzero[t], 1u); return false; }
            return apply_hit(slot, v, t, m_tr, m_off);
        };
        auto drain_queue = [&](uint32_t qn) {
            for (uint32_t e0 = 0; e0 < qn; e0 += WAVE) {
                const uint32_t e = e0 + (uint32_t)lane;
                uint32_t m_tr = 0, m_off = 0;
                bool multi = false;
                if (e < qn) {
                    const uint4 ce = qe[e];
                    multi = take_candidate(((uint64_t)ce.y << 32) | ce.x, (int)ce.z, m_tr, m_off);
                }
                const uint64_t mm = __ballot(multi);
                if (mm) { // hits with several postings: 16 lanes walk one hit's posting list, 4 hits at a time
                    if (multi) { const uint32_t j = (uint32_t)__popcll(mm & lt_mask); mq[2 * j] = m_tr; mq[2 * j + 1] = m_off; }
                    wave_sync();
                    const uint32_t nm = (uint32_t)__popcll(mm);
                    const int g = lane >> 4, sl = lane & 15;
                    for (uint32_t j = (uint32_t)g; j < nm; j += WAVE / 16) {
                        const uint32_t tr = mq[2 * j], off = mq[2 * j + 1];
                        const uint32_t cnt = ix.post[off];
                        // as in k_classify_kmer: the 16 lanes' candidates for the read's running maximum are reduced over the row before
                        // ONE lane issues the atomic (16 atomics on one LDS address per step otherwise)
                        for (uint32_t c0 = 0; c0 < cnt; c0 += 16) {
                            const uint32_t c = c0 + (uint32_t)sl;
                            uint32_t v = 0;
                            if (c < cnt) {
                                const uint32_t ref = ix.post[off + 1 + 2 * c], mult = ix.post[off + 2 + 2 * c];
                                if ((tr & 0xFFu) < mult) v = count_posting((int)(tr >> 8), ref);
                            }
                            v = (uint32_t)row_max_i32((int)v);
                            if (sl == 0 && v) atomicMax(&best[tr >> 8], v);
                        }
                    }
                    wave_sync();
                }
            }
        };

        // ---- phase 1: one pass over the tile's byte positions per k-mer size ----------------------
        uint32_t qcount = 0; // queue length (wave-uniform)
        for (int kk = 0; kk < (KT ? 1 : ks.n); ++kk) {
            const int k = KT ? KT : ks.k[kk];
            const TailMasks tmasks = make_tail_masks(k);
            if (kk && !plain) { // later k-mer sizes rebuild the start bitmap
                wave_sync();
                for (uint32_t i = lane; i < bad_words; i += WAVE) bad[i] = 0;
                wave_sync();
                mark_tails(k);
                wave_sync();
            }
            if (MODE != 1 && has_invalid) { // windows holding a non-ACGT base hash to 0: count them, then skip them
                for (uint32_t p = lane; p < B; p += WAVE) {
                    if (!((bad[p >> 5] >> (p & 31)) & 1u) && !window_valid<KT>(s, p, k)) {
                        atomicOr(&bad[p >> 5], 1u << (p & 31));
                        atomicAdd(&nzero[read_of(p)], 1u);
                    }
                }
                wave_sync();
            }
            // Tiles whose reads all have the same length (the common case for short-read data) and hold no invalid
            // base need no start bitmap: compact window index w -> read t = w / nw, position p = w + t * (len - nw),
            // so no lane idles on read tails (k-1 positions per read).  Other tiles walk all byte positions.
            const uint32_t nw_u = (uint32_t)num_windows((int)ulen, k, pol.drop_last_window);
            const bool compact = plain;
            const uint32_t nW = compact ? nw_u * (uint32_t)Tn : B;
            const uint32_t nIt = (nW + WAVE - 1) / WAVE;
            // compact mapping without any state: window w = it * 64 + lane lies in read t = w / nw_u and starts at tile byte
            // p = w + t * dtail.  The division is one v_mul_hi_u32 by a per-tile magic constant (exact while w * nw_u < 2^32:
            // w < 16 reads x 1528 windows), so a step spends 3 VALU on the mapping instead of a running (read, window, position)
            // triple with a compare-and-wrap per step.
            const uint32_t dtail = ulen - nw_u;           // positions at the end of a read that start no window
            // magic = ceil(2^32 / nw_u) (nw_u >= 2 for plain tiles); the host supplies it for reads of the hinted length, so
            // only tiles of another length pay for an integer division (once per tile)
            uint32_t magic = 0;
            if (compact) magic = nw_u == (uint32_t)geo.magic_nw ? (uint32_t)geo.magic
                                                                : (uint32_t)__builtin_amdgcn_readfirstlane((int)(0xFFFFFFFFu / nw_u + 1u));
            u32x4 fb = {0u, 0u, 0u, 0u}; // bucket fetched for the previous position (lookup in flight)
            uint32_t fw = 0;             // ... or its word of the first-level filter (large panels)
            uint64_t hp = 0;
            uint32_t tp = 0; // read of the previous position
            uint32_t it = 0;
            for (;;) {
                // run positions until the queue may not take another wave of candidates (or the tile is done);
                // step nIt only examines the last lookup
                for (; it <= nIt && (MODE == 1 || qcount + WAVE <= QCAP); ++it) {
                    uint32_t t = 0;
                    uint64_t h = 0;
                    if (it < nIt) {
                        // what happens to a window's canonical hash: the -M count / mask, the zero-hash tally of its read
                        auto account = [&](uint64_t& hh, uint32_t tt, uint32_t pp) {
                            if (MODE == 1) {
                                if (pol.counter_counts_zero || hh != 0) {
                                    const uint64_t slot = mod_slots(hh, slots, geo.slots_m);
                                    // slot-partitioned count (rk_count.hip): the window's slot goes to the flat array, indexed by
                                    // the byte position of the window (tails keep the sentinel); no atomic here
                                    if (geo.cs.tab) cslot = (uint32_t)slot; // compact depth map: looked up at the loop level (below), all lanes together
                                    else if (out4) reinterpret_cast<uint32_t*>(out4)[(size_t)kk * geo.slot_stride + tstart + pp] = (uint32_t)slot;
                                    else atomicAdd(&counter[slot], 1);
                                }
                            } else {
                                if (MODE == 2) { // mask_by_frequency, rkmh.cpp:916
                                    // `counter` is the KEEP bitmap here (rk_set_depth_filter: bit s = the count of slot s passes the
                                    // threshold): 1 bit per slot instead of 4 bytes -- 25 MB for the reference's 200 M slots, which
                                    // the Infinity Cache holds, where the table itself (800 MB) is a random HBM access per window
                                    const uint64_t slot = mod_slots(hh, slots, geo.slots_m);
                                    const uint32_t wbits = reinterpret_cast<const uint32_t*>(counter)[slot >> 5];
                                    if (!((wbits >> ((uint32_t)slot & 31u)) & 1u)) hh = 0;
                                }
                                if (hh == 0) atomicAdd(&nzero[tt], 1u);
                            }
                        };
                        const bool split = compact && it + 1 == nIt && nW - it * WAVE <= 32u && !RK_DBG(4) && !(geo.dbg & 1024); // wave-uniform (RKMH_DBG=1024: A/B off)
                        if (split) {
                            // The tile's last <= 32 windows: lanes l and l + 32 take window l together -- the low half hashes
                            // the forward strand, the high half the reverse complement, one exchange gives both the minimum.
                            // The step then issues one murmur per lane instead of two for a mostly idle wave.
                            const uint32_t w = it * WAVE + (uint32_t)(lane & 31);
                            t = __umulhi(w, magic);
                            const uint32_t p = w + __umul24(t, dtail);
                            if (w < nW) { // same for both lanes of a pair
                                const bool low = lane < 32;
                                const uint32_t* img = low ? s.fwd : s.rc;
                                const uint32_t off = low ? s.fbase + p : B - (uint32_t)k - p;
                                const uint64_t own = murmur_window<KT, FOLD>(img, off, k, pol.seed, pol.fold);
                                const uint64_t oth = (uint64_t)__shfl_xor((long long)own, 32);
                                if (low) { h = own < oth ? own : oth; account(h, t, p); }
                            }
                        } else {
                        bool ok;
                        uint32_t p;
                        if (compact) { // wave-uniform
                            const uint32_t w = it * WAVE + (uint32_t)lane;
                            t = __umulhi(w, magic);
                            p = w + __umul24(t, dtail);
                            ok = w < nW;
                        } else {
                            p = it * WAVE + (uint32_t)lane;
                            ok = p < B && !((bad[p >> 5] >> (p & 31)) & 1u);
                            if (ok) t = (uint32_t)read_of(p);
                        }
                        if (ok) { // idle lanes (read tails, the end of the tile) keep h = 0 and are never looked up
                            if (MODE == 1 && has_invalid && !window_valid<KT>(s, p, k)) h = 0;
                            else if RK_DBG(4) {
                                h = ((uint64_t)(s.fwd[(s.fbase + p) >> 2] * 0x9E3779B1u) << 32) | (s.rc[(B - (uint32_t)k - p) >> 2] * 0x85EBCA6Bu);
                            } else {
                                if constexpr (KT == 0) { // run-time k: both strands through one block loop, uniform tail masks
                                    h = canonical_rt(s.fwd, s.fbase + p, s.rc, B - (uint32_t)k - p, k, tmasks, pol.seed, pol.fold);
                                } else {
                                    if constexpr (FOLD == 0) { // the half swap of fold 0 happens inside the minimum
                                        const uint64_t f = murmur_window<KT, 3>(s.fwd, s.fbase + p, k, pol.seed, 0);
                                        const uint64_t r = murmur_window<KT, 3>(s.rc, B - (uint32_t)k - p, k, pol.seed, 0);
                                        h = min_swapped(f, r);
                                    } else {
                                    const uint64_t f = murmur_window<KT, FOLD>(s.fwd, s.fbase + p, k, pol.seed, pol.fold);
                                    const uint64_t r = murmur_window<KT, FOLD>(s.rc, B - (uint32_t)k - p, k, pol.seed, pol.fold);
                                    h = f < r ? f : r;
                                    }
                                }
                            }
                            account(h, t, p);
                        }
                        }
                    }
                    if (MODE == 1) {
                        if (geo.cs.tab) { // wave-uniform: the compact depth map
                            uint32_t* cq = reinterpret_cast<uint32_t*>(qe);
                            word_wait(cword);
                            const uint32_t pbit = (cprev * 0x9E3779B1u) >> geo.cs.pre_shift;
                            const bool pass = cprev != CS_EMPTY && ((cword >> (pbit & 31u)) & 1u) != 0u;
                            const uint64_t m = __ballot(pass);
                            if (pass) cq[ccount + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = cprev;
                            ccount = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ccount + (uint32_t)__popcll(m)));
                            const uint32_t nbit = (cslot * 0x9E3779B1u) >> geo.cs.pre_shift; // (CS_EMPTY looks some word up too: never tested)
                            word_load_async(geo.cs.pre, (nbit >> 5) << 2, cword);
                            cprev = cslot;
                            cslot = CS_EMPTY;
                            if (ccount >= 2u * WAVE) { // (the queue holds 3 * 64 entries)
                                wave_sync();
                                compact_drain(ccount - WAVE, WAVE);
                                ccount -= WAVE;
                                wave_sync();
                            }
                        }
                        continue;
                    }
                    // examine the lookup issued one step ago: fingerprint matches / full buckets are queued
                    if constexpr (PRE) { // large panel: the filter word decides what the drain will look up in the table
                        word_wait(fw);
                        const uint32_t bm = index_pre_bits(hp);
                        const bool cand = (fw & bm) == bm;
                        const uint64_t m = __ballot(cand);
                        if (cand) {
                            const uint32_t q = qcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                            qe[q] = make_uint4((uint32_t)hp, (uint32_t)(hp >> 32), tp, 0u);
                        }
                        qcount = (uint32_t)__builtin_amdgcn_readfirstlane((int)(qcount + (uint32_t)__popcll(m)));
                        word_load_async(ix.pre, index_pre_word(h, ix.pmask) << 2, fw);
                        hp = h;
                        tp = t;
                        continue;
                    }
                    bucket_wait(fb);
                    if (!RK_DBG(1)) {
                        const uint32_t fp = index_fp(hp);
                        // eight halfword compares (SDWA word selects, one VALU each).  A bucket whose overflow flag (bit 15 of slot
                        // 0) is set makes every window that lands in it a candidate (its slot-0 compare fails by design, the drain
                        // sorts it out): ~0.1 % of the windows.  Lanes without a window carry hp = 0 and looked up bucket(0)
                        // like everyone else: hash 0 is never in the index, so at worst the drain rejects a few of them.
                        // one ballot per compare (each IS the compare's lane mask), OR-ed on the scalar unit; inverse_ballot turns
                        // the mask back into the branch predicate without the v_cndmask + v_cmp round trip hipcc emits for
                        // ballot(a | b)
                        const uint64_t m = __builtin_amdgcn_ballot_w64((fb.x & 0xffffu) == fp) | __builtin_amdgcn_ballot_w64((fb.x >> 16) == fp) |
                                           __builtin_amdgcn_ballot_w64((fb.y & 0xffffu) == fp) | __builtin_amdgcn_ballot_w64((fb.y >> 16) == fp) |
                                           __builtin_amdgcn_ballot_w64((fb.z & 0xffffu) == fp) | __builtin_amdgcn_ballot_w64((fb.z >> 16) == fp) |
                                           __builtin_amdgcn_ballot_w64((fb.w & 0xffffu) == fp) | __builtin_amdgcn_ballot_w64((fb.w >> 16) == fp) |
                                           __builtin_amdgcn_ballot_w64((fb.x & IDX_OVF) != 0u);
                        const bool cand = __builtin_amdgcn_inverse_ballot_w64(m);
                        if (cand) {
                            const uint32_t q = qcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                            qe[q] = make_uint4((uint32_t)hp, (uint32_t)(hp >> 32), tp, 0u);
                        }
                        qcount = (uint32_t)__builtin_amdgcn_readfirstlane((int)(qcount + (uint32_t)__popcll(m)));
                    }
                    if RK_DBG(8) { if (h == 0x1234567ull) nzero[0] = 1; } else
                    bucket_load_async(ix.fpb, index_bucket(h, ix.bmask) << 4, fb); // lands while the next position is hashed
                    hp = h;
                    tp = t;
                }
                if (MODE == 1) break;
                // drain: every lane takes queued candidates
                wave_sync();
                const bool last = it > nIt;
                const uint32_t qn = last ? qcount : (qcount & ~(uint32_t)(WAVE - 1)); // mid-tile: whole waves of candidates only
                if (!RK_DBG(32)) drain_queue(qn);
                wave_sync();
                if (last) { qcount = 0; break; }
                const uint32_t rem = qcount - qn; // < 64 candidates move to the front of the queue
                uint4 ce = make_uint4(0u, 0u, 0u, 0u);
                if ((uint32_t)lane < rem) ce = qe[qn + lane];
                wave_sync();
                if ((uint32_t)lane < rem) qe[lane] = ce;
                qcount = rem;
                wave_sync();
            }
        }
        if (MODE == 1) { wait_bases(); continue; } // see the end of the loop body

        // ---- phase 2: 16 lanes per read, or 8 when the tile holds more than four (one pass for up to eight reads: with 16, the fifth
        //      read of a five-read tile was a pass of its own with 48 lanes idle) -----------------------
        {
            const int lsh = Tn > 4 ? 3 : 4, LPR = 1 << lsh; // wave-uniform
            const int g = lane >> lsh, sl = lane & (LPR - 1);
            for (int t = g; t < Tn; t += WAVE >> lsh) {
                uint32_t* ct = c16 + t * geo.cwords;
                const int nmins = (int)nwin[t] - (int)nzero[t];
                // bottom-S selection matters, or the hit multiset overflowed: exact answer comes from the general path
                const bool reroute = nmins > S || flags[t] != 0;
                const uint32_t bk = best[t];
                if (reroute || RK_DBG(2)) {
                    for (int w = sl; w < geo.cwords; w += LPR) ct[w] = 0;
                    if (sl == 0) reinterpret_cast<int4*>(out4)[r0 + t] = reroute ? make_int4(-2, 0, 0, 0) : make_int4(0, (int)(bk >> 16), 0, nmins);
                    continue;
                }
                // first max wins (rkmh.cpp:878); diff = max - best EARLIER score (untouched refs score 0; none => -1)
                const int max_id = bk ? (int)(0xFFFFu - (bk & 0xFFFFu)) : 0;
                const int max_shared = (int)(bk >> 16);
                int prev = max_id > 0 ? 0 : -1;
                if (geo.csparse) {
                    for (int w = sl; w < geo.cwords; w += LPR) {
                        const uint32_t x = ct[w];
                        const int r_ = (int)(x >> 11) - 1, cj = (int)(x & 0x7FFu);
                        if (x != 0u && r_ < max_id && cj > prev) prev = cj;
                    }
                } else
                for (int w = sl; (w << clg) < max_id; w += LPR) {
                    uint32_t x = ct[w];
                    for (uint32_t j = 0; j <= cper_m1; ++j) { // counters of references (w << clg) + j < max_id
                        const int cj = (int)(x & cmask);
                        x >>= cbits;
                        if ((int)((uint32_t)(w << clg) + j) < max_id && cj > prev) prev = cj;
                    }
                }
                prev = LPR == 16 ? row_max_i32(prev) : half_row_max_i32(prev);
                wave_sync();
                for (int w = sl; w < geo.cwords; w += LPR) ct[w] = 0;
                if (sl == 0) reinterpret_cast<int4*>(out4)[r0 + t] = make_int4(max_id, max_shared, max_shared - prev, nmins < geo.nmin_cap ? nmins : geo.nmin_cap);
            }
        }
        wait_bases();
    }
    if (MODE == 1 && geo.cs.tab) { // the compact count's last step and what is left in its queue
        uint32_t* cq = reinterpret_cast<uint32_t*>(qe);
        word_wait(cword);
        const uint32_t pbit = (cprev * 0x9E3779B1u) >> geo.cs.pre_shift;
        const bool pass = cprev != CS_EMPTY && ((cword >> (pbit & 31u)) & 1u) != 0u;
        const uint64_t m = __ballot(pass);
        if (pass) cq[ccount + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = cprev;
        ccount += (uint32_t)__popcll(m);
        wave_sync();
        for (uint32_t f = 0; f < ccount; f += WAVE) compact_drain(f, ccount - f < (uint32_t)WAVE ? ccount - f : (uint32_t)WAVE);
    }
}

// LDS budget of one single-wave workgroup for 6 waves per SIMD (24 per CU, 512-byte allocation granules).  The kernel
// saturates VALU issue at 6 waves/SIMD and loses ~11 % at 5 (measured), so tiles never grow past this.
constexpr size_t LDS_BUDGET_6_WAVES = 6656;

// The RKMH_* knobs (A/B runs, the geometry sweeps of the test suite) are read ONCE per process: a launch costs no getenv.
struct TileKnobs {
    int qcap = 0, c16 = 0, sparse = 0, dset = 0, dbg = 0, tile_t = 0, tpb = 0, xcd = -1, pre_masked = -1;
    TileKnobs() {
        auto num = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
        qcap = num("RKMH_TILE_QCAP", 0);
        c16 = num("RKMH_TILE_C16", 0);
        sparse = getenv("RKMH_TILE_SPARSE") ? (num("RKMH_TILE_SPARSE", 0) > 0 ? 1 : -1) : 0;
        dset = num("RKMH_TILE_DSET", 0);
        dbg = num("RKMH_DBG", 0);
        tile_t = num("RKMH_TILE_T", 0);
        tpb = num("RKMH_TILE_TPB", 0);
        xcd = num("RKMH_TILE_XCD", -1);
        pre_masked = num("RKMH_PRE_MASKED", -1);
    }
};
static const TileKnobs& knobs() { static const TileKnobs k; return k; }

static TileGeom make_geom(int maxlen, int nref, int expect_hits, int win_per_read, int win_total) {
    const TileKnobs& kn = knobs();
    TileGeom g;
    if (maxlen < 1) maxlen = 1;
    g.qcap = kn.qcap > 0 ? kn.qcap : 128;
    g.clg = win_total <= 255 ? 2 : 1;
    if (kn.c16 > 0) g.clg = 1; // A/B knob: always 16-bit counters
    g.cwords = (nref + (1 << g.clg) - 1) >> g.clg;
    g.csparse = 0;
    // Many references: a dense counter row per read would eat the LDS budget (and reference ids beyond 2048 would not
    // fit at all), so the row becomes a 128-entry map of the references the read actually hits.
    const int sparse_min = kn.sparse > 0 ? 1 : (kn.sparse < 0 ? (1 << 30) : 129); // dense rows up to 128 words (512 B) stay dense
    if (nref > 0 && g.cwords >= sparse_min) { g.csparse = 1; g.cwords = 128; }
    int ds = 64;
    while (ds < 3 * expect_hits && ds < 1024) ds <<= 1;
    if (kn.dset > 0) ds = kn.dset;
    g.dset = ds;
    g.dbg = kn.dbg;
    // Reads per tile: the hashing loop walks T * win_per_read windows 64 at a time, so T is chosen for the best fill of
    // its last step (150 bp, k=16: T=3 fills 89.7 %, T=4 93.1 %) among the sizes that keep the occupancy target and
    // the short prefetch (<= 3 dwords per lane).
    auto fits = [&](int T, size_t budget) {
        g.T = T; g.cap_bytes = T * maxlen;
        return T * maxlen <= PF_MAX * WAVE * 4 - 8 && tile_lds_bytes(g) <= budget;
    };
    if (win_per_read < 1) win_per_read = 1;
    double fill[17] = {0};
    int tmax = 1;
    double best_fill = -1.0;
    for (int T = 1; T <= 16; ++T) {
        if (T > 1 && (!fits(T, LDS_BUDGET_6_WAVES) || T * maxlen > 3 * WAVE * 4 - 8)) break;
        const int nw = T * win_per_read;
        fill[T] = (double)nw / (double)(((nw + WAVE - 1) / WAVE) * WAVE);
        if (fill[T] > best_fill) best_fill = fill[T];
        tmax = T;
    }
    // the smallest tile within 2.5 % of the best fill: beyond that, bigger tiles measured slower (150 bp: T=5 fills
    // 95.2 % against 93.1 % for T=4 and is 0.7 % slower; T=6 4 % slower)
    int best = 1;
    for (int T = 1; T <= tmax; ++T)
        if (fill[T] >= best_fill - 0.025) { best = T; break; }
    // A small tile that fills its steps well still pays the per-tile work (staging, phase 2, the prefetch ramp: about 2.2 steps'
    // worth, measured) once per FEW reads: k = 22 .. 30 at 150 bp (<= 128 windows per read: one read fills two steps to 99 %) ran
    // one read per tile at 1.67 ms per 1 M reads; four per tile take 1.05 ms.  So a fill-chosen tile below four reads is re-examined
    // with that cost, tiles beyond four reads carrying the ~4 % per read they measured slower by.
    if (best < 4) {
        auto cost = [&](int T) {
            const int nw = T * win_per_read;
            return ((double)((nw + WAVE - 1) / WAVE) + 2.2) / (double)T * (1.0 + 0.04 * (double)(T > 4 ? T - 4 : 0));
        };
        int alt = best;
        for (int T = best + 1; T <= tmax && T <= 8; ++T) if (cost(T) < cost(alt)) alt = T;
        if (cost(alt) < 0.95 * cost(best)) best = alt;
    }
    if (kn.tile_t > 0) best = kn.tile_t;
    if (best > 16) best = 16;
    if (best < 1) best = 1;
    while (best > 1 && best * maxlen > PF_MAX * WAVE * 4 - 8) --best;
    g.T = best;
    g.cap_bytes = best * maxlen;
    return g;
}

// reference ids must fit the 16 bits they get in the running (count, -ref) maximum; large panels count sparsely;
// one tile (>= 1 read of maxlen bytes) must fit the prefetch registers
bool classify_tile_supported(int nref, int maxlen) { return nref <= 16384 && maxlen <= PF_MAX * WAVE * 4 - 8; }

hipError_t launch_classify_tile(const uint8_t* bases, const uint32_t* offs, uint32_t nreads, const KsArr& ks, int S,
                                const RefIndex& ix, int32_t* counter, uint64_t slots, int min_occ, int mode,
                                int32_t* out4, const DevPolicy& pol, int maxlen, int expect_hits, hipStream_t st,
                                uint32_t slot_stride, int nmin_cap, const CompactSlots* compact) {
    if (nreads == 0) return hipSuccess;
    const TileKnobs& kn = knobs();
    int win_total = 0; // most windows any read of the batch can have (all k): bounds every per-reference count
    for (int j = 0; j < ks.n; ++j) win_total += num_windows(maxlen, ks.k[j], pol.drop_last_window);
    TileGeom geo = make_geom(maxlen, mode == 1 ? 0 : ix.nref, mode == 1 ? 0 : expect_hits,
                             num_windows(maxlen, ks.k[0], pol.drop_last_window), win_total);
    if (mode == 1) { geo.qcap = compact ? 48 : 0; geo.dset = 0; } // (compact count: 3 * 64 queued slots of 4 bytes)
    while (tile_lds_bytes(geo) > 20 * 1024 && geo.T > 1) { geo.T -= 1; geo.cap_bytes = geo.T * maxlen; } // >= 8 waves per CU
    const size_t lds = tile_lds_bytes(geo);
    const uint32_t ntiles = (nreads + (uint32_t)geo.T - 1) / (uint32_t)geo.T;
    // Short-lived waves: each single-wave workgroup walks a couple of tiles (the first one cold, the next prefetched)
    // and retires, so the hardware dispatcher keeps balancing the CUs.  Measured on MI355X with the current kernel
    // (T=4): 1 or 2 tiles per workgroup are within noise of each other, 3 is -1 %, 6 is -4 %, chip-resident persistent
    // waves were -17 % (tail imbalance); 2 keeps the cross-tile prefetch useful when the batch streams from HBM.
    const int tpb = kn.tpb > 0 ? kn.tpb : 2;
    geo.tpb = tpb;
    geo.xcd = kn.xcd >= 0 ? (kn.xcd != 0) : 1;
    geo.slots_m = slots ? ~0ull / slots : 0;
    geo.slot_stride = slot_stride;
    geo.nmin_cap = nmin_cap;
    if (compact) geo.cs = *compact; else memset(&geo.cs, 0, sizeof geo.cs);
    geo.magic_nw = num_windows(maxlen, ks.k[0], pol.drop_last_window);
    geo.magic = geo.magic_nw >= 2 ? 0xFFFFFFFFu / (uint32_t)geo.magic_nw + 1u : 0u;
    const bool k16 = (ks.n == 1 && ks.k[0] == 16);
    const int kmode = mode == 1 ? 1 : (counter ? 2 : 0);
    // The masked (-M) form waits for a random read of the keep bitmap in every step: it is bound by memory requests,
    // not by VALU issue, and the filter word is one more request per window (C2: 2.86 ms without, 3.10 ms with).  It pays
    // only once the bucket table itself has left L2.
    // (Round 2: the count is read as one keep bit per slot -- 25 MB instead of 800 MB -- which took this form from 2.86 to 2.58 ms;
    // pipelining that lookup one step ahead, in the filter-word form, was measured SLOWER, 2.89 ms: the pass is bound by the
    // rate of random accesses that miss the L2, about 5 * 10^10 per second here, not by latency or instructions.)
    bool pre_masked = ((size_t)ix.bmask + 1) * 16 > ((size_t)3 << 20);
    if (kn.pre_masked >= 0) pre_masked = kn.pre_masked != 0; // tests force either form
#define RK_LAUNCH_P(KT, MODE, FOLD, PF)                                                                                    \
    do {                                                                                                             \
        if (lds > 64 * 1024) {                                                                                       \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_classify_tile<KT, MODE, FOLD, PF>),       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
            if (e != hipSuccess) return e;                                                                           \
        }                                                                                                            \
        uint32_t grid = (ntiles + (uint32_t)tpb - 1) / (uint32_t)tpb;                                                \
        grid = (grid + 7u) & ~7u; /* whole rounds of the 8 XCDs: the virtual ids then cover [0, grid) exactly */      \
        hipLaunchKernelGGL((k_classify_tile<KT, MODE, FOLD, PF>), dim3(grid), dim3(WAVE), lds, st, bases, offs, nreads,   \
                           ks, S, ix, counter, slots, min_occ, out4, pol, geo);                                      \
    } while (0)
#define RK_LAUNCH(KT, MODE, FOLD)                                                                                    \
    do {                                                                                                             \
        if (geo.cap_bytes <= 2 * WAVE * 4 - 8) RK_LAUNCH_P(KT, MODE, FOLD, 2);                                       \
        else if (geo.cap_bytes <= 3 * WAVE * 4 - 8) RK_LAUNCH_P(KT, MODE, FOLD, 3);                                  \
        else RK_LAUNCH_P(KT, MODE, FOLD, 6);                                                                         \
    } while (0)
#define RK_LAUNCH_M(KT, FOLD)                                                                                        \
    do {                                                                                                             \
        if (kmode == 1) RK_LAUNCH(KT, 1, FOLD);                                                                      \
        else if (kmode == 0 && !ix.pre) RK_LAUNCH(KT, 0, FOLD);                                                      \
        else if (kmode == 0) RK_LAUNCH(KT, 3, FOLD);                                                                 \
        else if (!ix.pre || !pre_masked) RK_LAUNCH(KT, 2, FOLD);                                                     \
        else RK_LAUNCH(KT, 4, FOLD);                                                                                 \
    } while (0)
#ifdef RK_TILE_ONLY_K // experiment builds : one compile-time k, seconds to compile
    if (ks.n != 1 || ks.k[0] != RK_TILE_ONLY_K) return hipErrorInvalidValue;
    RK_LAUNCH_M(RK_TILE_ONLY_K, -1);
    return hipGetLastError();
#endif
    // single k of 12, 20 (the reference's other documented settings), 21 or 31: window length known at compile time, runtime fold
    if (ks.n == 1 && ks.k[0] == 12) RK_LAUNCH_M(12, -1);
    else if (ks.n == 1 && ks.k[0] == 20) RK_LAUNCH_M(20, -1);
    else if (ks.n == 1 && ks.k[0] == 21) RK_LAUNCH_M(21, -1);   // Mash / sourmash defaults
    else if (ks.n == 1 && ks.k[0] == 31) RK_LAUNCH_M(31, -1);
    else if (!k16) RK_LAUNCH_M(0, -1);        // any k / several k: runtime fold
    else if (pol.fold == 0) RK_LAUNCH_M(16, 0);
    else if (pol.fold == 1) RK_LAUNCH_M(16, 1);
    else RK_LAUNCH_M(16, 2);
#undef RK_LAUNCH_M
#undef RK_LAUNCH
#undef RK_LAUNCH_P
    return hipGetLastError();
}

} // namespace rk
