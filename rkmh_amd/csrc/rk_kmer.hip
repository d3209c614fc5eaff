// rk_kmer.hip -- the k-mer-space form of the fused per-read kernel (gfx950, wave64): the shipped hot loop for a single k from 8 to 16.
//
// Replaces the body of main_stream's read loop, /root/reference/src/rkmh.cpp:856-888
//     to_upper -> calc_hashes -> minhashes -> R x hash_intersection_size -> argmax/diff
// for every read whose non-zero hashes all fit the sketch (other reads are flagged max_id = -2 and rerouted by the host through
// k_hash_tiles + k_sort_intersect, exactly as with k_classify_tile).  NO window is hashed here: when the references were set,
// k_enum_kmers hashed the whole 4^k k-mer universe and recorded every k-mer whose canonical hash is a sketch hash (or 0), so "this
// window's k-mer is not among them" PROVES that its hash is non-zero and in no sketch.  What is new
// against the MODE_ 5 instantiations of k_classify_tile (rk_classify.hip), which this kernel supersedes:
//   * phase 0 is ONE global_load_dwordx4 per lane (16 contiguous bases, 1 KB per wave instruction), packed to one dword of 2-bit
//     codes in registers (a multiply gathers the four codes of a dword) and stored with one conflict-free ds_write_b32.  There is
//     no upper-cased byte image, no reverse-complement image and no byte store.
//   * a lane examines FOUR consecutive windows (one 8-byte LDS read of the k+3-base super-window); the filter is addressed by the
//     forward k-mers (both orientations of every found k-mer are entered), one 16-byte sector per group chosen by the (k-3)-mer the
//     four windows share: one L2 request per four windows and no per-window complement / minimum.
//   * tiles of unequal reads and tiles with non-ACGT bases stay in k-mer space too (a window holding an invalid base hashes to 0 by
//     definition -- counted from a validity bitmap -- and read boundaries only change the group -> read mapping): the kernel
//     contains no murmur code at all.
//   * the drain canonicalises a candidate (bit reverse), resolves it by k-mer in the exact cuckoo map (kmap), and uses the k-mer
//     itself as identity in the per-read hit set; two drain rounds' lookups are in flight together.
// Work decomposition as before: ONE WAVE = one tile of T consecutive reads, a workgroup is a single wave (no barriers).
#include "rk_kernels.hpp"

#include <cstdlib>

namespace rk {

namespace {

constexpr int KW = 64;
constexpr int KM_MAX_T = 8;        // reads per tile (phase 2 takes eight reads in one pass; the group -> read search is unrolled for it)
constexpr int KM_MQ = 32;          // deferred multi-posting hits per tile (more are walked by their own lane)
constexpr int KM_CH = 4;           // steps (64 groups = 256 windows each) whose filter sectors are requested together
constexpr int KM_QCAP = 64 + 4 * KW; // candidate queue entries (4 bytes each): a step adds at most 256 to a remainder of < 64
#ifndef RK_KMER_WAVES
#define RK_KMER_WAVES 6
#endif

struct KmerGeom {
    int32_t T;         // reads per tile
    int32_t cap_bytes; // largest tile (bytes) that fits the staged quads: NQ * 1024 - 15
    int32_t cwords, clg, csparse; // per-read reference counters, as in k_classify_tile
    int32_t dset;      // slots of the per-read hit set (power of two)
    int32_t tpb, xcd;
    int32_t L;         // hinted read length ...
    int32_t gpr;       // ... its groups per read = ceil(windows / 4) ...
    uint32_t magic;    // ... and ceil(2^32 / gpr): the group -> read division of tiles made of such reads
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) km_pair1 { uint32_t x, y; };

__host__ __device__ inline int km_pk_dwords(int nq) { return nq * 64 + 2; }
__host__ __device__ inline int km_inv_dwords(int nq) { return nq * 32 + 2; }
__host__ __device__ inline size_t km_lds_bytes(const KmerGeom& g, int nq) {
    return ((size_t)km_pk_dwords(nq) + km_inv_dwords(nq) + (size_t)KM_QCAP + 2 * KM_MQ + 4 * (KM_MAX_T + 1) + 3 * KM_MAX_T +
            (size_t)g.T * (size_t)(g.cwords + g.dset)) * 4;
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int row_max_i32(int v) {
    int t;
    t = dpp_i32<0xB1>(v); v = t > v ? t : v;
    t = dpp_i32<0x4E>(v); v = t > v ? t : v;
    t = dpp_i32<0x141>(v); v = t > v ? t : v;
    t = dpp_i32<0x140>(v); v = t > v ? t : v;
    return v;
}
__device__ __forceinline__ int half_row_max_i32(int v) {
    int t;
    t = dpp_i32<0xB1>(v); v = t > v ? t : v;
    t = dpp_i32<0x4E>(v); v = t > v ? t : v;
    t = dpp_i32<0x141>(v); v = t > v ? t : v;
    return v;
}

// four ASCII bases (any case) -> their 2-bit codes (A=0 C=1 T=2 G=3) gathered in bits 24..31; mism collects the bytes that are
// not one of ACGTacgt (mkmh's to_upper only moves a..z onto A..Z among them, so the test on the case-folded byte is exact)
__device__ __forceinline__ uint32_t km_pack4_hi(uint32_t x, uint32_t& mism) {
    const uint32_t c = (x >> 1) & 0x03030303u;
    mism |= (x & 0xDFDFDFDFu) ^ __builtin_amdgcn_perm(0x47544341u, 0x47544341u, c); // code -> "ACTG"
    // byte i of c moves to bits 24 + 2i: multiplier 2^24 + 2^18 + 2^12 + 2^6 (no two partial products overlap below bit 32)
    return c * 0x01041040u;
}
// 16 bases in four dwords -> one dword of 2-bit codes (base i in bits [2i, 2i+2))
__device__ __forceinline__ uint32_t km_pack16(const u32x4& v, uint32_t& mism) {
    const uint32_t m0 = km_pack4_hi(v.x, mism), m1 = km_pack4_hi(v.y, mism), m2 = km_pack4_hi(v.z, mism), m3 = km_pack4_hi(v.w, mism);
    // v_perm_b32: selector bytes 0-3 pick from the second operand, 4-7 from the first, 0x0c = zero
    const uint32_t lo = __builtin_amdgcn_perm(m1, m0, 0x0c0c0703u); // byte 0 = m0 byte 3, byte 1 = m1 byte 3
    const uint32_t hi = __builtin_amdgcn_perm(m3, m2, 0x07030c0cu); // byte 2 = m2 byte 3, byte 3 = m3 byte 3
    return lo | hi;
}
// bit q set <=> byte q of m is non-zero
__device__ __forceinline__ uint32_t km_nonzero4(uint32_t m) {
    const uint32_t y = (m | ((m & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
    return ((y >> 7) * 0x01020408u) >> 24;
}

template <int KT, int NQ>
__global__ __launch_bounds__(KW, RK_KMER_WAVES) void k_classify_kmer(const uint8_t* __restrict__ bases, const uint32_t* __restrict__ offs,
                                                                     uint32_t nreads, int S, RefIndex ix, int32_t* __restrict__ out4,
                                                                     DevPolicy pol, KmerGeom geo) {
    static_assert(KT >= 4 && KT <= 16, "a k-mer packs into 32 bits; the core is the (k-3)-mer");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr uint32_t KMASK = KT == 16 ? 0xffffffffu : ((1u << (2 * KT)) - 1u);
    constexpr uint32_t CMASK = (1u << (2 * (KT - 3))) - 1u;
    constexpr uint32_t KBITS = (1u << KT) - 1u; // KT validity bits
    const int T = geo.T;
    constexpr uint32_t QCAP = (uint32_t)KM_QCAP;
    const uint32_t DS = (uint32_t)geo.dset;
    uint32_t* pk = smem;                                          // packed 2-bit image of the staged quads (position P = byte P of the quads)
    uint32_t* inv = pk + km_pk_dwords(NQ);                        // bit P set <=> base P is not ACGT (built only for tiles that hold one)
    uint32_t* q = inv + km_inv_dwords(NQ);                        // candidate queue: P | read << 11 | (window within the read) << 15
    uint32_t* mq = q + QCAP;                                      // [KM_MQ][2] deferred hits with a posting list: read, list offset
    uint4* rinfo = reinterpret_cast<uint4*>(mq + 2 * KM_MQ);      // [T+1] {start position, windows, first group, -} of read t (16-byte aligned: all sizes above are even... see launcher)
    uint32_t* nzero = reinterpret_cast<uint32_t*>(rinfo + (KM_MAX_T + 1)); // [T] zero hashes of read t
    uint32_t* best = nzero + KM_MAX_T;                            // [T] max over increments of (count << 16 | 0xFFFF - ref)
    uint32_t* flags = best + KM_MAX_T;                            // [T] read must take the general path
    uint32_t* cnt = flags + KM_MAX_T;                             // [T][cwords] packed per-reference counters
    uint32_t* dset = cnt + T * geo.cwords;                        // [T][DS] k-mers (+1) the read has hit
    const int lane = threadIdx.x;
    const uint32_t clg = (uint32_t)geo.clg, cper_m1 = (1u << clg) - 1u, cbits = 32u >> clg, cmask = (1u << cbits) - 1u;

    for (int i = lane; i < T * geo.cwords; i += KW) cnt[i] = 0; // re-zeroed by phase 2 after use
    const uint32_t ntiles = (nreads + (uint32_t)T - 1) / (uint32_t)T;
    const uintptr_t gb = reinterpret_cast<uintptr_t>(bases);
    // readable byte range of the batch: [bases, bases + offs[nreads] + 4) (the ABI asks for 4 bytes of slack), whole dwords
    const uintptr_t safe_hi = (gb + (uintptr_t)offs[nreads] + 4u) & ~(uintptr_t)3;

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
    // The tile's bases as 16-byte quads: quad i = bytes [16 i, 16 i + 16) from the 16-byte boundary at or below the tile's first
    // base; lane l holds quads l, 64 + l, ...  A quad that reaches outside the batch's readable range (only possible in the first
    // and the last tile) is fetched dword by dword with the unreadable dwords taken as 0.
    u32x4 pf[NQ];
    auto fetch_tile = [&](uint32_t a_, uint32_t b_) {
        const uintptr_t first = gb + a_;
        const uintptr_t base16 = first & ~(uintptr_t)15;
        const uint32_t nbytes = (uint32_t)(first - base16) + (b_ - a_);
        uint32_t nq = (nbytes + 15u) >> 4;
        if (nbytes > (uint32_t)(NQ * 1024)) nq = 0; // oversized tile: rerouted, nothing to stage
        const bool safe = base16 >= gb && base16 + 16u * (uintptr_t)nq <= safe_hi; // wave-uniform
#pragma unroll
        for (int r = 0; r < NQ; ++r) {
            const uint32_t qi = (uint32_t)r * KW + (uint32_t)lane;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (qi < nq) {
                const uintptr_t qa = base16 + 16u * (uintptr_t)qi;
                if (safe) v = *reinterpret_cast<const u32x4*>(qa);
                else {
                    uint32_t d[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uintptr_t da = qa + 4u * (uintptr_t)j;
                        d[j] = (da >= gb && da + 4u <= safe_hi) ? *reinterpret_cast<const uint32_t*>(da) : 0u;
                    }
                    v.x = d[0]; v.y = d[1]; v.z = d[2]; v.w = d[3];
                }
            }
            pf[r] = v;
        }
    };

    uint32_t vb = blockIdx.x;
    if (geo.xcd) { const uint32_t per = (gridDim.x + 7u) >> 3; vb = (blockIdx.x & 7u) * per + (blockIdx.x >> 3); }
    const uint32_t tile0 = vb * (uint32_t)geo.tpb;
    const uint32_t tile_end = (tile0 + (uint32_t)geo.tpb) < ntiles ? (tile0 + (uint32_t)geo.tpb) : ntiles;
    uint32_t cur_a = 0, cur_b = 0, cur_o = 0;
#pragma unroll
    for (int r = 0; r < NQ; ++r) pf[r] = u32x4{0u, 0u, 0u, 0u};
    if (tile0 < ntiles) { load_offsets(tile0, cur_a, cur_b, cur_o); fetch_tile(cur_a, cur_b); }

    for (uint32_t tile = tile0; tile < tile_end; ++tile) {
        const uint32_t r0 = tile * (uint32_t)T;
        const uint32_t ta = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur_a);
        const uint32_t tb = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur_b);
        const uint32_t ntile = tile + 1u < tile_end ? tile + 1u : ntiles;
        uint32_t nxt_a = 0, nxt_b = 0, nxt_o = 0;
        if (ntile < ntiles) load_offsets(ntile, nxt_a, nxt_b, nxt_o);
        wave_sync(); // previous tile fully consumed
        const uint32_t mis = (uint32_t)((gb + ta) & 15u);       // position of the tile's first base inside quad 0
        const uint32_t nbytes = mis + (tb - ta);
        const bool oversized = nbytes > (uint32_t)(NQ * 1024);
        if (oversized && lane < tile_reads(tile)) reinterpret_cast<int4*>(out4)[r0 + lane] = make_int4(-2, 0, 0, 0);
        const int Tn = oversized ? 0 : tile_reads(tile);
        const uint32_t nq = oversized ? 0u : (nbytes + 15u) >> 4;

        // ---- phase 0: per-read bookkeeping (lane t = read t), packed image, validity ballot ---------------------------------
        const uint32_t o_next = (uint32_t)__shfl_down((int)cur_o, 1);
        const uint32_t len = o_next - cur_o;
        uint32_t nw = 0, ng = 0;
        if (lane < Tn) { nw = (uint32_t)num_windows((int)len, KT, pol.drop_last_window); ng = (nw + 3u) >> 2; }
        uint32_t gs = ng; // inclusive prefix of the group counts over the tile's reads (lanes 0..7)
        { uint32_t t1 = (uint32_t)__shfl_up((int)gs, 1); if (lane >= 1) gs += t1;
          t1 = (uint32_t)__shfl_up((int)gs, 2); if (lane >= 2) gs += t1;
          t1 = (uint32_t)__shfl_up((int)gs, 4); if (lane >= 4) gs += t1; }
        const uint32_t NG = (uint32_t)__builtin_amdgcn_readlane((int)gs, KM_MAX_T - 1); // groups of the tile
        if (lane <= KM_MAX_T) { // entries past the tile's last read: no windows, first group = NG
            rinfo[lane] = make_uint4(cur_o - ta + mis, nw, gs - ng, 0u);
            if (lane < Tn) {
                nzero[lane] = 0; best[lane] = 0;
                // more windows than a packed counter can count (only possible when the caller's length hint was too small)
                flags[lane] = nw > (geo.csparse ? 0x7FFu : cmask) ? 1u : 0u;
            }
        }
        const uint32_t ulen = (uint32_t)__builtin_amdgcn_readfirstlane((int)len);
        const bool same_len = __ballot(lane < Tn && len != ulen) == 0ull;
        for (uint32_t i = lane; i < (uint32_t)Tn * DS; i += KW) dset[i] = 0;
        uint32_t mism = 0;
#pragma unroll
        for (int r = 0; r < NQ; ++r) {
            const uint32_t qi = (uint32_t)r * KW + (uint32_t)lane;
            uint32_t m = 0;
            pk[qi] = km_pack16(pf[r], m);
            if (qi < nq) mism |= m;
        }
        if (lane < 2) pk[NQ * KW + lane] = 0;
        const bool has_invalid = __ballot(mism != 0u) != 0ull;
        if (has_invalid) { // rare: 16 validity bits per quad (the quads are still in registers)
#pragma unroll
            for (int r = 0; r < NQ; ++r) {
                const uint32_t qi = (uint32_t)r * KW + (uint32_t)lane;
                uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
                km_pack4_hi(pf[r].x, m0); km_pack4_hi(pf[r].y, m1); km_pack4_hi(pf[r].z, m2); km_pack4_hi(pf[r].w, m3);
                const uint32_t b16 = km_nonzero4(m0) | (km_nonzero4(m1) << 4) | (km_nonzero4(m2) << 8) | (km_nonzero4(m3) << 12);
                reinterpret_cast<uint16_t*>(inv)[qi] = (uint16_t)(qi < nq ? b16 : 0u);
            }
            if (lane < 2) inv[NQ * 32 + lane] = 0;
        }
        cur_a = nxt_a; cur_b = nxt_b; cur_o = nxt_o;
        wave_sync();

        // group -> read mapping: tiles of equally long reads divide by a magic constant, the others search the group starts
        const uint32_t nw_u = (uint32_t)num_windows((int)ulen, KT, pol.drop_last_window), gpr_u = (nw_u + 3u) >> 2;
        const bool uniform = same_len && gpr_u >= 2u; // >= 2: the magic division needs a divisor > 1
        uint32_t magic = 0;
        if (uniform) magic = ulen == (uint32_t)geo.L ? geo.magic : (uint32_t)__builtin_amdgcn_readfirstlane((int)(0xFFFFFFFFu / gpr_u + 1u));
        // +1 for reference `ref` of read t; the monotone counters make (max_shared, first max_id) a running atomicMax
        auto add_posting = [&](uint32_t t, uint32_t ref) {
            uint32_t c;
            if (geo.csparse) { // wave-uniform: large panels keep (ref, count) pairs of the references a read really hits
                uint32_t* row = cnt + t * (uint32_t)geo.cwords;
                const uint32_t M1 = (uint32_t)geo.cwords - 1u, key = ref + 1u;
                uint32_t idx = ((ref * 0x9E3779B1u) >> 16) & M1, probe = 0;
                c = 0;
                for (; probe <= M1; ++probe) {
                    const uint32_t old = atomicCAS(&row[idx], 0u, (key << 11) | 1u);
                    if (old == 0u) { c = 1u; break; }
                    if ((old >> 11) == key) { c = (atomicAdd(&row[idx], 1u) & 0x7FFu) + 1u; break; }
                    idx = (idx + 1u) & M1;
                }
                if (probe > M1) { flags[t] = 1; return; } // the read hits more references than the map holds: general path
            } else {
                const uint32_t sh = (ref & cper_m1) * cbits;
                const uint32_t old = atomicAdd(&cnt[t * (uint32_t)geo.cwords + (ref >> clg)], 1u << sh);
                c = ((old >> sh) & cmask) + 1u;
            }
            atomicMax(&best[t], (c << 16) | (0xFFFFu - ref));
        };
        auto walk_list = [&](uint32_t t, uint32_t rank, uint32_t off) { // the postings of a list this occurrence counts for
            const uint32_t n = ix.post[off];
            for (uint32_t c = 0; c < n; ++c) if (rank < ix.post[off + 2 + 2 * c]) add_posting(t, ix.post[off + 1 + 2 * c]);
        };
        uint32_t mqn = 0; // deferred list hits (wave-uniform)
        auto process_mq = [&]() { // 16 lanes walk one hit's posting list, 4 hits at a time
            wave_sync();
            const uint32_t g = (uint32_t)lane >> 4, sl = (uint32_t)lane & 15u;
            for (uint32_t j = g; j < mqn; j += KW / 16) {
                const uint32_t tr = mq[2 * j], off = mq[2 * j + 1];
                const uint32_t n = ix.post[off];
                for (uint32_t c = sl; c < n; c += 16) if ((tr >> 8) < ix.post[off + 2 + 2 * c]) add_posting(tr & 0xFFu, ix.post[off + 1 + 2 * c]);
            }
            wave_sync();
            mqn = 0;
        };
        // one candidate of the queue: canonical k-mer, its read, its two buckets of the exact map (loads in flight on return)
        struct Cand { uint32_t key, t; bool ok; uint4 c1, c2; };
        auto lookup = [&](uint32_t e, uint32_t qn) -> Cand {
            Cand c;
            c.key = 0; c.t = 0; c.ok = false;
            c.c1 = make_uint4(KMAP_EMPTY, 0u, KMAP_EMPTY, 0u); c.c2 = c.c1;
            if (e < qn) {
                const uint32_t ent = q[e];
                const uint32_t P = ent & 2047u, t = (ent >> 11) & 15u, o = ent >> 15;
                const km_pair1 w = *reinterpret_cast<const km_pair1*>(reinterpret_cast<const uint8_t*>(pk) + (P >> 2));
                const uint32_t x = __builtin_amdgcn_alignbit(w.y, w.x, (P & 3u) << 1) & KMASK;
                const uint32_t r = packed_revcomp(x, KT);
                c.key = x < r ? x : r;
                c.t = t;
                c.ok = o < rinfo[t].y; // the last group of a read may reach past its last window
                c.c1 = ix.kmap[kmap_cell1(c.key, ix.kmap_m)];
                c.c2 = ix.kmap[kmap_cell2(c.key, ix.kmap_m)];
            }
            return c;
        };
        auto apply = [&](const Cand& c) {
            uint32_t val = KMAP_EMPTY; // KMAP_EMPTY is no index value (bit 31 set => postings offset < 2^31 - 1)
            uint32_t rank = 0;
            if (c.c1.x == c.key) val = c.c1.y;
            else if (c.c1.z == c.key) val = c.c1.w;
            else if (c.c2.x == c.key) val = c.c2.y;
            else if (c.c2.z == c.key) val = c.c2.w;
            bool multi = false;
            if (c.ok && val != KMAP_EMPTY) { // else: a false positive of the bit filter, or a window past its read's last
                if (val == KMAP_ZERO) atomicAdd(&nzero[c.t], 1u); // a k-mer whose canonical hash is 0
                else {
                    // hit MULTISET of the read: the canonical k-mer (+1: never 0xFFFFFFFF) is the key's identity, every occurrence
                    // adds one more entry, and the entries passed on the way to the free slot are this occurrence's rank.  The merge
                    // of rkmh.cpp:869 counts min(occurrences in the read, multiplicity in the sketch): occurrence `rank` of a k-mer
                    // counts for a posting iff rank < its multiplicity.
                    uint32_t* ds = dset + c.t * DS;
                    const uint32_t id = c.key + 1u;
                    uint32_t idx = (c.key * 0x9E3779B1u) >> (32u - (uint32_t)__builtin_ctz(DS));
                    bool done = false;
                    for (uint32_t probe = 0; probe < DS && !done; ++probe) {
                        const uint32_t old = atomicCAS(&ds[idx], 0u, id);
                        if (old == 0u) done = true;
                        else { rank += old == id ? 1u : 0u; idx = (idx + 1u) & (DS - 1u); }
                    }
                    if (!done) flags[c.t] = 1; // more hits than the set holds: general path
                    else if (!(val >> 31)) { // one posting (with multiplicity) or two single postings, stored inline
                        const bool two = ((val >> 29) & 3u) != 0u;
                        if (rank < (two ? 1u : ((val >> 20) & 0x1FFu))) {
                            add_posting(c.t, val & (two ? 0x7FFu : 0xFFFFFu));
                            if (two) add_posting(c.t, (val >> 11) & 0x7FFu);
                        }
                    } else multi = true;
                }
            }
            const uint64_t mm = __ballot(multi);
            if (mm) { // hits with a posting list are deferred to the end of the drain; the few that find the list full walk their own
                const uint32_t j = mqn + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
                if (multi) {
                    if (j < (uint32_t)KM_MQ) { mq[2 * j] = c.t | (rank << 8); mq[2 * j + 1] = val & 0x7fffffffu; }
                    else walk_list(c.t, rank, val & 0x7fffffffu);
                }
                const uint32_t tot = mqn + (uint32_t)__popcll(mm);
                mqn = tot < (uint32_t)KM_MQ ? tot : (uint32_t)KM_MQ;
            }
        };
        auto drain = [&](uint32_t qn) {
            for (uint32_t e0 = 0; e0 < qn; e0 += 2 * KW) { // two rounds of lookups in flight
                const bool two = e0 + KW < qn;
                const Cand a = lookup(e0 + (uint32_t)lane, qn);
                Cand b = a;
                if (two) b = lookup(e0 + KW + (uint32_t)lane, qn);
                apply(a);
                if (two) apply(b);
            }
            if (mqn) process_mq();
        };

        // ---- window phase: 64 groups (256 windows) per step, KM_CH steps' filter sectors requested together ---------------------
        const uint32_t nsteps = (NG + KW - 1) / KW;
        const uint32_t rs0 = mis;
        uint32_t qcount = 0;
        uint32_t step = 0;
        for (;;) {
            uint32_t wl[KM_CH], wh[KM_CH], e0v[KM_CH];
            u32x4 fw[KM_CH];
#pragma unroll
            for (int s = 0; s < KM_CH; ++s) {
                wl[s] = 0; wh[s] = 0; e0v[s] = 0xFFFFFFFFu; fw[s] = u32x4{0u, 0u, 0u, 0u};
                if (step + (uint32_t)s < nsteps) { // wave-uniform
                    const uint32_t G = (step + (uint32_t)s) * KW + (uint32_t)lane;
                    if (G < NG) {
                        uint32_t t, g, rs;
                        if (uniform) { // wave-uniform
                            t = __umulhi(G, magic);
                            g = G - t * gpr_u;
                            rs = rs0 + t * ulen;
                        } else {
                            t = 0; // the last read whose first group is <= G (rinfo[i].z of reads past the tile's last is NG)
#pragma unroll
                            for (int i = 1; i < KM_MAX_T; ++i) t += G >= rinfo[i].z ? 1u : 0u;
                            const uint4 ri = rinfo[t];
                            g = G - ri.z;
                            rs = ri.x;
                        }
                        const uint32_t P0 = rs + 4u * g;
                        const km_pair1 w = *reinterpret_cast<const km_pair1*>(reinterpret_cast<const uint8_t*>(pk) + (P0 >> 2));
                        const uint32_t sh = (P0 & 3u) << 1;
                        wl[s] = __builtin_amdgcn_alignbit(w.y, w.x, sh);
                        wh[s] = w.y >> sh;
                        e0v[s] = P0 | (t << 11) | (g << 17);
                        const uint32_t core = (__builtin_amdgcn_alignbit(wh[s], wl[s], 6)) & CMASK;
                        fw[s] = reinterpret_cast<const u32x4*>(ix.kf4)[kf4_sector(core, ix.kf4_lg)];
                    }
                }
            }
            bool stop = false;
            uint32_t done = 0;
#pragma unroll
            for (int s = 0; s < KM_CH; ++s) {
                if (step + (uint32_t)s < nsteps && !stop) { // wave-uniform
                    if (qcount + 4u * KW > QCAP) stop = true;
                    else {
                        ++done;
                        { // lanes without a group carry an all-zero sector: none of their windows passes, and every lane takes part
                          // in the ballots that advance the (wave-uniform) queue length
                            const bool active = e0v[s] != 0xFFFFFFFFu;
                            uint32_t ib = 0, nz = 0, nwt = 0;
                            if (has_invalid && active) { // has_invalid: wave-uniform, rare
                                const uint32_t P0 = e0v[s] & 2047u;
                                ib = __builtin_amdgcn_alignbit(inv[(P0 >> 5) + 1], inv[P0 >> 5], P0 & 31u);
                                nwt = rinfo[(e0v[s] >> 11) & 15u].y;
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const uint32_t x = (j == 0 ? wl[s] : __builtin_amdgcn_alignbit(wh[s], wl[s], 2 * j)) & KMASK;
                                const uint32_t h = kf4_h(x);
                                const uint32_t f = j == 0 ? fw[s].x : (j == 1 ? fw[s].y : (j == 2 ? fw[s].z : fw[s].w));
                                bool cand = (((f >> (h >> 27)) & (f >> ((h >> 22) & 31u))) & 1u) != 0u;
                                if (has_invalid) {
                                    if (active && ((ib >> j) & KBITS) != 0u) { // a window holding a non-ACGT base hashes to 0 (if it is a window of the read)
                                        cand = false;
                                        if ((e0v[s] >> 15) + (uint32_t)j < nwt) ++nz;
                                    }
                                }
                                const uint64_t m = __ballot(cand);
                                if (cand) {
                                    const uint32_t qi = qcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                                    q[qi] = e0v[s] + (uint32_t)j * 0x8001u;
                                }
                                qcount = (uint32_t)__builtin_amdgcn_readfirstlane((int)(qcount + (uint32_t)__popcll(m)));
                            }
                            if (has_invalid && nz) atomicAdd(&nzero[(e0v[s] >> 11) & 15u], nz);
                        }
                    }
                }
            }
            step += done;
            const bool last = step >= nsteps;
            // the next tile's bases are requested once this tile's filter traffic is over: they land during the drain and phase 2
            if (last && ntile < ntiles) fetch_tile(cur_a, cur_b);
            wave_sync();
            const uint32_t qn = last ? qcount : (qcount & ~(uint32_t)(KW - 1)); // mid-tile: whole waves of candidates only
            drain(qn);
            wave_sync();
            if (last) break;
            const uint32_t rem = qcount - qn; // < 64 candidates move to the front of the queue
            uint32_t ce = 0;
            if ((uint32_t)lane < rem) ce = q[qn + lane];
            wave_sync();
            if ((uint32_t)lane < rem) q[lane] = ce;
            qcount = rem;
            wave_sync();
        }

        // ---- phase 2: 16 lanes per read, or 8 when the tile holds more than four ------------------------------------------------
        {
            const int lsh = Tn > 4 ? 3 : 4, LPR = 1 << lsh; // wave-uniform
            const int g = lane >> lsh, sl = lane & (LPR - 1);
            for (int t = g; t < Tn; t += KW >> lsh) {
                uint32_t* ct = cnt + t * geo.cwords;
                const int nmins = (int)rinfo[t].y - (int)nzero[t];
                // bottom-S selection matters, or the hit set overflowed: exact answer comes from the general path
                const bool reroute = nmins > S || flags[t] != 0;
                const uint32_t bk = best[t];
                if (reroute) {
                    for (int w = sl; w < geo.cwords; w += LPR) ct[w] = 0;
                    if (sl == 0) reinterpret_cast<int4*>(out4)[r0 + t] = make_int4(-2, 0, 0, 0);
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
                if (sl == 0) reinterpret_cast<int4*>(out4)[r0 + t] = make_int4(max_id, max_shared, max_shared - prev, nmins);
            }
        }
    }
}

// LDS budget of one single-wave workgroup for 8 waves per SIMD (32 per CU of 160 KB)
constexpr size_t KM_LDS_BUDGET = 5120;

KmerGeom make_kmer_geom(int maxlen, int nref, int expect_hits, int nw_per_read, int win_total, int nq) {
    KmerGeom g;
    if (maxlen < 1) maxlen = 1;
    g.cap_bytes = nq * 1024 - 15;
    g.clg = win_total <= 255 ? 2 : 1;
    g.cwords = (nref + (1 << g.clg) - 1) >> g.clg;
    g.csparse = 0;
    if (nref > 0 && g.cwords >= 129) { g.csparse = 1; g.cwords = 128; }
    int ds = 64;
    while (ds < 3 * expect_hits && ds < 1024) ds <<= 1;
    g.dset = ds;
    int T = g.cap_bytes / maxlen;
    if (T > KM_MAX_T) T = KM_MAX_T;
    if (T < 1) T = 1;
    static const int forced_t = getenv("RKMH_KMER_T") ? atoi(getenv("RKMH_KMER_T")) : 0;
    if (forced_t > 0 && forced_t < T) T = forced_t;
    g.T = T;
    while (g.T > 1 && km_lds_bytes(g, nq) > KM_LDS_BUDGET) g.T -= 1;
    g.tpb = 2; g.xcd = 1;
    g.L = maxlen;
    g.gpr = (nw_per_read + 3) >> 2;
    g.magic = g.gpr >= 2 ? 0xFFFFFFFFu / (uint32_t)g.gpr + 1u : 0u;
    return g;
}

} // namespace

// reads of up to 2 * 1024 - 15 bytes (two quads per lane), panels the 16-bit reference field of the running maximum can name
bool classify_kmer_supported(int nref, int maxlen, int k) {
    return nref <= 16384 && maxlen <= 2 * 1024 - 15 && k >= KPRE_MIN_K && k <= 16;
}

hipError_t launch_classify_kmer(const uint8_t* bases, const uint32_t* offs, uint32_t nreads, int k, int S, const RefIndex& ix,
                                int32_t* out4, const DevPolicy& pol, int maxlen, int expect_hits, hipStream_t st) {
    if (nreads == 0) return hipSuccess;
    const int nq = maxlen <= 1024 - 15 ? 1 : 2;
    const int nw = num_windows(maxlen, k, pol.drop_last_window);
    KmerGeom geo = make_kmer_geom(maxlen, ix.nref, expect_hits, nw, nw, nq);
    static const int tpb_env = getenv("RKMH_TILE_TPB") ? atoi(getenv("RKMH_TILE_TPB")) : 0;
    static const int xcd_env = getenv("RKMH_TILE_XCD") ? atoi(getenv("RKMH_TILE_XCD")) : -1;
    if (tpb_env > 0) geo.tpb = tpb_env;
    if (xcd_env >= 0) geo.xcd = xcd_env != 0;
    const size_t lds = km_lds_bytes(geo, nq);
    const uint32_t ntiles = (nreads + (uint32_t)geo.T - 1) / (uint32_t)geo.T;
    uint32_t grid = (ntiles + (uint32_t)geo.tpb - 1) / (uint32_t)geo.tpb;
    grid = (grid + 7u) & ~7u; // whole rounds of the 8 XCDs: the virtual ids then cover [0, grid) exactly
#define RK_KM_LAUNCH(KT, NQ)                                                                                                   \
    do {                                                                                                                       \
        if (lds > 64 * 1024) {                                                                                                 \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_classify_kmer<KT, NQ>),                         \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                          \
            if (e != hipSuccess) return e;                                                                                     \
        }                                                                                                                      \
        hipLaunchKernelGGL((k_classify_kmer<KT, NQ>), dim3(grid), dim3(KW), lds, st, bases, offs, nreads, S, ix, out4, pol, geo); \
    } while (0)
#define RK_KM_K(KT) do { if (nq == 1) RK_KM_LAUNCH(KT, 1); else RK_KM_LAUNCH(KT, 2); } while (0)
    switch (k) {
        case 8: RK_KM_K(8); break;
        case 9: RK_KM_K(9); break;
        case 10: RK_KM_K(10); break;
        case 11: RK_KM_K(11); break;
        case 12: RK_KM_K(12); break;
        case 13: RK_KM_K(13); break;
        case 14: RK_KM_K(14); break;
        case 15: RK_KM_K(15); break;
        case 16: RK_KM_K(16); break;
        default: return hipErrorInvalidValue;
    }
#undef RK_KM_K
#undef RK_KM_LAUNCH
    return hipGetLastError();
}

} // namespace rk
