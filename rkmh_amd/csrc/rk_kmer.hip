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
//   * the drain canonicalises a candidate (bit reverse), resolves it by k-mer with ONE 16-byte load of the exact map (km1), and uses the k-mer
//     itself as identity in the per-read hit multiset; two drain rounds' lookups are in flight together.
//   * the LDS layout is static (compile-time offsets: no address arithmetic on a run-time base, no SGPRs for the sub-arrays).
// Work decomposition as before: ONE WAVE = one tile of T consecutive reads, a workgroup is a single wave (no barriers).
#include "rk_kernels.hpp"

#include <cstdlib>
#include <type_traits>

namespace rk {

namespace {

constexpr int KW = 64;
constexpr int KM_MAX_T = 8;          // reads per tile (phase 2 takes eight reads in one pass; the group -> read search is unrolled for it)
constexpr int KM_MQ = 32;            // deferred multi-posting hits (a full list is worked off at once: 16 lanes per hit)
// (the timing experiments this kernel was shaped by -- parts switched off, loads faked -- live in tools/ubench/rk_kmer_ablation.hip, not here)
constexpr int KM_CH = 2;             // steps (64 groups = 256 windows each) whose filter sectors are requested together
constexpr int KM_QCAP = 64 + 4 * KW; // candidate queue entries (4 bytes each): a step adds at most 256 to a remainder of < 64
constexpr int KM_LDS_SMALL = 5120;   // static LDS of the common instantiation: 32 single-wave workgroups per CU (8 per SIMD)
constexpr int KM_LDS_BIG = 20480;    // ... of the one for large hit multisets / counter rows (8 per CU)
constexpr int KM_WAVES = 8;

// per-read reference counters: packed 8-bit (no read has more than 255 windows), packed 16-bit, or a 128-entry map ref -> count
enum { CM_DENSE8 = 0, CM_DENSE16 = 1, CM_SPARSE = 2 };

struct KmerGeom {
    int32_t T;         // reads per tile
    int32_t cwords;    // counter words per read
    int32_t dset;      // slots of the per-read hit multiset (power of two)
    int32_t xcd;
    int32_t nmin_cap;  // row field 3 = min(non-zero hashes, nmin_cap) (-M with a bounded min_num: rk_set_min_num_bound; else INT_MAX)
    int32_t L;         // hinted read length ...
    int32_t gpr[KM_MAX_KS];    // ... its groups per read = ceil(windows / 4), per k-mer size ...
    uint32_t magic[KM_MAX_KS]; // ... and ceil(2^32 / gpr): the group -> read division of tiles made of such reads
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) km_pair1 { uint32_t x, y; };
struct __attribute__((packed, aligned(4))) km_pair4 { uint32_t x, y; }; // two dwords at any dword-aligned address (posting entries)

// static LDS layout (dwords)
template <int NQ>
struct KmLds {
    static constexpr int PK = 0;                              // packed 2-bit image of the staged quads + two zero dwords (position P = byte P of the quads)
    static constexpr int INV = PK + NQ * 64 + 2;              // bit P set <=> base P is not ACGT (built only for tiles that hold one)
    static constexpr int Q = INV + NQ * 32 + 2;               // candidate queue: P | read << 12 (read 8: a lane without a group; rinfo[8] has no windows)
    static constexpr int MQ = Q + KM_QCAP;                    // [KM_MQ][2] deferred hits with a posting list: read | rank << 8, list offset
    static constexpr int RI = MQ + 2 * KM_MQ;                 // [9] uint4 {start position, windows, first group, -} of read t
    static constexpr int NZ = RI + 4 * (KM_MAX_T + 1);        // [8] zero hashes of read t
    static constexpr int BEST = NZ + KM_MAX_T;                // [8] max over increments of (count << 16 | 0xFFFF - ref)
    static constexpr int FLAGS = BEST + KM_MAX_T;             // [8] read must take the general path
    static constexpr int NWT = FLAGS + KM_MAX_T;              // [8] windows of read t, all k-mer sizes together
    static constexpr int FAM = NWT + KM_MAX_T;                // [8][4] hits of read t on lists stored as (base, exceptions): eight 16-bit counters, one per base
    static constexpr int FAMF = FAM + 4 * KM_MAX_T;           // [8] read t has such hits: its arg-max comes from a scan of the counters, not from the running best
    static constexpr int CNT = FAMF + KM_MAX_T;               // [T][cwords] per-reference counters, then [T][dset] hit multisets
    static_assert(RI % 4 == 0, "rinfo is read with 16-byte LDS loads");
    static_assert(CNT % 4 == 0, "the multisets are cleared with 16-byte LDS stores");
};
inline size_t km_lds_bytes(const KmerGeom& g, int nq) {
    return ((size_t)(nq == 1 ? KmLds<1>::CNT : KmLds<2>::CNT) + (size_t)g.T * (size_t)(g.cwords + g.dset)) * 4;
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

// FAM: the index holds posting lists stored as (base, exceptions) (genome families, build_kpost).  A panel without them -- BASELINE
// config 2's -- runs the instantiation that carries none of their state: no per-read base counters, no test for such values, no
// expansion in phase 2.
template <int KT, int NQ, int CMODE, bool BIG, bool FAM>
__global__ __launch_bounds__(KW, BIG ? 2 : KM_WAVES) void k_classify_kmer(const uint8_t* __restrict__ bases, const uint32_t* __restrict__ offs,
                                                                              uint32_t nreads, int S, RefIndex ix, KmerSets ksets,
                                                                              int32_t* __restrict__ out4, DevPolicy pol, KmerGeom geo) {
    // KT = 0: several k-mer sizes (ksets.k[], each from 8 to 16), one pass over the tile's windows per size
    // KT = 32: ONE k-mer size from 17 to KW_MAX_K, known at run time (ix.kpk): wide k-mers, 64-bit arithmetic, the km2 map (rk_device.hpp)
    static_assert(KT == 0 || KT == 32 || (KT >= 4 && KT <= 16), "a k-mer packs into 32 bits; the core is the (k-3)-mer");
    constexpr bool WIDE = KT == 32;
    using L = KmLds<NQ>;
    __shared__ __attribute__((aligned(16))) uint32_t smem[(BIG ? KM_LDS_BIG : KM_LDS_SMALL) / 4];
    constexpr uint32_t PAD_P = NQ * 1024;       // position of the image's zero padding: where lanes without a group look
    constexpr uint32_t clg = CMODE == CM_DENSE8 ? 2 : 1, cbits = 32u >> clg, cmask = (1u << cbits) - 1u;
    uint32_t* const pk = smem + L::PK;
    uint32_t* const inv = smem + L::INV;
    uint32_t* const q = smem + L::Q;
    uint32_t* const mq = smem + L::MQ;
    uint4* const rinfo = reinterpret_cast<uint4*>(smem + L::RI);
    uint32_t* const nzero = smem + L::NZ;
    uint32_t* const best = smem + L::BEST;
    uint32_t* const flags = smem + L::FLAGS;
    uint32_t* const nwtot = smem + L::NWT;
    uint32_t* const fam = smem + L::FAM;
    uint32_t* const famf = smem + L::FAMF;
    uint32_t* const cnt = smem + L::CNT;
    const int T = geo.T;
    const uint32_t CW = (uint32_t)geo.cwords, DS = (uint32_t)geo.dset;
    uint32_t* const dset = cnt + (uint32_t)T * CW;
    const uint32_t ds_shift = 32u - (uint32_t)__builtin_ctz(DS);
    const int lane = threadIdx.x;

    for (uint32_t i = lane; i < ((uint32_t)T * CW) >> 2; i += KW) reinterpret_cast<uint4*>(cnt)[i] = make_uint4(0u, 0u, 0u, 0u);
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
        if (nq > (uint32_t)(NQ * KW)) nq = NQ * KW; // oversized tile: the reads that lie inside the staged quads are served, the others rerouted
        const bool safe = base16 >= gb && base16 + 16u * (uintptr_t)nq <= safe_hi; // wave-uniform
#pragma unroll
        for (int r = 0; r < NQ; ++r) {
            const uint32_t qi = (uint32_t)r * KW + (uint32_t)lane;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (qi < nq) {
                const uintptr_t qa = base16 + 16u * (uintptr_t)qi;
                if (safe) { // through a global-address-space pointer: the integer round trip would otherwise make it a flat load
                    typedef const u32x4 __attribute__((address_space(1))) * gq_t;
                    v = __builtin_nontemporal_load((gq_t)qa); // the bases are read once: streaming loads keep them from evicting the filter and the map from L2
                }
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

    // One tile per workgroup (measured with batches streaming from HBM: 1 tile 0.381 ms, 2 tiles with the second one's bases
    // prefetched 0.383, 4 tiles 0.392, 8 tiles 0.402 -- short-lived waves let the dispatcher balance the CUs, and the other waves of
    // the SIMD hide a fresh wave's two cold loads).  XCD-aware ownership: the dispatcher deals workgroups round-robin over the 8
    // XCDs, so workgroup b takes tile (b % 8) * ceil(grid / 8) + b / 8 and each XCD's L2 sees one contiguous eighth of the batch.
    uint32_t tile = blockIdx.x;
    if (geo.xcd) { const uint32_t per = (gridDim.x + 7u) >> 3; tile = (blockIdx.x & 7u) * per + (blockIdx.x >> 3); }
    if (tile >= ntiles) return;
    uint32_t cur_a = 0, cur_b = 0, cur_o = 0;
    load_offsets(tile, cur_a, cur_b, cur_o);
    fetch_tile(cur_a, cur_b);
    {
        const uint32_t r0 = tile * (uint32_t)T;
        const uint32_t ta = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur_a);
        const uint32_t tb = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur_b);
        const uint32_t mis = (uint32_t)((gb + ta) & 15u);       // position of the tile's first base inside quad 0
        const uint32_t nbytes = mis + (tb - ta);
        const uint32_t o_next = (uint32_t)__shfl_down((int)cur_o, 1);
        const uint32_t len = o_next - cur_o;
        // A tile longer than the staged quads (only possible when the caller's length hint was too small): the reads that end
        // inside the quads -- a prefix of the tile -- are served, the others are handed back (rows of -2).
        int Tn = tile_reads(tile);
        uint32_t nq = (nbytes + 15u) >> 4;
        if (nbytes > (uint32_t)(NQ * 1024)) { // wave-uniform, rare
            const int Tn0 = Tn;
            Tn = __popcll(__ballot(lane < Tn0 && o_next - ta + mis <= (uint32_t)(NQ * 1024)));
            if (lane >= Tn && lane < Tn0) reinterpret_cast<int4*>(out4)[r0 + lane] = make_int4(-2, 0, 0, 0);
            nq = NQ * KW;
        }

        // ---- phase 0: per-read bookkeeping (lane t = read t), packed image, validity ballot ---------------------------------
        constexpr int NKC = KT ? 1 : KM_MAX_KS;
        const int NK = KT ? 1 : ksets.n; // k-mer sizes: the tile's windows are walked once per size
        uint32_t nw_all = 0;             // windows of this lane's read, all sizes together
        if (lane < Tn) {
#pragma unroll
            for (int j = 0; j < NKC; ++j) if (j < NK) nw_all += (uint32_t)num_windows((int)len, WIDE ? (int)ix.kpk : (KT ? KT : ksets.k[j]), pol.drop_last_window);
        }
        if constexpr (FAM) { if (lane < 4 * KM_MAX_T) fam[lane] = 0; }
        if (lane < KM_MAX_T) {
            nzero[lane] = 0; best[lane] = 0; nwtot[lane] = nw_all;
            if constexpr (FAM) famf[lane] = 0;
            // more windows than a packed counter can count (only possible when the caller's length hint was too small)
            flags[lane] = nw_all > (CMODE == CM_SPARSE ? 0x7FFu : cmask) ? 1u : 0u;
        }
        const uint32_t ulen = (uint32_t)__builtin_amdgcn_readfirstlane((int)len);
        const bool same_len = __ballot(lane < Tn && len != ulen) == 0ull;
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
        for (int kk = 0; kk < NK; ++kk) {
        // ---- one k-mer size: its windows, its filter and map; the per-read counters, zero counts and running best carry over ----
        const int k = WIDE ? (int)ix.kpk : (KT ? KT : ksets.k[kk]);
        const uint32_t KMASK = (WIDE || k == 16) ? 0xffffffffu : ((1u << (2 * k)) - 1u);
        const uint32_t CMASK = WIDE ? 0xffffffffu : (1u << (2 * (k - 3))) - 1u;
        const uint64_t KMASK64 = WIDE ? (1ull << (2 * k)) - 1ull : 0ull, CMASK64 = WIDE ? (1ull << (2 * (k - 3))) - 1ull : 0ull; // wide k-mers
        const uint32_t KBITS = (1u << k) - 1u; // k validity bits
        const uint4* const kf4p = KT ? ix.kf4 : ksets.kf4[kk];
        const uint4* const km1p = KT ? ix.km1 : ksets.km1[kk];
        const uint32_t* const km1v = KT ? ix.km1_vals : ksets.km1_vals[kk];
        const uint32_t kf4_n = KT ? ix.kf4_n : ksets.kf4_n[kk], km1_b = KT ? ix.km1_b : ksets.km1_b[kk];
        wave_sync(); // the previous size's pass is over (first pass: the image is staged)
        uint32_t nw = 0, ng = 0;
        if (lane < Tn) { nw = (uint32_t)num_windows((int)len, k, pol.drop_last_window); ng = (nw + 3u) >> 2; }
        uint32_t gs = ng; // inclusive prefix of the group counts over the tile's reads (lanes 0..7)
        // (DPP row shifts: lanes 0..7 lie in one 16-lane row, a lane without a source reads 0)
        gs += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)gs, 0x111, 0xf, 0xf, true); // row_shr:1
        gs += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)gs, 0x112, 0xf, 0xf, true); // row_shr:2
        gs += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)gs, 0x114, 0xf, 0xf, true); // row_shr:4
        const uint32_t NG = (uint32_t)__builtin_amdgcn_readlane((int)gs, KM_MAX_T - 1); // groups of the tile
        if (lane <= KM_MAX_T) rinfo[lane] = make_uint4(cur_o - ta + mis, nw, gs - ng, 0u); // entries past the tile's last read (and entry 8, always): no windows, first group = NG
        { // clear the hit multisets (repeats only exist within one k-mer size: different sizes never share a hash): 16 bytes per lane and store
            uint4* d4 = reinterpret_cast<uint4*>(dset);
            for (uint32_t i = lane; i < ((uint32_t)Tn * DS) >> 2; i += KW) d4[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        wave_sync();

        // group -> read mapping: tiles of equally long reads divide by a magic constant, the others search the group starts
        const uint32_t nw_u = (uint32_t)num_windows((int)ulen, k, pol.drop_last_window), gpr_u = (nw_u + 3u) >> 2;
        const bool uniform = same_len && gpr_u >= 2u; // >= 2: the magic division needs a divisor > 1
        uint32_t magic = 0;
        if (uniform) magic = ulen == (uint32_t)geo.L ? geo.magic[kk] : (uint32_t)__builtin_amdgcn_readfirstlane((int)(0xFFFFFFFFu / gpr_u + 1u));
        // P0 of group G of a tile of equally long reads: read t = G / gpr starts at mis + t ulen, group g = G - t gpr at + 4 g, so
        // P0 = (mis + 4 G) + t (ulen - 4 gpr): one multiply-add.  The two constants live in VGPRs (broadcast): as scalars they were
        // spilled to VGPR lanes and read back with a v_readlane each in every step (the kernel has registers to spare, not SGPRs).
        uint32_t magic_v = magic, dlen_v = ulen - 4u * gpr_u;
        asm volatile("" : "+v"(magic_v), "+v"(dlen_v));

        // +1 for reference `ref` of read t, whose counter row starts crow_b bytes into cnt; the monotone counters make
        // (max_shared, first max_id) a running atomicMax
        // count_posting: +1 for reference `ref` of read t; returns the candidate for the read's running maximum, count << 16 | ~ref
        // (0: the read left for the general path).  add_posting = count + atomicMax.
        auto count_posting = [&](uint32_t t, uint32_t crow_b, uint32_t ref) -> uint32_t {
            uint32_t c;
            if constexpr (CMODE == CM_SPARSE) { // large panels keep (ref, count) pairs of the references a read really hits
                uint32_t* crow = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(cnt) + crow_b);
                const uint32_t M1 = CW - 1u, key = ref + 1u;
                uint32_t idx = ((ref * 0x9E3779B1u) >> 16) & M1, probe = 0;
                c = 0;
                for (; probe <= M1; ++probe) {
                    const uint32_t old = atomicCAS(&crow[idx], 0u, (key << 11) | 1u);
                    if (old == 0u) { c = 1u; break; }
                    if ((old >> 11) == key) { c = (atomicAdd(&crow[idx], 1u) & 0x7FFu) + 1u; break; }
                    idx = (idx + 1u) & M1;
                }
                if (probe > M1) { flags[t] = 1; return 0u; } // the read hits more references than the map holds: general path
            } else {
                const uint32_t sh = (ref << (5 - clg)) & (32u - cbits);           // bit position of the counter inside its word
                const uint32_t woff = (ref >> clg) << 2;                          // byte offset of the word inside the row
                const uint32_t old = atomicAdd(reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(cnt) + crow_b + woff), 1u << sh);
                c = __builtin_amdgcn_ubfe(old, sh, cbits) + 1u;
            }
            return (c << 16) + (0xFFFFu - ref);
        };
        auto add_posting = [&](uint32_t t, uint32_t crow_b, uint32_t ref) {
            const uint32_t v = count_posting(t, crow_b, ref);
            if (v) atomicMax(&best[t], v);
        };
        uint32_t mqn = 0; // deferred list hits (wave-uniform)
        auto process_mq = [&]() { // 16 lanes walk one hit's posting list, 4 hits at a time
            wave_sync();
            const uint32_t g = (uint32_t)lane >> 4, sl = (uint32_t)lane & 15u;
            // The 16 lanes of a hit all feed the SAME running maximum best[t]: left to themselves they issue 16 atomicMax on one LDS
            // address per step (and the four hits of a step are often four hits of one read: 64 on one address, served one after the
            // other).  Their candidates are reduced over the 16-lane row first (DPP) and one lane issues the atomic -- this and working a
            // full list of deferred hits off at once took reads of a 60-member family of near-identical references (every hit: 60
            // postings) from 2.3 to 1.1 ms per 1 M reads and BASELINE config 3's panel from 0.90 to 0.69 (profiles/r04_c3_probe.txt).
            // (Tried and rejected: lane sl taking postings 4 sl .. 4 sl + 3 so that neighbouring lanes do not meet in one counter word --
            // four dependent atomics per lane cost more than the four-way conflicts: family reads 1.1 -> 2.0 ms.)
            // (Also tried and rejected: the four rows of a step -- often four hits of one read with one list -- starting at different
            // chunks of the list, so that they do not add to the same counter words in the same instruction: 1.08 -> 1.14 ms.)
            auto walk = [&](uint32_t tr, const uint32_t* lp, uint32_t hdr, const km_pair4& pm0) {
                const uint32_t t = tr & 0xFFu, n = hdr & 0xFFFFFFu, base = FAM ? hdr >> 24 : 0u;
                if (base) {
                    // a list stored as (base, exceptions) (build_kpost; every member holds the hash once, so only a k-mer's first
                    // occurrence counts): one more hit on the base -- expanded once per read in phase 2 -- and +1 / -1 for the few
                    // references in which this list differs from it.  A packed counter may pass below zero on the way; the row is
                    // only read after the expansion, when every field is its final value again (the adds commute mod 2^32).
                    if ((tr >> 8) != 0u) return;
                    if (sl == 0) { atomicAdd(&fam[4u * t + ((base - 1u) >> 1)], 1u << (16u * ((base - 1u) & 1u))); famf[t] = 1u; }
                    for (uint32_t c0 = 0; c0 < n; c0 += 16) {
                        const uint32_t c = c0 + sl;
                        if (c < n) {
                            km_pair4 pm = pm0;
                            if (c0) pm = *reinterpret_cast<const km_pair4*>(lp + 1 + 2 * c);
                            const uint32_t sh = (pm.x << (5 - clg)) & (32u - cbits), woff = (pm.x >> clg) << 2;
                            atomicAdd(reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(cnt) + __umul24(t, CW * 4u) + woff), (pm.y >> 31) ? 0u - (1u << sh) : 1u << sh);
                        }
                    }
                    return;
                }
                for (uint32_t c0 = 0; c0 < n; c0 += 16) { // (n is the same for the 16 lanes of the row: they run the same trips)
                    const uint32_t c = c0 + sl;
                    uint32_t v = 0;
                    if (c < n) {
                        km_pair4 pm = pm0; // (reference, multiplicity)
                        if (c0) pm = *reinterpret_cast<const km_pair4*>(lp + 1 + 2 * c);
                        if ((tr >> 8) < pm.y) v = count_posting(t, __umul24(t, CW * 4u), pm.x);
                    }
                    v = (uint32_t)row_max_i32((int)v); // (candidates are < 2^31: counts below 2^15)
                    if (sl == 0 && v) atomicMax(&best[t], v);
                }
            };
            // A list's header and its first sixteen postings are requested TOGETHER (the posting array ends in 256 bytes of padding,
            // rk_api.hip), and the NEXT hit's while this one is counted: the posting lists live in global memory and a row used to pay
            // two dependent round trips per hit, one hit after the other.
            uint32_t j = g;
            uint32_t tr = 0, n = 0;
            const uint32_t* lp = ix.kpost;
            km_pair4 pm0{0u, 0u};
            if (j < mqn) { tr = mq[2 * j]; lp = ix.kpost + mq[2 * j + 1]; n = lp[0]; pm0 = *reinterpret_cast<const km_pair4*>(lp + 1 + 2 * sl); }
            while (j < mqn) {
                const uint32_t jn = j + KW / 16;
                uint32_t trn = 0, nn = 0;
                const uint32_t* lpn = ix.kpost;
                km_pair4 pmn{0u, 0u};
                if (jn < mqn) { trn = mq[2 * jn]; lpn = ix.kpost + mq[2 * jn + 1]; nn = lpn[0]; pmn = *reinterpret_cast<const km_pair4*>(lpn + 1 + 2 * sl); }
                walk(tr, lp, n, pm0);
                j = jn; tr = trn; lp = lpn; n = nn; pm0 = pmn;
            }
            wave_sync();
            mqn = 0;
        };
        // one candidate of the queue: canonical k-mer, its read, its bucket of the exact map (load in flight on return)
        struct Cand { uint32_t key, t, y, key_hi; uint4 c; }; // (wide: key / key_hi = the k-mer, y = bucket << KW_TAG | tag)
        const uint32_t km_r = WIDE ? 8u : 2u * (uint32_t)k - km1_b, km_vb1 = 32u - KM1_HB - km_r;      // remainder bits; value id bits + the flag bit (wide: unused)
        const uint32_t km_vmask = (1u << (km_vb1 - 1u)) - 1u, km_rmask = (1u << km_r) - 1u, nref = (uint32_t)ix.nref;
        // wide k-mers: the bucket's cells whose (hop, tag) match are settled by the full k-mer in kkeys; res = {value id, -, key number}
        // (value id KW_VID_NONE: no such k-mer); again: the search goes on in the next bucket
        auto wide_probe = [&](const Cand& c, const uint4& b, uint32_t hop, uint3& res, bool& again) {
            const uint32_t want = (hop << KW_TAG) | (c.y & ((1u << KW_TAG) - 1u));
            const uint32_t cells[4] = {b.x, b.y, b.z, b.w};
            res = make_uint3(KW_VID_NONE, 0u, 0u);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t num = cells[q] & KW_EMPTY;
                if ((cells[q] >> (KW_IDBITS + 1)) == want && num != KW_EMPTY && res.x == KW_VID_NONE) {
                    const uint2 kk = ix.kkeys[num];
                    if (kk.x == c.key && (kk.y & 0xFFu) == c.key_hi) res = make_uint3(kk.y >> 8, 0u, num);
                }
            }
            again = res.x == KW_VID_NONE && ((b.w >> KW_IDBITS) & 1u) != 0u && c.t != 0xFFFFFFFFu;
        };
        auto lookup = [&](uint32_t e, uint32_t qn) -> Cand {
            Cand c;
            uint32_t ent = PAD_P | (8u << 12); // past the queue's end: the all-A k-mer of the padding, of read 8 which has no windows
            if (e < qn) ent = q[e];
            const uint32_t P = ent & 4095u, t = ent >> 12;
            const km_pair1 w = *reinterpret_cast<const km_pair1*>(reinterpret_cast<const uint8_t*>(pk) + (P >> 2));
            const uint2 ri = *reinterpret_cast<const uint2*>(&rinfo[t]);
            c.t = P - ri.x < ri.y ? t : 0xFFFFFFFFu; // window number = position - read start; the last group of a read may reach past its last window
            if constexpr (WIDE) {
                const uint64_t x = (join64(w.x, w.y) >> ((P & 3u) << 1)) & KMASK64; // (58 bits follow the byte the window starts in: k <= 20 needs 40)
                const uint64_t r = packed_revcomp64(x, k);
                const uint64_t key = x < r ? x : r;
                c.key = (uint32_t)key; c.key_hi = (uint32_t)(key >> 32);
                c.y = (uint32_t)(kw_y(key, k) >> (2u * (uint32_t)k - km1_b - KW_TAG));
                c.c = km1p[c.y >> KW_TAG];
                return c;
            }
            const uint32_t x = __builtin_amdgcn_alignbit(w.y, w.x, (P & 3u) << 1) & KMASK;
            const uint32_t r = packed_revcomp(x, k);
            c.key = x < r ? x : r;
            c.key_hi = 0;
            c.y = km1_y(c.key, k);
            c.c = km1p[c.y >> km_r];
            return c;
        };
        // the cell of the bucket whose tag equals the one in wantsh (all ones: none)
        auto match_cell = [&](const uint4& b, uint32_t wantsh) -> uint32_t {
            const uint32_t lim = 1u << km_vb1;
            return (b.x ^ wantsh) < lim ? b.x : ((b.y ^ wantsh) < lim ? b.y : ((b.z ^ wantsh) < lim ? b.z : ((b.w ^ wantsh) < lim ? b.w : 0xFFFFFFFFu)));
        };
        // first bucket: the cell, and whether the search must go on (a miss in a bucket that sent a key further on: that key may be
        // this one, stored up to 7 buckets past its own and tagged with the distance)
        auto first_match = [&](const Cand& c, uint32_t& cell, bool& again) {
            cell = match_cell(c.c, (c.y & km_rmask) << km_vb1);
            again = (cell & km_vmask) == km_vmask && ((c.c.w >> (km_vb1 - 1u)) & 1u) != 0u && c.t != 0xFFFFFFFFu;
        };
        auto next_match = [&](const Cand& c, uint32_t hop, uint32_t& cell, bool& again) {
            if (again) {
                const uint4 nb = km1p[((c.y >> km_r) + hop) & ((1u << km1_b) - 1u)];
                cell = match_cell(nb, ((c.y & km_rmask) << km_vb1) | (hop << (32u - KM1_HB)));
                again = (cell & km_vmask) == km_vmask && ((nb.w >> (km_vb1 - 1u)) & 1u) != 0u;
            }
        };
        auto apply = [&](const Cand& c, uint32_t cell, uint3 wres = make_uint3(0u, 0u, 0u)) {
            // narrow: the cell's value id (all ones: no such k-mer; all ones - 1: canonical hash 0).  wide: wres from wide_probe.
            uint32_t vid = cell & km_vmask;
            bool hit = vid != km_vmask && c.t != 0xFFFFFFFFu; // else: a false positive of the bit filter, or a window past its read's last
            bool zero = vid == km_vmask - 1u;
            if constexpr (WIDE) {
                vid = wres.x;
                hit = vid != KW_VID_NONE && c.t != 0xFFFFFFFFu;
                zero = vid == KW_VID_ZERO;
                // -M with a bounded min_num: a key the mask drops is a zero-hash k-mer (the narrow form reads a masked copy of its map)
                if (hit && !zero && ix.keepkey) { const uint32_t kid = ix.kslots[wres.z]; if (!((ix.keepkey[kid >> 5] >> (kid & 31u)) & 1u)) zero = true; }
            }
            uint32_t val = vid | (1u << 20); // a single posting of multiplicity 1, in the RefIndex::kv value format
            uint32_t valy = 0, valz = 0, valw = 0;
            if (hit && vid >= nref && !zero) { // compound value, four dwords (a few KB: L1-resident)
                const uint4 vv = *reinterpret_cast<const uint4*>(km1v + 4u * (vid - nref));
                val = vv.x; valy = vv.y; valz = vv.z; valw = vv.w;
                if constexpr (CMODE == CM_SPARSE) { if ((val >> 29) == 7u) val = 0x80000000u | valz; } // (base, exceptions): the sparse counters cannot subtract
            }
            uint32_t rank = 0;
            bool multi = false;
            if (hit) {
                if (zero) atomicAdd(&nzero[c.t], 1u); // a k-mer whose canonical hash is 0
                else {
                    // hit MULTISET of the read: the canonical k-mer (+1: never 0xFFFFFFFF) is the key's identity, every occurrence
                    // adds one more entry, and the entries passed on the way to the free slot are this occurrence's rank.  The merge
                    // of rkmh.cpp:869 counts min(occurrences in the read, multiplicity in the sketch): occurrence `rank` of a k-mer
                    // counts for a posting iff rank < its multiplicity.
                    uint32_t* ds = dset + c.t * DS;
                    const uint32_t id = (WIDE ? wres.z : c.key) + 1u; // (wide: the key's number in kkeys)
                    uint32_t idx = ((id - 1u) * 0x9E3779B1u) >> ds_shift;
                    uint32_t old = atomicCAS(&ds[idx], 0u, id);
                    uint32_t probes = 1;
                    while (old != 0u && probes < DS) { // the wave leaves this loop when its last lane has found a free slot
                        rank += old == id ? 1u : 0u;
                        idx = (idx + 1u) & (DS - 1u);
                        old = atomicCAS(&ds[idx], 0u, id);
                        ++probes;
                    }
                    if (old != 0u) flags[c.t] = 1; // more hits than the set holds: general path
                    else if (FAM && (val >> 29) == 7u) { // a list within eight exceptions of a base list (build_kpost): one more hit on the base, +-1 for the exceptions
                        if (rank == 0) {
                            const uint32_t base = (val >> 26) & 7u, nex = (val >> 22) & 15u, crow_b = __umul24(c.t, CW * 4u);
                            atomicAdd(&fam[4u * c.t + (base >> 1)], 1u << (16u * (base & 1u)));
                            famf[c.t] = 1u;
                            auto signed_add = [&](uint32_t f) { // f = reference | 512 for -1
                                const uint32_t ref = f & 511u, sh = (ref << (5 - clg)) & (32u - cbits);
                                atomicAdd(reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(cnt) + crow_b + ((ref >> clg) << 2)), (f & 512u) ? 0u - (1u << sh) : 1u << sh);
                            };
                            if (nex > 0u) signed_add(val & 1023u);
                            if (nex > 1u) signed_add((val >> 10) & 1023u);
                            if (nex > 2u) signed_add(valy & 1023u);
                            if (nex > 3u) signed_add((valy >> 10) & 1023u);
                            if (nex > 4u) signed_add((valy >> 20) & 1023u);
                            if (nex > 5u) signed_add(valw & 1023u);
                            if (nex > 6u) signed_add((valw >> 10) & 1023u);
                            if (nex > 7u) signed_add((valw >> 20) & 1023u);
                        }
                    }
                    else if ((val >> 30) == 3u) { // three to six references that hold the hash once each, stored inline (build_index)
                        if (rank == 0) {
                            const uint32_t crow_b = __umul24(c.t, CW * 4u), n3 = (val >> 27) & 3u;
                            add_posting(c.t, crow_b, val & 511u);
                            add_posting(c.t, crow_b, (val >> 9) & 511u);
                            add_posting(c.t, crow_b, (val >> 18) & 511u);
                            if (n3 > 0u) add_posting(c.t, crow_b, valy & 511u);
                            if (n3 > 1u) add_posting(c.t, crow_b, (valy >> 9) & 511u);
                            if (n3 > 2u) add_posting(c.t, crow_b, (valy >> 18) & 511u);
                        }
                    } else if (!(val >> 31)) { // one posting (with multiplicity) or two single postings, stored inline
                        const bool two = ((val >> 29) & 3u) != 0u;
                        if (rank < (two ? 1u : ((val >> 20) & 0x1FFu))) {
                            const uint32_t crow_b = __umul24(c.t, CW * 4u);
                            add_posting(c.t, crow_b, val & (two ? 0x7FFu : 0xFFFFFu));
                            if (two) add_posting(c.t, crow_b, (val >> 11) & 0x7FFu);
                        }
                    } else multi = true;
                }
            }
            // hits with a posting list are deferred to the end of the drain (16 lanes then walk each list); when the list of deferred
            // hits is full it is worked off at once and filled again (a read of a family of near-identical references has nothing but
            // such hits: walking them one lane per list took most of its time)
            // the form of the list this kernel walks: (base, exceptions) where there is one -- the sparse counters cannot subtract
            const uint32_t listoff = (CMODE == CM_SPARSE ? valz : val) & 0x3fffffffu;
            uint64_t mm = __ballot(multi);
            while (mm) { // wave-uniform
                const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
                const uint32_t space = (uint32_t)KM_MQ - mqn;
                if (multi && pos < space) {
                    mq[2 * (mqn + pos)] = c.t | (rank << 8); mq[2 * (mqn + pos) + 1] = listoff;
                    multi = false;
                }
                const uint32_t tot = (uint32_t)__popcll(mm);
                mqn += tot < space ? tot : space;
                if (mqn == (uint32_t)KM_MQ) process_mq();
                mm = __ballot(multi);
            }
        };
        auto drain = [&](uint32_t qn) {
            for (uint32_t e0 = 0; e0 < qn; e0 += 2 * KW) { // two rounds of lookups in flight
                const bool two = e0 + KW < qn;
                const Cand a = lookup(e0 + (uint32_t)lane, qn);
                Cand b = a;
                if (two) b = lookup(e0 + KW + (uint32_t)lane, qn);
                if constexpr (WIDE) {
                    uint3 ra, rb = make_uint3(KW_VID_NONE, 0u, 0u);
                    bool ga, gb = false;
                    wide_probe(a, a.c, 0u, ra, ga);
                    if (two) wide_probe(b, b.c, 0u, rb, gb);
                    for (uint32_t hop = 1; hop < (1u << KM1_HB) && __ballot(ga || gb); ++hop) {
                        const uint32_t bm = (1u << km1_b) - 1u;
                        if (ga) { const uint4 nb = km1p[((a.y >> KW_TAG) + hop) & bm]; wide_probe(a, nb, hop, ra, ga); }
                        if (gb) { const uint4 nb = km1p[((b.y >> KW_TAG) + hop) & bm]; wide_probe(b, nb, hop, rb, gb); }
                    }
                    apply(a, 0u, ra);
                    if (two) apply(b, 0u, rb);
                    continue;
                }
                uint32_t ca, cb = 0xFFFFFFFFu;
                bool ga, gb = false;
                first_match(a, ca, ga);
                if (two) first_match(b, cb, gb);
                // the searches that go on (about one lookup in ten, once) are continued for both rounds together: one more memory
                // round trip per hop for the tile, not per round; the wave leaves when its last lane is done
                for (uint32_t hop = 1; hop < (1u << KM1_HB) && __ballot(ga || gb); ++hop) { next_match(a, hop, ca, ga); next_match(b, hop, cb, gb); }
                apply(a, ca);
                if (two) apply(b, cb);
            }
            if (mqn) process_mq();
        };

        // ---- window phase: 64 groups (256 windows) per step, KM_CH steps' filter sectors requested together ---------------------
        const uint32_t nsteps = (NG + KW - 1) / KW;
        // a last step of 1 .. 16 groups runs a QUARTER wide: four lanes per group, one window (and one dword of the sector) each --
        // a fourth of the window tests for the step that would otherwise run 64 lanes for a dozen groups (six 150-base reads: 204 groups)
        const bool qlast = !WIDE && NG != 0u && ((NG - 1u) & (uint32_t)(KW - 1)) < 16u;
        const uint32_t kf4_n16 = kf4_n << 4; // (sector count < 2^28: checked where the filter is built)
        uint32_t qcount = 0;
        uint32_t step = 0;
        for (;;) {
            // issue: every lane computes; a lane without a group looks at the image's zero padding (the all-A k-mer) and tags its
            // candidates, should the filter pass that k-mer, with read 8, which has no windows -- the drain drops them
            uint32_t wl[KM_CH], wh[KM_CH], e0v[KM_CH], fq[KM_CH];
            u32x4 fw[KM_CH];
            bool isqv[KM_CH];
#pragma unroll
            for (int s = 0; s < KM_CH; ++s) {
                const bool isq = qlast && step + (uint32_t)s + 1u == nsteps; // wave-uniform
                isqv[s] = isq;
                const uint32_t jq = isq ? (uint32_t)lane & 3u : 0u;
                const uint32_t G = (step + (uint32_t)s) * KW + (isq ? (uint32_t)lane >> 2 : (uint32_t)lane);
                uint32_t t, P0;
                if (uniform) { // wave-uniform
                    t = __umulhi(G, magic_v);
                    P0 = __umul24(t, dlen_v) + ((G << 2) + mis);
                } else {
                    t = 0; // the last read whose first group is <= G (rinfo[i].z of reads past the tile's last is NG)
#pragma unroll
                    for (int i = 1; i < KM_MAX_T; ++i) t += G >= rinfo[i].z ? 1u : 0u;
                    const uint4 ri = rinfo[t];
                    P0 = ri.x + 4u * (G - ri.z);
                }
                const bool act = G < NG;
                if (!act) { P0 = PAD_P; t = 8u; }
                const km_pair1 w = *reinterpret_cast<const km_pair1*>(reinterpret_cast<const uint8_t*>(pk) + (P0 >> 2));
                const uint32_t sh = (P0 & 3u) << 1;
                wl[s] = __builtin_amdgcn_alignbit(w.y, w.x, sh);
                wh[s] = w.y >> sh;
                uint32_t core = k == 16 ? wl[s] >> 6 : (wl[s] >> 6) & CMASK; // k = 16: the 13-mer is all of bits 6..31
                if constexpr (WIDE) core = kw_fold((join64(wl[s], wh[s]) >> 6) & CMASK64); // the (k - 3)-mer, up to 34 bits, folded to 32
                // byte offset of the sector: 16 * (hashed core scaled to [0, kf4_n)) = the high product with 16 kf4_n, less its low four bits
                const uint32_t sect_b = __umulhi(core * 0x85EBCA6Bu, kf4_n16) & ~15u;
                e0v[s] = P0 | (t << 12);
                fq[s] = 0u;
                fw[s] = u32x4{0u, 0u, 0u, 0u}; // (defined on both paths: left undefined, the compiler reuses a register still in flight and waits)
                if (isq) { // this lane's window of the group: its k-mer moves to the front, its entry names position + jq
                    wl[s] = __builtin_amdgcn_alignbit(wh[s], wl[s], 2u * jq);
                    if (act) e0v[s] += jq;
                    fq[s] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(kf4p) + (sect_b + 4u * jq));
                } else fw[s] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(kf4p) + sect_b);
            }
            // test + push.  has_invalid (a tile with a non-ACGT base, rare) runs its own copy of the code.
            bool stop = false;
            uint32_t done = 0;
            auto test_step = [&](int s, auto inv_tag, auto nw_tag) {
                constexpr bool INV = decltype(inv_tag)::value;
                constexpr int NW = decltype(nw_tag)::value; // windows this lane tests: 4, or 1 in a quarter-wide step (wl / fq / e0v are then that window's)
                uint32_t ib = 0, nz = 0, nwt = 0, t_ = 0, o0 = 0;
                bool act = true;
                if constexpr (INV) {
                    const uint32_t P0 = e0v[s] & 4095u;
                    t_ = e0v[s] >> 12;
                    act = t_ != 8u;
                    if (act) {
                        ib = __builtin_amdgcn_alignbit(inv[(P0 >> 5) + 1], inv[P0 >> 5], P0 & 31u);
                        const uint2 ri = *reinterpret_cast<const uint2*>(&rinfo[t_]);
                        nwt = ri.y; o0 = P0 - ri.x; // windows of the read, window number of this lane's first window
                    }
                }
#pragma unroll
                for (int j = 0; j < NW; ++j) {
                    uint32_t x = j == 0 ? wl[s] : __builtin_amdgcn_alignbit(wh[s], wl[s], 2 * j);
                    if (k < 16) x &= KMASK;
                    if constexpr (WIDE) x = kw_fold((join64(wl[s], wh[s]) >> (2 * j)) & KMASK64);
                    const uint32_t f = NW == 1 ? fq[s] : (j == 0 ? fw[s].x : (j == 1 ? fw[s].y : (j == 2 ? fw[s].z : fw[s].w)));
                    const uint32_t fb = kf4_bits_dev(x);
                    bool cand = (fb & f) == fb; // (not (fb & ~f) == 0: the compiler moves a NOT of the loaded dword up to the load and waits there)
                    if constexpr (INV) {
                        if (act && ((ib >> j) & KBITS) != 0u) { // a window holding a non-ACGT base hashes to 0 (if it is a window of the read)
                            cand = false;
                            if (o0 + (uint32_t)j < nwt) ++nz;
                        }
                    }
                    const uint64_t m = __builtin_amdgcn_ballot_w64(cand);
                    // inverse_ballot turns the mask back into the branch predicate (no second compare for the exec mask)
                    if (__builtin_amdgcn_inverse_ballot_w64(m)) {
                        const uint32_t mb = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(q) + ((mb << 2) + (qcount << 2))) = e0v[s] + (uint32_t)j;
                    }
                    qcount += (uint32_t)__builtin_popcountll(m);
                }
                if constexpr (INV) { if (nz) atomicAdd(&nzero[t_], nz); }
            };
#pragma unroll
            for (int s = 0; s < KM_CH; ++s) {
                if (step + (uint32_t)s < nsteps && !stop) { // wave-uniform
                    if (qcount + 4u * KW > (uint32_t)KM_QCAP) stop = true;
                    else {
                        ++done;
                        if (isqv[s]) { // wave-uniform
                            if (has_invalid) test_step(s, std::true_type{}, std::integral_constant<int, 1>{});
                            else test_step(s, std::false_type{}, std::integral_constant<int, 1>{});
                        } else if (has_invalid) test_step(s, std::true_type{}, std::integral_constant<int, 4>{});
                        else test_step(s, std::false_type{}, std::integral_constant<int, 4>{});
                    }
                }
            }
            step += done;
            const bool last = step >= nsteps;
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

        } // k-mer sizes
        wave_sync();

        // ---- phase 2: 16 lanes per read, or 8 when the tile holds more than four ------------------------------------------------
        {
            const int lsh = Tn > 4 ? 3 : 4, LPR = 1 << lsh; // wave-uniform
            const int g = lane >> lsh, sl = lane & (LPR - 1);
            for (int t = g; t < Tn; t += KW >> lsh) {
                uint32_t* ct = cnt + (uint32_t)t * CW;
                const int nmins = (int)nwtot[t] - (int)nzero[t];
                // bottom-S selection matters, or the hit multiset overflowed: exact answer comes from the general path
                const bool reroute = nmins > S || flags[t] != 0;
                uint32_t bk = best[t];
                if (reroute) {
                    if (sl == 0) reinterpret_cast<int4*>(out4)[r0 + t] = make_int4(-2, 0, 0, 0);
                    continue;
                }
                if constexpr (FAM && CMODE != CM_SPARSE) {
                    if (famf[t]) { // hits on (base, exceptions) lists: each touched base is expanded into the counters ONCE, then the
                        // maximum and its first reference come from a scan of the row (the running best saw partial counts)
                        for (uint32_t b = 0; b < 8u; ++b) {
                            const uint32_t h = (fam[4u * (uint32_t)t + (b >> 1)] >> (16u * (b & 1u))) & 0xFFFFu;
                            if (h == 0u) continue;
                            const uint32_t start = ix.kbase[2u * b], nm = ix.kbase[2u * b + 1u];
                            for (uint32_t m0 = 0; m0 < nm; m0 += (uint32_t)LPR) {
                                const uint32_t m = m0 + (uint32_t)sl;
                                if (m < nm) {
                                    const uint32_t ref = ix.kbase[start + m];
                                    atomicAdd(&ct[ref >> clg], h << ((ref << (5 - clg)) & (32u - cbits)));
                                }
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        uint32_t km = 0;
                        for (uint32_t w = (uint32_t)sl; w < CW; w += (uint32_t)LPR) {
                            uint32_t x = ct[w];
#pragma unroll
                            for (uint32_t q = 0; q < (1u << clg); ++q) {
                                const uint32_t cq = x & cmask, ref = (w << clg) + q;
                                x >>= cbits;
                                const uint32_t key = (cq << 16) + (0xFFFFu - ref);
                                km = (cq != 0u && key > km) ? key : km;
                            }
                        }
                        bk = (uint32_t)(LPR == 16 ? row_max_i32((int)km) : half_row_max_i32((int)km));
                    }
                }
                // first max wins (rkmh.cpp:878); diff = max - best EARLIER score (untouched refs score 0; none => -1)
                const int max_id = bk ? (int)(0xFFFFu - (bk & 0xFFFFu)) : 0;
                const int max_shared = (int)(bk >> 16);
                int prev = max_id > 0 ? 0 : -1;
                if constexpr (CMODE == CM_SPARSE) {
                    for (uint32_t w = sl; w < CW; w += LPR) {
                        const uint32_t x = ct[w];
                        const int r_ = (int)(x >> 11) - 1, cj = (int)(x & 0x7FFu);
                        if (x != 0u && r_ < max_id && cj > prev) prev = cj;
                    }
                } else if (max_id > 0) {
                    // counters of references below max_id: whole words up to the one that holds reference max_id - 1, of which only
                    // the low fields count
                    const int wb = (max_id - 1) >> clg;
                    const uint32_t kb = (uint32_t)max_id - ((uint32_t)wb << clg);              // 1 .. counters per word
                    const uint32_t mb = 0xFFFFFFFFu >> (32u - kb * cbits);
                    uint32_t acc = 0;
                    for (int w = sl; w <= wb; w += LPR) {
                        const uint32_t x = ct[w] & (w == wb ? mb : 0xFFFFFFFFu);
                        if constexpr (CMODE == CM_DENSE8) {
                            const uint32_t m01 = (x & 0xFFu) > ((x >> 8) & 0xFFu) ? (x & 0xFFu) : ((x >> 8) & 0xFFu);
                            const uint32_t m23 = ((x >> 16) & 0xFFu) > (x >> 24) ? ((x >> 16) & 0xFFu) : (x >> 24);
                            const uint32_t m = m01 > m23 ? m01 : m23;
                            acc = acc > m ? acc : m;
                        } else {
                            const uint32_t m = (x & 0xFFFFu) > (x >> 16) ? (x & 0xFFFFu) : (x >> 16);
                            acc = acc > m ? acc : m;
                        }
                    }
                    prev = (int)acc > prev ? (int)acc : prev;
                }
                prev = LPR == 16 ? row_max_i32(prev) : half_row_max_i32(prev);
                if (sl == 0) reinterpret_cast<int4*>(out4)[r0 + t] = make_int4(max_id, max_shared, max_shared - prev, nmins < geo.nmin_cap ? nmins : geo.nmin_cap);
            }
        }
    }
}

// win_total: windows of a read of the hinted length, all k-mer sizes together (bounds every per-reference count)
bool make_kmer_geom(KmerGeom& g, int maxlen, int nref, int expect_hits, int win_total, const int* nw_k, int nk, int nq, int& cmode, bool& big) {
    if (maxlen < 1) maxlen = 1;
    cmode = win_total <= 255 ? CM_DENSE8 : CM_DENSE16;
    const int clg = cmode == CM_DENSE8 ? 2 : 1;
    g.cwords = (nref + (1 << clg) - 1) >> clg;
    // Many references: a dense counter row per read would eat the LDS budget (and reference ids beyond 2048 would not fit at all),
    // so the row becomes a 128-entry map of the references the read actually hits.
    if (nref > 0 && g.cwords >= 129) { cmode = CM_SPARSE; g.cwords = 128; }
    g.cwords = (g.cwords + 3) & ~3; // rows stay 16-byte aligned
    int ds = 64;
    while (ds < 3 * expect_hits && ds < 1024) ds <<= 1;
    g.dset = ds;
    int T = (nq * 1024 - 15) / maxlen;
    if (T > KM_MAX_T) T = KM_MAX_T;
    if (T < 1) T = 1;
    g.T = T;
    big = false;
    // the small layout if at least half the reads the staged quads could hold fit it, else the large one
    while (g.T > 1 && km_lds_bytes(g, nq) > (size_t)KM_LDS_SMALL) g.T -= 1;
    if (km_lds_bytes(g, nq) > (size_t)KM_LDS_SMALL || 2 * g.T < T) {
        big = true;
        g.T = T;
        while (g.T > 1 && km_lds_bytes(g, nq) > (size_t)KM_LDS_BIG) g.T -= 1;
        if (km_lds_bytes(g, nq) > (size_t)KM_LDS_BIG) return false;
    }
    g.xcd = 1;
    g.L = maxlen;
    for (int j = 0; j < KM_MAX_KS; ++j) {
        g.gpr[j] = j < nk ? (nw_k[j] + 3) >> 2 : 0;
        g.magic[j] = g.gpr[j] >= 2 ? 0xFFFFFFFFu / (uint32_t)g.gpr[j] + 1u : 0u;
    }
    return true;
}

template <int KT>
hipError_t launch_k(int nq, int cmode, bool big, bool fam, dim3 grid, hipStream_t st, const uint8_t* bases, const uint32_t* offs, uint32_t nreads,
                    int S, const RefIndex& ix, const KmerSets& ksets, int32_t* out4, const DevPolicy& pol, const KmerGeom& geo) {
#define RK_KM_GO(NQ, CM, BIG, FAM) hipLaunchKernelGGL((k_classify_kmer<KT, NQ, CM, BIG, FAM>), grid, dim3(KW), 0, st, bases, offs, nreads, S, ix, ksets, out4, pol, geo)
    // (the sparse counters cannot subtract: they walk the plain form of every list and never see a (base, exceptions) value)
#define RK_KM_CM(NQ, BIG)                                                                                    \
    do {                                                                                                     \
        if (cmode == CM_DENSE8) { if (fam) RK_KM_GO(NQ, CM_DENSE8, BIG, true); else RK_KM_GO(NQ, CM_DENSE8, BIG, false); } \
        else if (cmode == CM_DENSE16) { if (fam) RK_KM_GO(NQ, CM_DENSE16, BIG, true); else RK_KM_GO(NQ, CM_DENSE16, BIG, false); } \
        else RK_KM_GO(NQ, CM_SPARSE, BIG, true);                                                             \
    } while (0)
    if (nq == 1 && !big) RK_KM_CM(1, false);
    else if (nq == 1) RK_KM_CM(1, true);
    else if (!big) RK_KM_CM(2, false);
    else RK_KM_CM(2, true);
#undef RK_KM_CM
#undef RK_KM_GO
    return hipGetLastError();
}

} // namespace

// reads of up to 2 * 1024 - 15 bytes (two quads per lane), panels the 16-bit reference field of the running maximum can name
bool classify_kmer_supported(int nref, int maxlen, int k) {
    return nref <= 16384 && maxlen <= 2 * 1024 - 15 && k >= KPRE_MIN_K && k <= KW_MAX_K; // (17 .. 20: the wide form, when its structures were built)
}

// ksets: the structures of every k-mer size of the run (n = 1: the compile-time-k kernels, whose structures are ix.kf4 / km1 too)
hipError_t launch_classify_kmer(const uint8_t* bases, const uint32_t* offs, uint32_t nreads, const KmerSets& ksets, int S, const RefIndex& ix,
                                int32_t* out4, const DevPolicy& pol, int maxlen, int expect_hits, hipStream_t st, int nmin_cap) {
    if (nreads == 0) return hipSuccess;
    if (ksets.n < 1 || ksets.n > KM_MAX_KS) return hipErrorInvalidValue;
    const int nq = maxlen <= 1024 - 15 ? 1 : 2;
    int nw_k[KM_MAX_KS] = {0}, win_total = 0;
    for (int j = 0; j < ksets.n; ++j) { nw_k[j] = num_windows(maxlen, ksets.k[j], pol.drop_last_window); win_total += nw_k[j]; }
    KmerGeom geo;
    int cmode = 0;
    bool big = false;
    if (!make_kmer_geom(geo, maxlen, ix.nref, expect_hits, win_total, nw_k, ksets.n, nq, cmode, big)) return hipErrorInvalidConfiguration;
    geo.nmin_cap = nmin_cap;
    const uint32_t ntiles = (nreads + (uint32_t)geo.T - 1) / (uint32_t)geo.T;
    const bool fam = ix.kbase_n != 0u; // lists stored as (base, exceptions) exist
    uint32_t grid = ntiles;
    grid = (grid + 7u) & ~7u; // whole rounds of the 8 XCDs: the virtual ids then cover [0, grid) exactly
    if (ksets.n > 1) return launch_k<0>(nq, cmode, big, fam, dim3(grid), st, bases, offs, nreads, S, ix, ksets, out4, pol, geo);
#define RK_KM_K(KT) case KT: return launch_k<KT>(nq, cmode, big, fam, dim3(grid), st, bases, offs, nreads, S, ix, ksets, out4, pol, geo)
    switch (ksets.k[0]) {
        RK_KM_K(8); RK_KM_K(9); RK_KM_K(10); RK_KM_K(11); RK_KM_K(12); RK_KM_K(13); RK_KM_K(14); RK_KM_K(15); RK_KM_K(16);
        case 17: case 18: case 19: case 20: // wide k-mers: one run-time-k instantiation (KT = 32)
            return ix.kkeys ? launch_k<32>(nq, cmode, big, fam, dim3(grid), st, bases, offs, nreads, S, ix, ksets, out4, pol, geo) : hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
#undef RK_KM_K
}

} // namespace rk
